#!/bin/bash
# Lock-step decompression: where a step's time goes -- inside gmx_chainstep_step (graph launch, wait for the stamps;
# GMX_STEP_TRACE) against the whole step (the fibres' host work and the barriers are the rest).
#   scripts/exp_decode_split.sh "64 256" [bytes = 3000]
cd "$(dirname "$0")/.."
LIST=${1:-"64 256"}; N=${2:-3000}
W=$(mktemp -d)
for S in $LIST; do
  rm -rf $W/f $W/c; mkdir -p $W/f
  for i in $(seq 0 $((S-1))); do python3 scripts/corpus.py $W/f/$(printf %04d $i) $N $((i*1531)) > /dev/null; done
  dropin/_build/gmix_chain_many $W/c $W/f/* > /dev/null 2>&1
  C=$(for i in $(seq 0 $((S-1))); do echo $W/c/$i.gmix; done)
  GMX_STEP_TRACE=1 dropin/_build/gmix_chain_many -d $W/b $C > $W/j.json 2> $W/err
  grep "gmx step" $W/err | tail -n 2
  python3 -c "import json;j=json.load(open('$W/j.json'));print('S=%d workers=%d: %.1f us per step, %.3g bits/s in the loops' % (j['files'], j['pinned_cpus'], j['wall_seconds']*1e6/j['launches'], j['bits_per_second']))"
done
rm -rf $W
