#!/bin/bash
# Lock-step decompression, the whole process: where the seconds outside the coding loops go.
#   scripts/exp_decode_cold.sh [files = 256] [bytes = 3000]
cd "$(dirname "$0")/.."
S=${1:-256}; N=${2:-3000}
W=$(mktemp -d)
mkdir -p $W/f
for i in $(seq 0 $((S-1))); do python3 scripts/corpus.py $W/f/$(printf %04d $i) $N $((i*1531)) > /dev/null; done
dropin/_build/gmix_chain_many $W/c $W/f/* > /dev/null 2>&1
C=$(for i in $(seq 0 $((S-1))); do echo $W/c/$i.gmix; done)
for FLAG in ${FLAGS:-"--plain-exit" "" "--plain-exit" ""}; do
  T0=$(date +%s.%N)
  GMX_POOL_TRACE=1 dropin/_build/gmix_chain_many -d $FLAG $W/b $C > $W/j.json 2> $W/err
  T1=$(date +%s.%N)
  grep "gmx decode" $W/err
  python3 -c "import json;j=json.load(open('$W/j.json'));print('S=%d $FLAG: process %.2f s by the shell; call %.2f s = first Predictor %.2f + the rest of the build %.2f + loops %.2f + teardown %.2f; %.1f s of CPU' % (j['files'], $T1-$T0, j['total_seconds'], j['first_predictor_seconds'], j['build_seconds']-j['first_predictor_seconds'], j['wall_seconds'], j['teardown_seconds'], j['cpu_seconds']))"
done
rm -rf $W
