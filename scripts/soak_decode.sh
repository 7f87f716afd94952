#!/bin/bash
# Lock-step decompression over and over: N runs of 64 files (fibres migrating between workers, ragged ends), every
# restored file compared with its original each time.   scripts/soak_decode.sh [runs = 10] [files = 64] [bytes = 2000]
cd "$(dirname "$0")/.."
RUNS=${1:-10}; S=${2:-64}; N=${3:-2000}
W=$(mktemp -d)
mkdir -p $W/f
for i in $(seq 0 $((S-1))); do python3 scripts/corpus.py $W/f/$(printf %04d $i) $((N + 37 * i)) $((i*1531)) > /dev/null; done
bad=0
for EXE in gmix_chain_many gmix_many; do
  dropin/_build/$EXE $W/c_$EXE $W/f/* > /dev/null 2>&1 || { echo "$EXE: compression failed"; bad=1; continue; }
  C=$(for i in $(seq 0 $((S-1))); do echo $W/c_$EXE/$i.gmix; done)
  for r in $(seq 1 $RUNS); do
    rm -rf $W/b
    G=$(( (r % 2) + 1 ))
    dropin/_build/$EXE -d --groups $G --cpus $(( 3 + (r * 5) % 12 )) $W/b $C > $W/j.json 2> $W/err || { echo "$EXE run $r: exit $?"; tail -n 3 $W/err; bad=1; continue; }
    k=0
    for f in $W/f/*; do cmp -s $f $W/b/$k.out || { echo "$EXE run $r: file $k differs"; bad=1; }; k=$((k+1)); done
    python3 -c "import json;j=json.load(open('$W/j.json'));print('$EXE run $r (pools $G, %d workers): %d files, failed %d, %.1f us per step' % (j['pinned_cpus'], j['files'], j['failed'], j['wall_seconds']*1e6/max(1,j['launches'])))"
  done
done
rm -rf $W
[ $bad = 0 ] && echo "soak: all restored" || { echo "soak: FAILURES"; exit 1; }
