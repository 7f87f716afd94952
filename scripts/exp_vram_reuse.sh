#!/bin/bash
cd "$(dirname "$0")/.." 2>/dev/null
S=256; N=1500
W=$(mktemp -d); mkdir -p $W/f
for i in $(seq 0 $((S-1))); do python3 scripts/corpus.py $W/f/$(printf %04d $i) $N $((i*1531)) > /dev/null; done
dropin/_build/gmix_chain_many $W/c $W/f/* > /dev/null 2>&1
C=$(for i in $(seq 0 $((S-1))); do echo $W/c/$i.gmix; done)
for pause in 0 0 8 0 15; do
  sleep $pause
  GMX_POOL_TRACE=1 dropin/_build/gmix_chain_many -d $W/b $C 2>&1 >/dev/null | grep "gmx decode" | sed "s/^/after a pause of $pause s: /" | cut -c1-230
done
rm -rf $W
