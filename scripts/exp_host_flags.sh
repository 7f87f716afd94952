#!/bin/bash
# Does it pay to build the drop-in's host side (the reference's feature models, compiled where they lie) with
# -O3 -march=x86-64-v3 -ffp-contract=off instead of the strict oracle's -O2?  (Value-safe flags: no fast-math, no
# contraction -- SURVEY.md section 8c found such builds byte-identical.)  64 files x N bytes through both builds of
# gmix_chain_many: rate of the coding loops, and whether every output is the same file.
#   make -C dropin OBJ=/tmp/gmx_v3 OUT=$PWD/dropin/_build_v3 STRICT='-std=c++17 -O3 -march=x86-64-v3 -ffp-contract=off -w -include cstring' $PWD/dropin/_build_v3/gmix_chain_many
#   bash scripts/exp_host_flags.sh [bytes = 100000] [files = 64]
cd "$(dirname "$0")/.."
N=${1:-100000}; S=${2:-64}
W=$(mktemp -d); mkdir $W/f
for i in $(seq 0 $((S-1))); do python3 scripts/corpus.py $W/f/$(printf %03d $i) $N $((i*157000 % 9000000)) > /dev/null; done
for b in _build _build_v3 _build _build_v3; do
  [ -x dropin/$b/gmix_chain_many ] || continue
  dropin/$b/gmix_chain_many $W/out$b $W/f/* > $W/j.json 2> $W/err || { echo "$b failed"; tail -3 $W/err; continue; }
  python3 -c "import json;j=json.load(open('$W/j.json'));print('$b: coding loops %.3g bits/s (%.2f s), cold %.3g bits/s, setup %.2f s, device wait %.0f%%' % (j['bits_per_second'], j['wall_seconds'], j['bits_per_second_cold'], j['build_seconds'], 100*j['wait_seconds']/j['wall_seconds']))"
done
same=0; for i in $(seq 0 $((S-1))); do cmp -s $W/out_build/$i.gmix $W/out_build_v3/$i.gmix && same=$((same+1)); done
echo "outputs identical between the two builds: $same of $S"
rm -rf $W
