#!/bin/bash
# rocprofv3 kernel stats of the run-ahead chain for S files side by side:  scripts/prof_chain_many.sh S bytes outprefix
cd "$(dirname "$0")/.."
S=${1:-128}; N=${2:-8000}; OUT=${3:-gpurun_out/chain_many_S$S}
W=$(mktemp -d); cat DESIGN.md SURVEY.md INTEGRATION.md DESIGN.md SURVEY.md INTEGRATION.md DESIGN.md SURVEY.md INTEGRATION.md DESIGN.md SURVEY.md INTEGRATION.md > $W/c
mkdir $W/f; for i in $(seq 0 $((S-1))); do tail -c +$((i*1531+1)) $W/c | head -c $N > $W/f/$i; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $W/prof -o run -- dropin/_build/gmix_chain_many -T 2048 $W/out $W/f/* > $OUT.json 2> $OUT.err
find $W/prof -name "*kernel_stats.csv" -exec cp {} ${OUT}_kernel_stats.csv \;
head -12 ${OUT}_kernel_stats.csv | cut -c1-200
rm -rf $W
