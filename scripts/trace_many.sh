#!/bin/bash
# Where the submitting thread's time goes with S files side by side (GMX_POOL_TRACE): scripts/trace_many.sh "16 64" [bytes]
cd "$(dirname "$0")/.."
LIST=${1:-"16 64"}
N=${2:-30000}
W=$(mktemp -d)
cat DESIGN.md SURVEY.md INTEGRATION.md README.md DESIGN.md SURVEY.md INTEGRATION.md README.md DESIGN.md SURVEY.md INTEGRATION.md README.md > $W/corpus
. scripts/_paths.sh
for S in $LIST; do
  rm -rf $W/f; mkdir -p $W/f
  for i in $(seq 0 $((S-1))); do tail -c +$((i*1531+1)) $W/corpus | head -c $N > $W/f/$i; done
  echo "== chain, $S files x $N bytes"
  GMX_POOL_TRACE=1 $(gmxbin gmix_chain_many) -T ${CHUNK:-2048} $W/out $W/f/* 2>&1 | tail -40
done
rm -rf $W
