#!/bin/bash
# Whole compressor end to end on the GPU box: the stock build (reference, CPU), the per-bit drop-in and the run-ahead
# compressor (gmix_amd/host/gmx_batched.h) on the same input -- identical files, time per bit of each.
#   scripts/e2e_batched.sh [bytes] [out file]        GMX_CORPUS=/path/to/enwik8 picks the input (default: this repo's docs)
set -e
cd "$(dirname "$0")/.."
N=${1:-100000}
OUT=${2:-gpurun_out/e2e_batched.txt}
W=$(mktemp -d)
SRC=${GMX_CORPUS:-}
if [ -z "$SRC" ]; then  # this repository's documents, repeated until there are N bytes
  : > $W/corpus
  while [ $(wc -c < $W/corpus) -lt $N ]; do cat DESIGN.md SURVEY.md INTEGRATION.md README.md >> $W/corpus; done
  SRC=$W/corpus
fi
head -c $N "$SRC" > $W/in
. scripts/_paths.sh
mkdir -p $(dirname $OUT)
{
echo "input: $(wc -c < $W/in) bytes of ${GMX_CORPUS:-DESIGN.md+SURVEY.md+INTEGRATION.md+README.md}; host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2), $(nproc) cores visible"
for exe in ${EXES:-gmix_strict gmix_gpu gmix_batched gmix_chain_batched}; do
  mkdir -p $W/$exe; ( cd $W/$exe; s=$(date +%s.%N); $(gmxbin $exe) -c $W/in out > log 2>/dev/null; e=$(date +%s.%N);
  echo "$exe: $(wc -c < out) bytes, $(echo "$e $s $N" | awk '{printf "%.2f s, %.2f us per bit", $1-$2, ($1-$2)*1e6/(8*$3)}') (whole process: Predictor construction included), md5 $(md5sum < out | cut -c1-12)" )
done
for exe in gmix_gpu gmix_batched gmix_chain_batched; do
  [ -f $W/$exe/out ] || continue
  cmp $W/gmix_strict/out $W/$exe/out && echo "$exe -c == gmix_strict -c"
  [ $exe = gmix_gpu ] || { cmp $W/gmix_strict/analysis/entropy.tsv $W/$exe/analysis/entropy.tsv && cmp $W/gmix_strict/analysis/memory.tsv $W/$exe/analysis/memory.tsv && echo "analysis tables identical ($exe)"; }
done
last=$(ls -d $W/gmix_chain_batched $W/gmix_batched 2>/dev/null | head -1)
( cd $W/gmix_strict; $(gmxbin gmix_strict) -d $last/out back > /dev/null 2>&1; cmp back $W/in && echo "gmix_strict -d restores what $(basename $last) -c wrote" )
} | tee $OUT
rm -rf $W
