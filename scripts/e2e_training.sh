#!/bin/bash
# `gmix -t train test` (runner_utils::RunTraining) through every build: wall time and whether the files agree.
#   scripts/e2e_training.sh [train bytes = 100000] [test bytes = 5000] [out = gpurun_out/training.txt]
cd "$(dirname "$0")/.."
N=${1:-100000}; M=${2:-5000}; OUT=${3:-gpurun_out/training.txt}
REF=$PWD/oracle/_ref
W=$(mktemp -d)
cat DESIGN.md SURVEY.md INTEGRATION.md README.md > $W/corpus
head -c $N $W/corpus > $W/train
tail -c $M $W/corpus > $W/test
{
echo "host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2); train $N bytes, test $M bytes (scored by a copy of the Predictor every other per cent: 49 times)"
for exe in ${EXES:-gmix_strict gmix_fast gmix_batched gmix_chain_batched}; do
  mkdir -p $W/$exe; cd $W/$exe
  s=$(date +%s.%N); $REF/$exe -t $W/train $W/test > log 2>&1; rc=$?; e=$(date +%s.%N)
  cd - > /dev/null
  same=""
  if [ $exe != gmix_strict ]; then
    same="; same as gmix_strict:"
    for f in data/tmp analysis/training.tsv data/trained_checkpoint.long; do cmp -s $W/gmix_strict/$f $W/$exe/$f && same="$same $f yes" || same="$same $f NO"; done
  fi
  echo "$exe -t: rc $rc, $(echo "$e $s" | awk '{printf "%.1f s", $1-$2}'), $(grep -o 'training cross entropy: [0-9.]*' $W/$exe/log)$same"
done
} | tee $OUT
rm -rf $W
