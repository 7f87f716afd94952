#!/bin/bash
# Soak of the many-files path: S files of different lengths, an odd chunk size, every output against `gmix_strict -c`.
#   scripts/soak_many.sh [S = 100] [chunk bits = 1992] [base bytes = 20000] [step = 137] [out = gpurun_out/soak_many.txt]
cd "$(dirname "$0")/.."
S=${1:-100}; T=${2:-1992}; B=${3:-20000}; D=${4:-137}; OUT=${5:-gpurun_out/soak_many.txt}
. scripts/_paths.sh
W=$(mktemp -d); mkdir $W/f $W/ref
: > $W/corpus; for i in 1 2 3 4 5 6; do cat DESIGN.md SURVEY.md INTEGRATION.md README.md >> $W/corpus; done
for i in $(seq 0 $((S-1))); do tail -c +$((i*2731+1)) $W/corpus | head -c $((B + i*D)) > $W/f/$(printf %03d $i); done
{
echo "$S files of $B .. $((B + (S-1)*D)) bytes, chunks of $T bits, $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for exe in gmix_chain_many; do
  $(gmxbin $exe) -T $T $W/out_$exe $W/f/* > $W/j.json 2> $W/err || { echo "$exe FAILED"; tail -3 $W/err; }
  python3 -c "import json;j=json.load(open('$W/j.json'));print('$exe: %d files, %d failed, %.3g bits/s aggregate, %.2f s wall, %d launches' % (j['files'],j['failed'],j['bits_per_second'],j['wall_seconds'],j['launches']))"
done
ls $W/f | xargs -P 16 -I{} sh -c "mkdir -p $W/ref/{} && cd $W/ref/{} && $(gmxbin gmix_strict) -c $W/f/{} out > /dev/null 2>&1"
same=0; diff=0
k=0
for f in $(ls $W/f); do
  if cmp -s $W/ref/$f/out $W/out_gmix_chain_many/$k.gmix; then same=$((same+1)); else diff=$((diff+1)); echo "file $f differs"; fi
  k=$((k+1))
done
echo "identical to gmix_strict -c: $same of $S; different: $diff"
} | tee $OUT
rm -rf $W
