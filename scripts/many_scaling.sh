#!/bin/bash
# How many files side by side pay: gmix_many (run-ahead, one device group) against the same number of stock
# `gmix -c` processes run at once on the same host.    scripts/many_scaling.sh "8 16 32 64" [bytes] [out]
cd "$(dirname "$0")/.."
LIST=${1:-"8 16 32 64"}
N=${2:-30000}
OUT=${3:-gpurun_out/many_scaling.txt}
W=$(mktemp -d)
cat DESIGN.md SURVEY.md INTEGRATION.md README.md DESIGN.md SURVEY.md INTEGRATION.md README.md DESIGN.md SURVEY.md INTEGRATION.md README.md > $W/corpus
. scripts/_paths.sh
{
echo "host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2), $(nproc) cpus visible, cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), $N bytes per file"
for S in $LIST; do
  rm -rf $W/f; mkdir -p $W/f
  for i in $(seq 0 $((S-1))); do tail -c +$((i*1531+1)) $W/corpus | head -c $N > $W/f/$i; done
  for exe in gmix_many gmix_chain_many; do for mode in "" "--no-pin"; do
    $(gmxbin $exe) -T ${CHUNK:-2048} $mode $W/out $W/f/* > $W/j.json
    python3 -c "import json;j=json.load(open('$W/j.json'));print('$exe S=%d %s: %.3g bits/s aggregate, %.2f s wall, %.2f us per bit per stream, build %.1f s, pinned %d' % (j['files'],'$mode',j['bits_per_second'],j['wall_seconds'],j['wall_seconds']*1e6/(8*$N),j['build_seconds'],j['pinned_threads']))"
  done; done
  s=$(date +%s.%N)
  for i in $(seq 0 $((S-1))); do ( mkdir -p $W/s$i; cd $W/s$i; $(gmxbin gmix_strict) -c $W/f/$i out >/dev/null 2>&1 ) & done; wait
  e=$(date +%s.%N)
  echo "$S stock processes at once: $(echo "$e $s $N $S" | awk '{printf "%.3g bits/s aggregate, %.2f s wall (construction included)", 8*$3*$4/($1-$2), $1-$2}')"
  rm -rf $W/s*
done
} | tee $OUT
rm -rf $W
