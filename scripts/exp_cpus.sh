#!/bin/bash
# S files side by side with the threads kept on k cores of the device's node: scripts/exp_cpus.sh "0 16 32" "64 128" [bytes]
cd "$(dirname "$0")/.."
KS=${1:-"0 16 32"}
LIST=${2:-"64"}
N=${3:-30000}
W=$(mktemp -d)
cat DESIGN.md SURVEY.md INTEGRATION.md README.md DESIGN.md SURVEY.md INTEGRATION.md README.md DESIGN.md SURVEY.md INTEGRATION.md README.md > $W/corpus
. scripts/_paths.sh
echo "host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2), $(nproc) cpus visible, cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), $N bytes per file"
for S in $LIST; do
  rm -rf $W/f; mkdir -p $W/f
  for i in $(seq 0 $((S-1))); do tail -c +$((i*1531+1)) $W/corpus | head -c $N > $W/f/$i; done
  for k in $KS; do for exe in ${EXES:-gmix_chain_many gmix_many}; do
    $(gmxbin $exe) -T ${CHUNK:-2048} --cpus $k $W/out $W/f/* > $W/j.json
    python3 -c "import json;j=json.load(open('$W/j.json'));print('$exe S=%d cpus=%d: %.3g bits/s aggregate, %.2f s wall, submit %.2f wait %.2f' % (j['files'],j['pinned_cpus'],j['bits_per_second'],j['wall_seconds'],j['submit_seconds'],j['wait_seconds']))"
  done; done
done
rm -rf $W
