#!/bin/bash
# How much CPU time the many-files run takes against its wall time: scripts/exp_cputime.sh "1 16 64"
cd "$(dirname "$0")/.."
W=$(mktemp -d)
cat DESIGN.md SURVEY.md INTEGRATION.md README.md DESIGN.md SURVEY.md INTEGRATION.md README.md > $W/corpus
. scripts/_paths.sh
for S in ${1:-"1 16 64"}; do
  rm -rf $W/f; mkdir -p $W/f
  for i in $(seq 0 $((S-1))); do tail -c +$((i*1531+1)) $W/corpus | head -c 60000 > $W/f/$i; done
  a=$(grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
  TIMEFORMAT="S=$S: wall %R s, user %U s, sys %S s"
  time $(gmxbin gmix_chain_many) $W/out $W/f/* > $W/j.json
  python3 -c "import json;j=json.load(open('$W/j.json'));print('   compression phase %.2f s, %.3g bits/s' % (j['wall_seconds'], j['bits_per_second']))"
  echo "   cgroup before: $a"
  echo "   cgroup after:  $(grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')"
done
rm -rf $W
