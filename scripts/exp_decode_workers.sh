#!/bin/bash
# Lock-step decompression: how many worker threads should carry the fibres?  (They spin at the step's barrier: on a
# container with a CPU quota, workers = quota leaves no room for the runtime's own threads and the group gets throttled.)
#   scripts/exp_decode_workers.sh "64 256" "6 8 12 14 16" [bytes = 3000]
cd "$(dirname "$0")/.."
LIST=${1:-"64 256"}; WS=${2:-"6 8 12 14 16"}; N=${3:-3000}
W=$(mktemp -d)
echo "host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), $N bytes per file"
for S in $LIST; do
  rm -rf $W/f $W/c; mkdir -p $W/f
  for i in $(seq 0 $((S-1))); do python3 scripts/corpus.py $W/f/$(printf %04d $i) $N $((i*1531)) > /dev/null; done
  dropin/_build/gmix_chain_many $W/c $W/f/* > /dev/null 2>&1
  C=$(for i in $(seq 0 $((S-1))); do echo $W/c/$i.gmix; done)
  for w in $WS; do
    a=$(grep -E "nr_throttled" /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
    dropin/_build/gmix_chain_many -d --cpus $w --groups ${GROUPS_N:-1} $W/b $C > $W/j.json 2> $W/err
    b=$(grep -E "nr_throttled" /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
    python3 -c "import json;j=json.load(open('$W/j.json'));print('S=%d workers=%d groups=${GROUPS_N:-1}: %.1f us/step (steps of all groups counted), %.3g bits/s in the loops, throttled periods %d' % (j['files'], j['pinned_cpus'], j['wall_seconds']*1e6/j['launches'], j['bits_per_second'], $b-$a))"
  done
done
rm -rf $W
