#!/bin/bash
# Lock-step decompression, A/B on one box: pools 1 / 2, worker threads pinned or not, every case twice.
#   scripts/exp_decode_ab.sh "64 256" [bytes = 3000]
cd "$(dirname "$0")/.."
LIST=${1:-"64 256"}; N=${2:-3000}
W=$(mktemp -d)
echo "host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), load $(cut -d' ' -f1-3 /proc/loadavg), $N bytes per file"
for S in $LIST; do
  rm -rf $W/f $W/c; mkdir -p $W/f
  for i in $(seq 0 $((S-1))); do python3 scripts/corpus.py $W/f/$(printf %04d $i) $N $((i*1531)) > /dev/null; done
  dropin/_build/gmix_chain_many $W/c $W/f/* > /dev/null 2>&1
  C=$(for i in $(seq 0 $((S-1))); do echo $W/c/$i.gmix; done)
  for rep in 1 2; do
    for G in 1 2; do
      for PIN in "" "--no-pin"; do
        dropin/_build/gmix_chain_many -d --groups $G $PIN $W/b $C > $W/j.json 2> $W/err
        python3 -c "import json;j=json.load(open('$W/j.json'));print('S=%d pools=$G %-8s: %6.1f us per cycle of all pools, %.3g bits/s in the loops, failed %d; whole process %.1f s wall, %.1f s of CPU, %d involuntary switches' % (j['files'], '$PIN' or 'pinned', j['wall_seconds']*1e6/j['launches']*$G, j['bits_per_second'], j['failed'], j['total_seconds'], j['cpu_seconds'], j['involuntary_switches']))"
      done
    done
  done
done
rm -rf $W
