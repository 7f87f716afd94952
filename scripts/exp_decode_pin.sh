#!/bin/bash
# Lock-step decompression: where the worker threads may run -- W cores of the device's node ("narrow"), the device's
# share of the node (default), the whole node, anywhere (--no-pin) -- every case twice, with what the host did to them.
#   scripts/exp_decode_pin.sh "64 256" [bytes = 3000]
cd "$(dirname "$0")/.."
LIST=${1:-"64 256"}; N=${2:-3000}
W=$(mktemp -d)
python3 scripts/host_busy.py 1 2>/dev/null | sed -n 1,3p
echo "host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), load $(cut -d' ' -f1-3 /proc/loadavg), $N bytes per file"
for S in $LIST; do
  rm -rf $W/f $W/c; mkdir -p $W/f
  for i in $(seq 0 $((S-1))); do python3 scripts/corpus.py $W/f/$(printf %04d $i) $N $((i*1531)) > /dev/null; done
  dropin/_build/gmix_chain_many $W/c $W/f/* > /dev/null 2>&1
  C=$(for i in $(seq 0 $((S-1))); do echo $W/c/$i.gmix; done)
  for rep in 1 2; do
    for MODE in ${MODES:-narrow share node none}; do
      PIN=""; [ $MODE = none ] && PIN="--no-pin"
      GMX_PIN_MODE=$MODE dropin/_build/gmix_chain_many -d --groups ${GROUPS_N:-1} $PIN $W/b $C > $W/j.json 2> $W/err
      python3 -c "import json;j=json.load(open('$W/j.json'));print('S=%d %-6s: %6.1f us per step, %.3g bits/s in the loops, failed %d; whole process %.1f s wall, %.1f s of CPU, %d involuntary switches' % (j['files'], '$MODE', j['wall_seconds']*1e6/j['launches'], j['bits_per_second'], j['failed'], j['total_seconds'], j['cpu_seconds'], j['involuntary_switches']))"
    done
  done
done
rm -rf $W
