cd "$(dirname "$0")/.."
head -c 30000 DESIGN.md > /tmp/f0
for T in 512 2048 8192; do
  dropin/_build/gmix_chain_many -T $T /tmp/o /tmp/f0 | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('chain S=1 T=$T: %.2f us/bit, submit %.2f s wait %.2f s of %.2f s wall, launches %d' % (j['wall_seconds']*1e6/240000, j['submit_seconds'], j['wait_seconds'], j['wall_seconds'], j['launches']))"
done
GPU_MAX_HW_QUEUES=8 dropin/_build/gmix_chain_many -T 2048 /tmp/o /tmp/f0 | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('chain S=1 T=2048 HWQ=8: %.2f us/bit, submit %.2f s wait %.2f s of %.2f s wall' % (j['wall_seconds']*1e6/240000, j['submit_seconds'], j['wait_seconds'], j['wall_seconds']))"
dropin/_build/gmix_many -T 2048 /tmp/o /tmp/f0 | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('mixers S=1 T=2048: %.2f us/bit, submit %.2f s wait %.2f s of %.2f s wall' % (j['wall_seconds']*1e6/240000, j['submit_seconds'], j['wait_seconds'], j['wall_seconds']))"
