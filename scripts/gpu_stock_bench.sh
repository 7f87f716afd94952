#!/bin/bash
# Stock-shape (90 inputs, 24/8/1) bench lines: contexts fresh every bit / held for 8 bits.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
for m in 0 2; do
  timeout -k 10 200 python3 bench.py --config stock --streams 1024 --bits 256 --steps 8 --ctx-mode $m --no-cpu-baseline > gpurun_out/stock_m$m.json 2>gpurun_out/stock_m$m.err
  python3 -c "import json;r=json.load(open('gpurun_out/stock_m$m.json'));print('ctx-mode $m', r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'])"
done
timeout -k 10 200 python3 bench.py --config stock --streams 256 --bits 512 --steps 8 --ctx-mode 2 --no-cpu-baseline > gpurun_out/stock_s256.json 2>&1
python3 -c "import json;r=json.load(open('gpurun_out/stock_s256.json'));print('S=256 held', r['value'], r['roofline']['kernel_ms_avg'])"
