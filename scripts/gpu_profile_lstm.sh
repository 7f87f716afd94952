#!/bin/bash
# rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes for scripts/bench_lstm.py (the bench's 1 024 streams x 200 bytes).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/prof_lstm
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 scripts/bench_lstm.py --cpu-sample-bytes 100 > $out/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- python3 scripts/bench_lstm.py --cpu-sample-bytes 100 > $out/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- python3 scripts/bench_lstm.py --cpu-sample-bytes 100 > $out/write.log 2>&1
tail -1 $out/stats.log | cut -c1-300
