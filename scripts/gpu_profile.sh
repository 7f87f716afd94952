#!/bin/bash
# Profile one bench.py command on the GPU box: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE
# in separate --pmc passes (MI355X_MICROARCH.md, HBM section).  Usage:
#   bash scripts/gpu_profile.sh <tag> <bench.py args...>
# Output: gpurun_out/prof_<tag>/{stats,fetch,write}/... as CSV; summarise with scripts/pmc_summary.py.
set -e
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 bench.py "$@" --no-cpu-baseline --no-also > $out/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- python3 bench.py "$@" --no-cpu-baseline --no-also > $out/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- python3 bench.py "$@" --no-cpu-baseline --no-also > $out/write.log 2>&1
find $out -name '*.csv' | head -20
tail -1 $out/stats.log
