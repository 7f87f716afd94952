#!/bin/bash
# rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes for any bench script:
#   bash scripts/gpu_profile_script.sh <tag> <script.py> [args...]
set -e
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 "$@" > $out/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- python3 "$@" > $out/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- python3 "$@" > $out/write.log 2>&1
tail -1 $out/stats.log | cut -c1-200
