#!/bin/bash
# Where a wave's cycles go (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAIT_ANY = parked at s_waitcnt,
# SQ_WAIT_INST_ANY = issue stalls, SQ_ACTIVE_INST_ANY = issuing; the three add up to SQ_WAVE_CYCLES.
#   bash scripts/gpu_profile_waits.sh <tag> <bench.py args...>
set -e
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $out/waits -o waits -- python3 bench.py "$@" --no-cpu-baseline --no-also > $out/waits.log 2>&1 || { tail -5 $out/waits.log; exit 1; }
python3 scripts/pmc_avg.py $out/waits gmx_stock
