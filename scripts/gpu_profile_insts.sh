#!/bin/bash
# Instruction-issue counters for a bench command (separate --pmc pass, kernel-trace only):
#   bash scripts/gpu_profile_insts.sh <tag> <python script> [args...]
set -e
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/insts -o insts -- python3 "$@" > $out/insts.log 2>&1 || { tail -5 $out/insts.log; exit 1; }
find $out/insts -name '*counter_collection.csv' | head -2
