#!/bin/bash
# Memory-pipeline counters for a bench command (separate --pmc passes, kernel-trace only):
#   bash scripts/gpu_profile_mem.sh <tag> <python script> [args...]
set -e
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --list-avail > $out/avail.txt 2>&1 || true
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/mem$i -o mem -- python3 "$@" > $out/mem$i.log 2>&1 || { echo "set $i failed"; tail -3 $out/mem$i.log; }
done
find $out -name '*counter_collection.csv' | head
