#!/bin/bash
# Everything profiles/rNN_* is made from, in three GPU calls (each inside gpurun's 20 minutes):
#   bash scripts/gpu_round_profiles.sh <round> a|b|c
# (summaries are produced afterwards, in the build container, by scripts/round_profiles_summary.sh)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
PART=${2:-abc}
if [[ $PART == *a* ]]; then
S=$(date +%s)
timeout -k 10 600 python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $? in $(( $(date +%s) - S )) s"
bash scripts/gpu_profile.sh bench_single --config single --steps 12
bash scripts/gpu_profile.sh synth3 --config synth3 --steps 6
bash scripts/gpu_profile.sh stock_held --config stock_held --steps 8
bash scripts/gpu_profile.sh stock_fresh --config stock_fresh --steps 8
bash scripts/gpu_profile.sh stock_real --config stock_real --steps 8
bash scripts/gpu_profile.sh stock_S1 --config stock_S1 --steps 4
fi
if [[ $PART == *b* ]]; then
rm -f gpurun_out/stock_waits.txt
for c in "1024 --config stock_held --streams 1024" "256 --config stock_held --streams 256" "1 --config stock_S1"; do
  set -- $c; tag=$1; shift
  echo "== $* " >> gpurun_out/stock_waits.txt
  bash scripts/gpu_profile_waits.sh w$tag "$@" --steps 4 >> gpurun_out/stock_waits.txt 2>&1
done
export GMX_LIB=$PWD/gmix_amd/libgmxmix_prof.so
for c in "1 4096 2" "256 256 2" "1024 256 2" "1024 256 4" "1024 256 0"; do timeout -k 10 120 python3 scripts/stock_phase_profile.py $c; done > gpurun_out/stock_phase_profile.txt 2>&1
for c in "1 256" "256 200" "1024 200"; do timeout -k 10 120 python3 scripts/lstm_phase_profile.py $c; done > gpurun_out/lstm_phase_profile.txt 2>&1
unset GMX_LIB
bash scripts/gpu_profile_indirect.sh
bash scripts/gpu_profile_lstm.sh
timeout -k 10 300 python3 scripts/bench_indirect.py > gpurun_out/indirect_bench.json 2> gpurun_out/indirect_bench.err
timeout -k 10 300 python3 scripts/bench_real_trace.py > gpurun_out/real_trace.json 2> gpurun_out/real_trace.err
timeout -k 10 300 python3 scripts/bench_lstm.py > gpurun_out/lstm_bench.json 2> gpurun_out/lstm_bench.err
timeout -k 10 300 python3 scripts/bench_pipeline.py > gpurun_out/pipeline_bench.json 2> gpurun_out/pipeline_bench.err
fi
if [[ $PART == *c* ]]; then
# end to end: one file through every build, many files side by side, the kernel timeline
timeout -k 10 400 bash scripts/e2e_batched.sh 100000 gpurun_out/e2e.txt > /dev/null 2>&1
EXES=gmix_chain_many timeout -k 10 300 bash scripts/exp_cpus.sh "-1" "1 16 64 128" > gpurun_out/many_chain.txt 2>&1
timeout -k 10 300 bash scripts/many_scaling.sh "16 64" 30000 gpurun_out/many_scaling.txt > /dev/null 2>&1
timeout -k 10 200 bash scripts/trace_chain_timeline.sh 30000 2048 1 > gpurun_out/timeline_S1.txt 2>&1
timeout -k 10 200 bash scripts/trace_chain_timeline.sh 30000 2048 64 > gpurun_out/timeline_S64.txt 2>&1
fi
echo done
