#!/bin/bash
# Turn what scripts/gpu_round_profiles.sh left in gpurun_out/ into profiles/rNN_*:  bash scripts/round_profiles_summary.sh <round>
R=${1:-2}
cd "$(dirname "$0")/.."
S="--no-cpu-baseline --no-also"
python3 scripts/pmc_summary.py bench_single $R gmx_single_kernel 4096 1024 3080 "python bench.py --config single --steps 12 $S" 0 > /dev/null
python3 scripts/pmc_summary.py synth3 $R gmx_wide_kernel 2048 512 54608 "python bench.py --config synth3 --steps 6 $S" 0 > /dev/null
python3 scripts/pmc_summary.py stock_held $R gmx_stock_kernel 1024 1024 3193 "python bench.py --config stock_held --steps 8 $S" 2 > /dev/null
python3 scripts/pmc_summary.py stock_real $R gmx_stock_kernel 1024 1024 4943 "python bench.py --config stock_real --steps 8 $S" 4 > /dev/null
python3 scripts/pmc_summary.py stock_fresh $R gmx_stock_kernel 1024 1024 22072 "python bench.py --config stock_fresh --steps 8 $S" 0 > /dev/null
python3 scripts/pmc_summary.py indirect $R gmx_indirect_kernel 256 4096 743 "python scripts/bench_indirect.py" 0 > /dev/null
python3 scripts/pmc_summary.py lstm $R gmx_lstm_kernel 1024 200 452356 "python scripts/bench_lstm.py" 0 > /dev/null
python3 scripts/pmc_summary.py stock_S1 $R gmx_stock_kernel 1 8192 3193 "python bench.py --config stock_S1 --steps 4 $S" 2 > /dev/null
for f in e2e.txt many_scaling.txt many_chain.txt timeline_S1.txt timeline_S64.txt lstm_phase_profile.txt; do [ -f gpurun_out/$f ] && cp gpurun_out/$f profiles/r$(printf %02d $R)_$f; done
for f in lstm_bench pipeline_bench; do [ -s gpurun_out/$f.json ] && cp gpurun_out/$f.json profiles/r$(printf %02d $R)_$f.json; done
[ -s gpurun_out/bench_default.json ] && cp gpurun_out/bench_default.json profiles/r$(printf %02d $R)_bench_default.json
cp gpurun_out/indirect_bench.json profiles/r$(printf %02d $R)_indirect_bench.json
cp gpurun_out/real_trace.json profiles/r$(printf %02d $R)_real_trace_bench.json
cp gpurun_out/stock_phase_profile.txt profiles/r$(printf %02d $R)_stock_phase_profile.txt
python3 - "$R" <<'PY'
import csv, glob, collections, json, sys
R = int(sys.argv[1])
out = []
for tag, S, T in (("w1024", 1024, 1024), ("w256", 256, 1024), ("w1", 1, 8192)):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/prof_{tag}/waits/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "gmx_stock" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    d = {k: round(sum(v) / len(v) / (S * T) * (4 if ("WAIT" in k or "ACTIVE" in k or "WAVE_CYC" in k) else 1), 1)
         for k, v in acc.items() if k != "SQ_WAVES"}
    out.append(f"stock_held, {S} stream(s) x {T} bits; per stream-bit: cycles (quad-cycle counters x 4), instructions: " + json.dumps(d, sort_keys=True))
open(f"profiles/r{R:02d}_stock_waits.txt", "w").write(
    "# scripts/gpu_profile_waits.sh: rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES\n"
    + "\n".join(out) + "\n")
for f in sorted(glob.glob(f"profiles/r{R:02d}_*_pmc_summary.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], d["kernel"].split("(")[0][-40:], "ms", round(d["kernel_ms_avg_rocprof"], 4), "traffic/alg", round(d["traffic_over_algorithmic"], 3), (d.get("build") or "-")[-12:])
print("\n".join(out))
PY
