"""bench.py's one JSON line: the driver's contract fields, `roofline`, `cpu_baseline`, and under `also`
the north_star's other shapes -- on reduced sizes (headline launch, end-to-end files) so that the test
takes a minute or two."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_carries_the_contract(gpu):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--streams", "1024", "--bits", "256", "--steps", "4",
                        "--cpu-sample-bits", "100000", "--e2e-bytes", "3000", "--e2e-many-bytes", "3000", "--decode-bytes", "300",
                        "--decode-streams", "8", "--also-streams-div", "4"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "fracs", "also", "tail_summary"):
        assert k in d, k
    assert d["unit"] == "bits/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    kernels = {"synth3", "stock_held", "stock_real", "stock_fresh", "stock_S1", "single_S1", "indirect", "lstm"}
    whole = {"e2e_S1", "e2e_S1_mixers", "e2e_S64"}   # the run-ahead compressor on real files (scripts/bench_e2e.py)
    assert set(d["also"]) - {"e2e_decode"} == kernels | whole | {"real_trace"}
    assert list(d)[-1] == "tail_summary" and list(d).index("fracs") < list(d).index("also")
    assert set(d["fracs"]) >= kernels | {"single", "real_trace"} and d["tail_summary"]["fracs"] == d["fracs"]
    assert set(d["tail_summary"]["e2e"]) >= whole
    rt = d["also"]["real_trace"]   # the stock kernel on the reference's recorded mixer boundary
    assert "error" not in rt, rt
    assert rt["unit"] == "bits/s" and rt["value"] > 1e7 and rt["stream0_first_window_equals_reference"] is True
    assert rt["roofline"]["kernel"] == "gmx_stock_kernel" and 0 < rt["roofline"]["frac"] < 1
    for name in whole:
        e = d["also"][name]
        assert "error" not in e, (name, e)
        assert e["unit"] == "bits/s" and e["value"] > 1e4 and e["identical_to_stock"] is True
        assert e["config"]["streams"] == (64 if name == "e2e_S64" else 1) and e["cpu_baseline"]["value"] > 1e4
        # `value` is the whole process, exec to exit (like the reference CLI it is set against); the coding loops alone are faster
        assert e["value"] <= e["value_coding_loops"] and e["seconds"] >= e["in_process_seconds"] >= e["coding_loops_seconds"]
        assert abs(e["vs_cpu"] - e["value"] / e["cpu_baseline"]["value"]) < 1e-9
    assert d["also"]["e2e_S64"]["predictors_built_side_by_side"] is True
    for name, e in [("headline", d)] + [(k, d["also"][k]) for k in sorted(kernels)]:
        assert "error" not in e, (name, e)
        ro = e["roofline"]
        assert ro["bound"] == "hbm" and ro["peak"] == 8000.0 and ro["unit"] == "GB/s"
        assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12 and 0 < ro["frac"] < 1
        if name not in ("indirect", "lstm"):  # (the producers' entries come from their own scripts: other units)
            assert ro["bytes_per_launch"] == ro["algorithmic_bytes_per_bit"] * e["config"]["streams_per_gpu"] * \
                e["config"]["bits_per_stream_per_step"]
        # the live HIP-event figure and the wall clock tell the same story
        assert ro["kernel_ms_avg"] <= e["ms_per_step"] * 1.02
        assert e["value"] > 0 and e["cpu_baseline"]["cores"] == 1 and e["cpu_baseline"]["kind"] in ("reference", "port")
        assert e["cpu_baseline"]["value"] > 1e4
    assert d["also"]["synth3"]["config"]["n_inputs"] == 256 and d["also"]["synth3"]["config"]["mixers"] == "24/8/1"
    assert d["also"]["stock_S1"]["config"]["streams_per_gpu"] == 1
    assert d["also"]["indirect"]["unit"] == "bits/s" and d["also"]["indirect"]["roofline"]["kernel"] == "gmx_indirect_kernel"
    assert d["also"]["lstm"]["unit"] == "bytes/s" and d["also"]["lstm"]["roofline"]["kernel"] == "gmx_lstm_kernel"
