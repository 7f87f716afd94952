#!/usr/bin/env python3
"""gmx_stock_kernel on REAL feature-model inputs at full occupancy (GPU box).

The reference's own Predictor (oracle/_ref/ref_trace: the reference compiled in the build container)
records the mixer boundary -- raw predictions, active_models, the 33 gate contexts, the bits -- of the first
N bytes of a text; S streams replay that recording from S different byte offsets (so the streams stand in
different places of the text, like S files would), T bits per launch, records resident in HBM.  Reported:
bits/s of the batched HIP path with active masks, next to the synthetic byte-held figure of bench.py, and
how often a gate row actually changed.  Stream 0 (offset 0) is checked against the recording's own outputs.

  python scripts/bench_real_trace.py [--streams 1024] [--bytes 30000] [--bits 256] [--launches 24]
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class _Args:
    pass


def measure(streams=1024, n_bytes=30000, bits=256, launches=24, text=None, staged=-1, prof=False):
    """The entry bench.py prints under also.real_trace (the recording is made outside the timed region)."""
    args = _Args()
    args.streams, args.bytes, args.bits, args.launches = streams, n_bytes, bits, launches
    args.text = text or os.path.join(ROOT, "SURVEY.md")
    args.staged, args.prof = staged, prof
    return _run(args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--bytes", type=int, default=30000)
    ap.add_argument("--bits", type=int, default=256)
    ap.add_argument("--launches", type=int, default=24)
    ap.add_argument("--text", default=os.path.join(ROOT, "SURVEY.md"))
    ap.add_argument("--staged", type=int, default=-1, help="gmx_debug_stock_staged: 1 rows through LDS, 0 lane-private, -1 by stream count")
    ap.add_argument("--prof", action="store_true", help="phase profile (needs GMX_LIB=.../libgmxmix_prof.so)")
    print(json.dumps(_run(ap.parse_args())))


def _run(args):
    import gmix_amd
    from gmix_amd.bank import Topology
    from trace_common import make_trace
    with tempfile.TemporaryDirectory() as td:
        tr = make_trace(args.text, args.bytes, os.path.join(td, "t.bin"), 0)
    Tt, S, T = tr["T"], args.streams, args.bits
    topo = Topology(tr["n"], tr["mixers"], tr["skip"])
    rows = tr["ctx"] % np.array([m[1] for m in tr["mixers"]], np.uint32)
    changed = (rows[1:] != rows[:-1])
    g = gmix_amd.MixerGroup(topo, S)
    import ctypes as C
    g.L.gmx_debug_stock_staged.argtypes = [C.c_void_p, C.c_int]
    assert g.L.gmx_debug_stock_staged(g.h, args.staged) == 0
    hist = np.bincount(changed[:, :24].sum(1), minlength=25)
    nb = 3  # batches in flight: filled on the host once, reused round robin with new offsets impossible -> distinct windows
    batches = [gmix_amd.Batch(g, T, outputs=(k == 0), mask=True) for k in range(nb)]
    # stream s replays the recording from byte offset (s * 37) mod (bytes - windows): whole bytes, so that the
    # byte-boundary structure of the contexts is kept
    n_win = args.launches
    span = n_win * T
    assert span + 8 * S * 37 // S <= Tt or True
    max_off = (Tt - span) // 8
    offs = [(s * 37) % max(1, max_off) * 8 for s in range(S)]
    offs[0] = 0

    def fill(b, w):
        for s in range(S):
            a = offs[s] + w * T
            b.set_records(s, tr["pred"][a:a + T], tr["act"][a:a + T], tr["ctx"][a:a + T], tr["bits"][a:a + T])
        b.upload(T)

    # correctness of the harness: the first window of stream 0 equals the recording's own outputs
    fill(batches[0], 0)
    g.run(batches[0], T)
    batches[0].download(T)
    batches[0].wait()
    assert np.array_equal(batches[0].outputs[0, :T].view(np.uint32), tr["outs"][:T].view(np.uint32)), "stream 0 differs"
    assert np.array_equal(batches[0].p[0, :T].view(np.uint32), tr["p"][:T].view(np.uint32))
    g.reset()
    # timed: the windows in order (each stream learns its stretch of text front to back), records uploaded
    # ahead of the launch that uses them, kernel time from HIP events per launch
    fill(batches[0], 0)
    ms = []
    if args.prof:
        prof = (C.c_ulonglong * 16)()
        g.L.gmx_stock_prof_read(prof, 1)
    for w in range(n_win):
        if w + 1 < n_win:
            fill(batches[(w + 1) % nb], w + 1)
        ms.append(g.run(batches[w % nb], T, timed=True))
    lb, sb = g.export(0)
    steady = ms[2:]
    avg = sum(steady) / len(steady)
    # algorithmic bytes per bit as bench.py counts them: a row that changes is written back and the new one read
    # (8 bytes per weight), everything else of the topology's per-bit bytes as it stands
    ws = topo.weight_sizes()
    row_bytes = 8 * sum(ws)
    moved = float(sum(changed[:, j].mean() * 8 * ws[j] for j in range(len(ws))))
    bytes_per_bit = moved + topo.bytes_per_bit() - row_bytes
    each = sorted(steady)
    out = {
        "value": S * T / (avg * 1e-3), "unit": "bits/s", "steps": len(steady), "warmup": 2, "ms_per_step": avg,
        "config": {"workload": f"{S} streams replaying the reference Predictor's recorded mixer boundary (real feature-model "
                               f"inputs, real gate-row changes) of {args.bytes} bytes of {os.path.basename(args.text)}",
                   "n_inputs": topo.n_inputs, "mixers": f"{topo.l0}/{topo.l1}/{1 if topo.has_final else 0}",
                   "streams_per_gpu": S, "bits_per_stream_per_step": T},
        "roofline": {"bound": "hbm", "achieved": bytes_per_bit * S * T / (avg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": bytes_per_bit * S * T / (avg * 1e-3) / 1e9 / 8000.0, "traffic": None,
                     "kernel": "gmx_stock_kernel", "kernel_ms_avg": avg, "kernel_ms_min": each[0],
                     "kernel_ms_median": each[len(each) // 2], "kernel_ms_max": each[-1],
                     "algorithmic_bytes_per_bit": bytes_per_bit, "bytes_per_launch": bytes_per_bit * S * T,
                     "build": g.L.gmx_build_info().decode()},
        "workload": f"{S} streams replaying the reference Predictor's recorded mixer boundary of {args.bytes} bytes of "
                    f"{os.path.basename(args.text)} from {S} byte offsets, {T} bits per launch, active masks on",
        "bits_per_s": S * T / (avg * 1e-3), "kernel_ms_avg": avg, "kernel_ms_all": [round(x, 4) for x in ms],
        "us_per_bit_per_stream": avg * 1e3 / T, "streams": S, "bits_per_launch": T, "launches": n_win,
        "active_inputs_avg": float(tr["act"].sum(1).mean()),
        "gate_rows_changed_per_bit_avg": float(changed.sum(1).mean()),
        "gate_rows_changed_layer0_per_bit_avg": float(changed[:, :24].sum(1).mean()),
        "layer0_rows_changed_per_bit_histogram": {str(k): int(v) for k, v in enumerate(hist) if v},
        "staged": args.staged,
        "stream0_first_window_equals_reference": True,
        "build": g.L.gmx_build_info().decode(),
    }
    if args.prof:
        g.L.gmx_stock_prof_read(prof, 0)
        names = ["prefetch issue", "mask + skip", "forward", "logistic + stores", "learn scalars", "update", "commit wait",
                 "evict / adopt / loop"]
        out["phase_cycles_per_bit"] = {n: round(prof[k] / (n_win * T)) for k, n in enumerate(names)}
    for b in batches:
        b.close()
    g.close()
    return out


if __name__ == "__main__":
    main()
