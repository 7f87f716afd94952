#!/usr/bin/env python3
"""Auxiliary bench: the device-resident slice of the ensemble end to end -- LSTM byte model ->
41 Indirect models -> 33 mixers, every hand-over in HBM -- S streams x
N bytes per step.  Per stream ~0.97 GB of state (174 MB mixers, 790 MB Indirect tables, 7 MB LSTM),
so a GPU holds ~280 streams.
  python scripts/bench_pipeline.py [--streams S --bytes N --steps K]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--bytes", type=int, default=200)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lstm-cus", type=int, default=16,
                    help="CUs per XCD (of 32) the LSTM kernels get; the mixers and Indirect models get the rest")
    ap.add_argument("--indirect-all", action="store_true", help="the Indirect models keep all CUs")
    ap.add_argument("--mask-layout", default="xcd", help="xcd: CU i of XCD x is mask bit 32x+i; flat: the first CUs")
    args = ap.parse_args()
    import gmix_amd
    import goldenlib
    from gmix_amd import synth, topology
    _, z = goldenlib.load("ind_tiny_dense")
    models = topology.stock_indirect()
    K, S, NB = len(models), args.streams, args.bytes
    T = 8 * NB
    slots = [(8 + 2 * i, 9 + 2 * i) for i in range(K)]
    lg = gmix_amd.LstmGroup(S)
    ig = gmix_amd.IndirectGroup(models, z["ns_next"], z["rm_next"], S, slots=slots)
    mg = gmix_amd.MixerGroup(topology.stock(90), S)
    if args.lstm_cus:
        n = args.lstm_cus
        if args.mask_layout == "xcd":
            lm = [(1 << n) - 1] * 8
        else:
            bits = (1 << (8 * n)) - 1
            lm = [(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
        rest = [(~w) & 0xFFFFFFFF for w in lm]
        lg.set_cu_mask(lm)          # the LSTM on `n` CUs of every XCD ...
        ig.set_cu_mask(None if args.indirect_all else rest)  # ... the Indirect models and the mixers on the others
        mg.set_cu_mask(rest)
    # two sets of downstream records: the LSTM works on step k+1 while the mixers still read step k's
    lb = gmix_amd.LstmBatch(lg, NB)
    ibs = [gmix_amd.IndirectBatch(ig, T) for _ in range(2)]
    mbs = [gmix_amd.Batch(mg, T, outputs=False, mask=True) for _ in range(2)]
    w = synth.lstm_initial_weights()
    ppm, data = synth.lstm_records(NB, seed=1, mask=63)
    rng = np.random.default_rng(0)
    for s in range(S):
        lg.set_weights(w, stream=s)
        lb.ppm[s] = ppm
        lb.bytes[s] = np.roll(data, int(rng.integers(0, NB)))
    for i in range(2):
        ibs[i].fill_synthetic(T, seed=3 + i, restart=True, ctx_mod=(300, 0, 70000, 5))   # contexts, bit contexts, bits
        mbs[i].fill_synthetic(T, seed=5 + i, restart=True, ctx_mode=2, zero_mod=12)      # the 7 other inputs, mixer contexts
    lb.upload(NB)
    lg.sync(); ig.sync(); mg.sync()

    def step(k):
        ib, mb = ibs[k & 1], mbs[k & 1]
        lg.run(lb, NB, learn=True)
        lg.feed(lb, NB, mixer_batch=mb, slot=1, mixer_ctx_col=22, ind_batch=ib, ind_ctx_col=16)
        ig.run(ib, T, learn=True, into=mb)
        mg.run(mb, T, learn=True)

    for k in range(args.warmup):
        step(k)
    lg.sync(); ig.sync(); mg.sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    lg.sync(); ig.sync(); mg.sync()
    el = time.perf_counter() - t0
    print(json.dumps({
        "metric": "device-resident ensemble slice bits/sec (LSTM -> 41 Indirect -> 33 mixers, forward+update)",
        "value": S * T * args.steps / el, "unit": "bits/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "data": "synthetic",
        "config": {"streams": S, "bytes_per_stream_per_step": NB, "lstm_cus_per_xcd": args.lstm_cus,
                   "state_bytes_per_stream": lg.bank_bytes + ig.bank_bytes + mg.bank_bytes}}))


if __name__ == "__main__":
    main()
