#!/usr/bin/env python3
"""bench.py -- mixer bits/s on MI355X, BASELINE.json configs[1]:
synthetic 256-input 1-layer mixer, random logits, forward + update, 10^8 bits by default.

A "step" is one pass of the hot path (Predict + Perceive + Learn for every bit) over one batch
of synthetic records: S independent streams x T bits, records already resident in HBM
(generated on the device before the timed region, BASELINE.json "synthetic").  Each stream
owns a dense 2^16-row x 256-weight gate table (64 MiB), so S streams use S x 64.5 MiB of HBM.

  python bench.py [--gpus N --steps K --warmup W] [--streams S --bits T --config single|synth3|stock]

N > 1: launched by torch.distributed.run, one rank per GPU; streams shard across ranks with
no data-path collective (weak scaling: S streams per GPU); the only communication is the
barrier and a MAX all-reduce of the elapsed time over RCCL.

Prints ONE JSON line (rank 0) with the contract's fields plus "roofline" and "cpu_baseline".
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def make_topology(name):
    from gmix_amd import topology
    if name == "single":
        return topology.single(256, 1 << 16, 0.005), "configs[1]: synthetic 256-input 1-layer mixer (1 mixer, 2^16-row gate table), random logits, forward+update"
    if name == "synth3":
        return topology.synth3(256, table0=1 << 12), "synthetic 256-input 3-layer 24/8/1 bank (2^12-row layer-0 tables), forward+update"
    if name == "stock":
        return topology.stock(90), "stock 24/8/1 topology of Predictor::AddMixers, 90 inputs, synthetic records, forward+update"
    raise SystemExit(f"unknown --config {name}")


def cpu_baseline(topo, sample_bits, ctx_mode=0, ctx_mod=1):
    """Reference Mixer (oracle/_ref, kind 'reference') or the C restatement (kind 'port'),
    single thread, on a bounded sample of the same workload."""
    spec = ",".join(f"{l}:{t}:{lr!r}" for l, t, lr in topo.mixers)
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_mixer_bench_fast")
    ncpu = os.cpu_count()
    if os.path.exists(exe):
        try:
            out = subprocess.run([exe, "--n", str(topo.n_inputs), "--topo", spec, "--bits", str(sample_bits),
                                  "--ctx-mode", str(ctx_mode), "--ctx-mod", str(ctx_mod)],
                                 capture_output=True, text=True, timeout=600, check=True).stdout
            r = json.loads(out.strip().splitlines()[-1])
            return {"value": r["bits_per_s"], "unit": "bits/s", "cores": 1, "kind": "reference",
                    "sample": f"{sample_bits} bits of the same synthetic stream through the reference's own "
                              f"Mixer::Predict+Learn (makefile flags -Ofast -march=native), 1 thread of {ncpu} host cores"}
        except Exception as e:  # fall through to the port
            sys.stderr.write(f"[bench] reference baseline failed: {e}\n")
    from oracle import gmxo
    pred, act, ctx, bits = gmxo.synth(topo.n_inputs, topo.n_mixers, sample_bits, ctx_mode=ctx_mode, ctx_mod=ctx_mod)
    b = gmxo.Bank(topo.n_inputs, topo.skip, topo.mixers)
    t0 = time.perf_counter()
    b.run(pred, act, ctx, bits, want_all=False)
    dt = time.perf_counter() - t0
    return {"value": sample_bits / dt, "unit": "bits/s", "cores": 1, "kind": "port",
            "sample": f"{sample_bits} bits of the same synthetic stream through oracle/gmx_oracle.c, 1 thread of {ncpu} host cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=None,
                    help="independent streams per GPU (each owns a dense 64.5 MiB gate table: 3072 = 194 GiB of HBM)")
    ap.add_argument("--bits", type=int, default=512, help="bits per stream per step")
    ap.add_argument("--config", default="single")
    ap.add_argument("--ring", type=int, default=4, help="distinct record batches cycled through")
    ap.add_argument("--variant", type=int, default=0,
                    help="tuning: lanes per stream of the single-mixer kernel (0 = library default)")
    ap.add_argument("--ctx-mode", type=int, default=0,
                    help="0: fresh 32-bit gate contexts every bit (BASELINE configs[1]); 2/3: contexts held "
                         "for 8 bits like byte-boundary contexts (oracle/gmx_synth.h)")
    ap.add_argument("--ctx-mod", type=int, default=1)
    ap.add_argument("--stock-pairs", action="store_true",
                    help="--config stock: the lane-pair kernel (gmx_wide.hip) instead of the generated streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-bits", type=int, default=20_000_000,
                    help="bits of the same stream for the one-core reference (about 15 s of CPU work)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("GMX_BENCH_FORCE_DIST"):  # the env var rehearses the N>1 path on one GPU
        # torch first: libgmxmix.so then binds to the HIP runtime torch has already loaded
        # (same soname), so RCCL and the mixer kernels share one runtime in this process.
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        sys.stderr.write(f"[bench] --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE\n")
    n_gpus = world

    import gmix_amd
    topo, workload = make_topology(args.config)
    if args.streams is None:  # what fills the chip for the shape: one wave per SIMD (4 one-mixer streams per wave)
        args.streams = 4096 if args.config == "single" else 1024
    S, T = args.streams, args.bits
    steps = args.steps if args.steps is not None else max(1, -(-100_000_000 // (S * T)))  # 10^8 bits

    g = None
    while g is None:
        try:
            g = gmix_amd.MixerGroup(topo, S, device=local_rank)
        except gmix_amd.GmxError as e:
            if e.status != -2 or S <= 64:
                raise
            S = max(64, (S * 3 // 4) // 64 * 64)  # dense tables did not fit: fewer streams
    if args.variant:
        import ctypes
        g.L.gmx_debug_single_variant.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert g.L.gmx_debug_single_variant(g.h, args.variant) == 0
    if args.stock_pairs:
        import ctypes
        g.L.gmx_debug_stock_pairs.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert g.L.gmx_debug_stock_pairs(g.h, 1) == 0
    ring = [gmix_amd.Batch(g, T, outputs=False, mask=False) for _ in range(args.ring)]
    for i, b in enumerate(ring):
        b.fill_synthetic(T, seed=0x9E3779B97F4A7C15 + 1000003 * (rank * args.ring + i), restart=True,
                         ctx_mode=args.ctx_mode, ctx_mod=args.ctx_mod)
    g.sync()

    for k in range(args.warmup):
        g.run(ring[k % len(ring)], T, learn=True)
    g.sync()
    if dist is not None:
        import torch
        dist.barrier()
        torch.cuda.synchronize()
    # K launches queued without a host synchronisation in between (the host prepares launch k+1
    # while launch k runs); HIP events on the group's stream bracket them for the roofline
    t0 = time.perf_counter()
    g.timer_start()
    for k in range(steps):
        g.run(ring[k % len(ring)], T, learn=True)
    gpu_ms = g.timer_stop()
    g.sync()
    if dist is not None:
        import torch
        torch.cuda.synchronize()
        dist.barrier()
    elapsed = time.perf_counter() - t0
    S_all = S * n_gpus
    if dist is not None:
        from gmix_amd import shard
        elapsed = shard.max_over_ranks(elapsed, dist, device="cuda")
        S_all = shard.sum_over_ranks(S, dist, device="cuda")  # a rank with less free HBM runs fewer streams
    kernel_ms = [gpu_ms / steps]

    if rank == 0:
        bits_per_step = S_all * T
        value = bits_per_step * steps / elapsed
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        # rows move when a gate context changes: every bit (ctx-mode 0/1) or every 8th bit (2/3)
        row_bytes = 8 * sum(topo.weight_sizes())
        hold = 8 if args.ctx_mode >= 2 else 1
        bytes_per_bit = row_bytes // hold + topo.bytes_per_bit() - row_bytes
        bytes_per_launch = bytes_per_bit * S * T
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        out = {
            "metric": "mixer bits/sec (synthetic 256-input mixer streams, forward+update)",
            "value": value, "unit": "bits/s", "n_gpus": n_gpus, "steps": steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload + ("" if hold == 1 else ", gate contexts redrawn every 8th bit"),
                       "n_inputs": topo.n_inputs,
                       "mixers": f"{topo.l0}/{topo.l1}/{1 if topo.has_final else 0}",
                       "streams_per_gpu": S, "bits_per_stream_per_step": T,
                       "bits_per_step": bits_per_step, "total_bits": bits_per_step * steps,
                       "bank_bytes_per_stream": g.bank_bytes, "parallelism": f"streams sharded over {n_gpus} GPU(s), no collective on the data path"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": ("gmx_single_kernel" if topo.n_mixers == 1 else
                                    ("gmx_wide_kernel" if args.stock_pairs else "gmx_stock_kernel") if args.config == "stock" else ("gmx_wide_kernel" if args.config == "synth3" else "gmx_bank_kernel")),
                         "kernel_ms_avg": avg_ms,
                         "algorithmic_bytes_per_bit": bytes_per_bit,
                         "bytes_per_launch": bytes_per_launch},
        }
        # HBM traffic per launch comes from rocprofv3 PMC passes of this same command (they
        # cannot run inside the timed process); the committed summary is quoted when it was
        # taken on the same launch shape, otherwise the field stays null.
        try:
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
                pm = json.load(open(f))
                if (pm.get("streams"), pm.get("bits_per_stream")) == (S, T) and \
                        out["roofline"]["kernel"] in pm.get("kernel", ""):
                    out["roofline"]["traffic"] = pm["traffic_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = os.path.relpath(f, ROOT)
                    break
        except Exception as e:
            sys.stderr.write(f"[bench] no PMC summary: {e}\n")
        if n_gpus == 1 and not args.no_cpu_baseline:
            sample = args.cpu_sample_bits if topo.n_mixers == 1 else min(args.cpu_sample_bits, 1_500_000)
            out["cpu_baseline"] = cpu_baseline(topo, sample, args.ctx_mode, args.ctx_mod)
        print(json.dumps(out), flush=True)
    for b in ring:
        b.close()
    g.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
