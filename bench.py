#!/usr/bin/env python3
"""bench.py -- mixer bits/s on MI355X.

The headline line is BASELINE.json configs[1]: synthetic 256-input 1-layer mixer, random logits,
forward + update, 10^8 bits by default.  A "step" is one pass of the hot path (Predict + Perceive +
Learn for every bit) over one batch of synthetic records: S independent streams x T bits, records
already resident in HBM (generated on the device before the timed region).  Each stream owns a
dense 2^16-row x 256-weight gate table (64 MiB), so S streams use S x 64.5 MiB of HBM.

  python bench.py [--gpus N --steps K --warmup W] [--streams S --bits T --config single|synth3|stock]

The same JSON line carries, under "also", the other shapes the north_star names, each timed the same
way (barrier, K steps, max over ranks) with its own roofline and one-core reference figure:
  synth3       256 inputs x 24/8/1 mixers (configs[2]'s shape), new gate rows every bit
  stock_held   the reference's own 90 inputs x 24/8/1, gate contexts held through a byte
  stock_real   the same with the row-change pattern of a real gmix run (the four bit-level contexts move every bit)
  stock_fresh  the same with every gate context new every bit (worst case)
  stock_S1     ONE stream of the reference's shape: what a single compressor sees
  single_S1    ONE stream of configs[1]'s shape (the 256-step dependent sum: latency, not bandwidth)
  real_trace   (one GPU) gmx_stock_kernel on the reference Predictor's RECORDED mixer boundary (real inputs, real
               gate-row changes; the recording is made by oracle/_ref/ref_trace outside the timed region) --
               scripts/bench_real_trace.py
  indirect     (one GPU) the 41 Indirect models in front of the mixers, 256 streams -- scripts/bench_indirect.py
  lstm         (one GPU) the LSTM byte model, 1024 streams, bytes/s -- scripts/bench_lstm.py
  e2e_S1 / e2e_S1_mixers / e2e_S64   (one GPU) whole files through the run-ahead compressor: the reference's feature
               models and coder on the host cores, LSTM + Indirect models + mixers (or the mixers alone) on the device in
               a ring of four batches; 1 file and 64 files side by side -- scripts/bench_e2e.py
  e2e_decode   (one GPU) S files restored side by side: S of the reference's Decoders in lock step, one device step per
               coded bit for all of them -- scripts/bench_e2e.py measure_decode()
The e2e_* values are COLD (whole process, exec to exit, Predictor construction included) like the reference CLI they
are set against.  "fracs" (top level, before "also") repeats every kernel's roofline fraction in one short object.
(--no-also leaves them out; --config X makes X the headline workload for profiling.)

N > 1: one rank per GPU over RCCL.  Under torch.distributed.run the ranks come from the environment;
started plainly (`python bench.py --gpus N`, WORLD_SIZE unset) this process touches neither torch nor
HIP and starts torch.distributed.run itself as a child, relaying rank 0's line and the exit code.
Streams shard across ranks with no data-path collective (weak scaling: S streams per GPU); the only
communication is the barrier, a MAX all-reduce of the elapsed time and a SUM of the stream counts.

Prints ONE JSON line (rank 0) with the contract's fields plus "roofline", "cpu_baseline", "also".
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (topology, ctx_mode, default streams per GPU, default bits per stream per step, target total bits,
    #        one-core sample bits, description)
    "single": ("single", 0, 4096, 1024, 100_000_000, 20_000_000,
               "configs[1]: synthetic 256-input 1-layer mixer (1 mixer, 2^16-row gate table), random logits, forward+update"),
    "synth3": ("synth3", 0, 2048, 512, 8_000_000, 300_000,
               "synthetic 256-input 3-layer 24/8/1 bank (2^12-row layer-0 tables), new gate rows every bit, forward+update"),
    "stock": ("stock", 0, 1024, 1024, 8_000_000, 400_000,
              "stock 24/8/1 topology of Predictor::AddMixers, 90 inputs, synthetic records, forward+update"),
    "stock_held": ("stock", 2, 1024, 1024, 16_000_000, 800_000,
                   "stock 24/8/1 topology of Predictor::AddMixers, 90 inputs, synthetic records, gate contexts redrawn every 8th bit, forward+update"),
    "stock_fresh": ("stock", 0, 1024, 1024, 8_000_000, 400_000,
                    "stock 24/8/1 topology of Predictor::AddMixers, 90 inputs, synthetic records, every gate context new every bit, forward+update"),
    "stock_real": ("stock", 4, 1024, 1024, 8_000_000, 800_000,
                   "stock 24/8/1 topology of Predictor::AddMixers, 90 inputs, synthetic records with a real run's "
                   "row-change pattern: all 33 gate contexts redrawn at byte boundaries, the four bit-level ones "
                   "(2 layer-0, 2 layer-1) every bit, forward+update"),
    "stock_S1": ("stock", 2, 1, 8192, 32_768, 800_000,
                 "ONE stream of the stock 24/8/1 topology (90 inputs), gate contexts redrawn every 8th bit, forward+update"),
    "single_S1": ("single", 0, 1, 8192, 65_536, 20_000_000,
                  "configs[1] for ONE stream: synthetic 256-input 1-layer mixer, random logits, forward+update"),
}
ALSO = ("synth3", "stock_held", "stock_real", "stock_fresh", "stock_S1", "single_S1")


def aux_bench(script):
    """scripts/<script> as a module (its measure() returns one result object in this file's conventions)."""
    import importlib.util
    path = os.path.join(ROOT, "scripts", script)
    spec = importlib.util.spec_from_file_location(script[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_topology(kind):
    from gmix_amd import topology
    if kind == "single":
        return topology.single(256, 1 << 16, 0.005)
    if kind == "synth3":
        return topology.synth3(256, table0=1 << 12)
    if kind == "stock":
        return topology.stock(90)
    raise SystemExit(f"unknown topology {kind}")


def host_cpu():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(topo, sample_bits, ctx_mode=0, ctx_mod=1):
    """Reference Mixer (oracle/_ref, kind 'reference') or the C restatement (kind 'port'),
    single thread, on a bounded sample of the same workload."""
    spec = ",".join(f"{l}:{t}:{lr!r}" for l, t, lr in topo.mixers)
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_mixer_bench_fast")
    ncpu = os.cpu_count()
    cpu = host_cpu()
    if os.path.exists(exe):
        try:
            out = subprocess.run([exe, "--n", str(topo.n_inputs), "--topo", spec, "--bits", str(sample_bits),
                                  "--ctx-mode", str(ctx_mode), "--ctx-mod", str(ctx_mod)],
                                 capture_output=True, text=True, timeout=600, check=True).stdout
            r = json.loads(out.strip().splitlines()[-1])
            return {"value": r["bits_per_s"], "unit": "bits/s", "cores": 1, "kind": "reference",
                    "sample": f"{sample_bits} bits of the same synthetic stream through the reference's own "
                              f"Mixer::Predict+Learn (-Ofast -march=x86-64-v3: the makefile's flags with a portable "
                              f"-march), 1 thread of {ncpu} host cores, {cpu}"}
        except Exception as e:  # SIGILL on an older host, missing binary ...: say so and use the port
            sys.stderr.write(f"[bench] reference baseline failed ({e}); timing the C restatement instead\n")
    from oracle import gmxo
    pred, act, ctx, bits = gmxo.synth(topo.n_inputs, topo.n_mixers, sample_bits, ctx_mode=ctx_mode, ctx_mod=ctx_mod)
    b = gmxo.Bank(topo.n_inputs, topo.skip, topo.mixers)
    t0 = time.perf_counter()
    b.run(pred, act, ctx, bits, want_all=False)
    dt = time.perf_counter() - t0
    return {"value": sample_bits / dt, "unit": "bits/s", "cores": 1, "kind": "port",
            "sample": f"{sample_bits} bits of the same synthetic stream through oracle/gmx_oracle.c, 1 thread of "
                      f"{ncpu} host cores, {cpu}"}


def kernel_of(topo, kind, stock_pairs):
    if topo.n_mixers == 1:
        return "gmx_single_kernel"
    if kind == "stock":
        return "gmx_wide_kernel" if stock_pairs else "gmx_stock_kernel"
    return "gmx_wide_kernel" if kind == "synth3" else "gmx_bank_kernel"


def kernels_of(build):
    """The hash of the mixer kernels' sources inside gmx_build_info() (the whole string if there is none)."""
    b = build or ""
    return b.split("kernels ")[-1] if "kernels " in b else b


def pmc_traffic(kernel, S, T, ctx_mode, build):
    """HBM traffic per launch from the committed rocprofv3 PMC summary of the same launch shape (the
    counters need separate profiled passes, scripts/gpu_profile.sh; they cannot run inside the timed
    process).  Marked stale when the summary was taken on another build of the library."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
        try:
            pm = json.load(open(f))
        except Exception:
            continue
        if (pm.get("streams"), pm.get("bits_per_stream")) != (S, T) or kernel not in pm.get("kernel", ""):
            continue
        if pm.get("ctx_mode", ctx_mode) != ctx_mode:
            continue
        return {"traffic": pm["traffic_bytes_per_launch"], "traffic_source": os.path.relpath(f, ROOT),
                "traffic_stale": kernels_of(pm.get("build")) != kernels_of(build)}
    return {"traffic": None}


class Comm:
    """The three collectives of the bench (RCCL when torch.distributed is up, identity otherwise)."""

    def __init__(self, dist, n_gpus, rank, pinned=None):
        self.dist, self.n_gpus, self.rank = dist, n_gpus, rank
        self.pinned = pinned  # how many cpus of the GPU's NUMA node this rank was kept on (None: not pinned)

    def barrier(self):
        if self.dist is not None:
            import torch
            self.dist.barrier()
            torch.cuda.synchronize()

    def max(self, v):
        from gmix_amd import shard
        return shard.max_over_ranks(v, self.dist, device="cuda") if self.dist is not None else float(v)

    def sum(self, v):
        from gmix_amd import shard
        return shard.sum_over_ranks(v, self.dist, device="cuda") if self.dist is not None else int(v)

    def gather(self, v):
        from gmix_amd import shard
        return shard.gather_floats(v, self.dist, device="cuda") if self.dist is not None else [float(v)]


def run_workload(name, comm, local_rank, streams=None, bits=None, steps=None, warmup=2, ring_n=4,
                 ctx_mode=None, ctx_mod=1, stock_pairs=False, variant=0, want_cpu=True, cpu_sample_bits=None):
    """Time K steps of one workload on this rank's GPU; rank 0 gets the result dict."""
    import gmix_amd
    kind, mode0, S0, T0, total, sample0, workload = WORKLOADS[name]
    topo = make_topology(kind)
    ctx_mode = mode0 if ctx_mode is None else ctx_mode
    S = streams or S0
    T = bits or T0
    g = None
    while g is None:
        try:
            g = gmix_amd.MixerGroup(topo, S, device=local_rank)
        except gmix_amd.GmxError as e:
            if e.status != -2 or S <= 64:
                raise
            S = max(64, (S * 3 // 4) // 64 * 64)  # dense tables did not fit: fewer streams
    if steps is None:
        steps = max(1, -(-total // (S * T)))
    if variant:
        import ctypes
        g.L.gmx_debug_single_variant.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert g.L.gmx_debug_single_variant(g.h, variant) == 0
    if stock_pairs:
        import ctypes
        g.L.gmx_debug_stock_pairs.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert g.L.gmx_debug_stock_pairs(g.h, 1) == 0
    ring = [gmix_amd.Batch(g, T, outputs=False, mask=False) for _ in range(ring_n)]
    for i, b in enumerate(ring):
        b.fill_synthetic(T, seed=0x9E3779B97F4A7C15 + 1000003 * (comm.rank * ring_n + i), restart=True,
                         ctx_mode=ctx_mode, ctx_mod=ctx_mod)
    g.sync()
    for k in range(warmup):
        g.run(ring[k % len(ring)], T, learn=True)
    g.sync()
    comm.barrier()
    # K launches queued without a host synchronisation in between (the host prepares launch k+1
    # while launch k runs); HIP events on the group's stream bracket them for the roofline
    t0 = time.perf_counter()
    g.timer_start()
    for k in range(steps):
        g.run(ring[k % len(ring)], T, learn=True)
    gpu_ms = g.timer_stop()
    g.sync()
    comm.barrier()
    mine = time.perf_counter() - t0
    elapsed = comm.max(mine)
    S_all = comm.sum(S)  # a rank with less free HBM runs fewer streams
    rank_rates = comm.gather(S * T * steps / mine)  # every rank's own bits/s (its clock ends at the common barrier)
    # the same launches once more, each bracketed by its own HIP events (outside the timed region: bracketing
    # synchronises): how much a step varies
    each = sorted(g.run(ring[k % len(ring)], T, learn=True, timed=True) for k in range(min(steps, 8)))
    bank_bytes = g.bank_bytes
    build = g.L.gmx_build_info().decode()
    for b in ring:
        b.close()
    g.close()
    if comm.rank != 0:
        return None
    avg_ms = gpu_ms / steps
    bits_per_step = S_all * T
    # rows move when a gate context changes: every bit (ctx-mode 0/1) or every 8th bit (2/3)
    row_bytes = 8 * sum(topo.weight_sizes())
    hold = 8 if ctx_mode >= 2 else 1
    moved = row_bytes // hold
    if ctx_mode >= 4:  # ... plus the four bit-level rows on the other seven bits of a byte (oracle/gmx_synth.h)
        ws = topo.weight_sizes()
        moved += 7 * 8 * sum(ws[j] for j in (2, 11, 26, 29) if j < len(ws)) // 8
    bytes_per_bit = moved + topo.bytes_per_bit() - row_bytes
    bytes_per_launch = bytes_per_bit * S * T
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    kernel = kernel_of(topo, kind, stock_pairs)
    out = {
        "value": bits_per_step * steps / elapsed, "unit": "bits/s", "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "config": {"workload": workload + (", gate contexts redrawn every 8th bit"
                                           if hold == 8 and "8th" not in workload else ""),
                   "n_inputs": topo.n_inputs, "mixers": f"{topo.l0}/{topo.l1}/{1 if topo.has_final else 0}",
                   "streams_per_gpu": S, "bits_per_stream_per_step": T, "bits_per_step": bits_per_step,
                   "total_bits": bits_per_step * steps, "bank_bytes_per_stream": bank_bytes,
                   "parallelism": f"streams sharded over {comm.n_gpus} GPU(s), no collective on the data path"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": kernel, "kernel_ms_avg": avg_ms,
                     "kernel_ms_min": each[0], "kernel_ms_median": each[len(each) // 2], "kernel_ms_max": each[-1],
                     "algorithmic_bytes_per_bit": bytes_per_bit, "bytes_per_launch": bytes_per_launch,
                     "build": build},
    }
    try:
        out["roofline"].update(pmc_traffic(kernel, S, T, ctx_mode, build))
    except Exception as e:
        sys.stderr.write(f"[bench] no PMC summary: {e}\n")
    if comm.n_gpus > 1:
        out["per_rank"] = {"bits_per_s": rank_rates, "min": min(rank_rates), "max": max(rank_rates),
                           "numa_pinned_cpus": comm.pinned}
    if want_cpu and comm.n_gpus == 1:
        out["cpu_baseline"] = cpu_baseline(topo, cpu_sample_bits or sample0, ctx_mode, ctx_mod)
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: this parent has touched neither torch
    nor HIP; it starts torch.distributed.run as a CHILD process (never an exec) and hands on its
    output and exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=None,
                    help="independent streams per GPU (each owns a dense 64.5 MiB gate table: 3072 = 194 GiB of HBM)")
    ap.add_argument("--bits", type=int, default=None, help="bits per stream per step")
    ap.add_argument("--config", default="single", choices=sorted(WORKLOADS))
    ap.add_argument("--ring", type=int, default=2, help="distinct record batches cycled through")
    ap.add_argument("--variant", type=int, default=0,
                    help="tuning: lanes per stream of the single-mixer kernel (0 = library default)")
    ap.add_argument("--ctx-mode", type=int, default=None,
                    help="0: fresh 32-bit gate contexts every bit (BASELINE configs[1]); 2/3: contexts held "
                         "for 8 bits like byte-boundary contexts (oracle/gmx_synth.h)")
    ap.add_argument("--ctx-mod", type=int, default=1)
    ap.add_argument("--stock-pairs", action="store_true",
                    help="--config stock*: the lane-pair kernel (gmx_wide.hip) instead of the generated streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="only the headline workload")
    ap.add_argument("--only-also", default="", help="comma-separated: of the also{} entries, only these")
    ap.add_argument("--e2e-bytes", type=int, default=30000, help="bytes of the one file of also.e2e_S1 / e2e_S1_mixers")
    ap.add_argument("--e2e-many-bytes", type=int, default=100000, help="bytes per file of also.e2e_S64")
    ap.add_argument("--also-streams-div", type=int, default=1,
                    help="divide the stream counts of the also{} kernel entries by this (a test run: dense banks of a "
                         "thousand streams take seconds to allocate and clear)")
    ap.add_argument("--decode-bytes", type=int, default=5000, help="bytes per file of also.e2e_decode")
    ap.add_argument("--decode-streams", type=int, default=256, help="files of also.e2e_decode")
    ap.add_argument("--rehearse-cpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-sample-bits", type=int, default=None,
                    help="bits of the same stream for the one-core reference (default: about 5-15 s of CPU work)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("GMX_BENCH_FORCE_DIST")):
        sys.exit(spawn_ranks(args.gpus))  # GMX_BENCH_FORCE_DIST rehearses the N > 1 path on one GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.stderr.write(f"[bench] --gpus {args.gpus} but WORLD_SIZE {world}: refusing to report a figure for a "
                         f"GPU count that is not the one that ran\n")
        sys.exit(2)
    dist = None
    if args.rehearse_cpu:
        # the launch plumbing alone, on CPU: gloo, no GPU work, no figure (tests/test_bench_spawn.py)
        import torch.distributed as dist
        dist.init_process_group("gloo")
        from gmix_amd import shard
        total = shard.sum_over_ranks(1, dist)
        slowest = shard.max_over_ranks(float(rank), dist)
        each = shard.gather_floats(100.0 + rank, dist)       # the per-rank figures' collective
        lowest = shard.min_over_ranks(100.0 + rank, dist)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"rehearsal": True, "n_gpus": world, "ranks_counted": total, "max_rank": slowest,
                              "per_rank": each, "min": lowest, "value": None}), flush=True)
        dist.destroy_process_group()
        return
    if world > 1 or os.environ.get("GMX_BENCH_FORCE_DIST"):
        # torch first: libgmxmix.so then binds to the HIP runtime torch has already loaded
        # (same soname), so RCCL and the mixer kernels share one runtime in this process.
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if local_rank >= torch.cuda.device_count():
            sys.stderr.write(f"[bench] rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible)\n")
            sys.exit(3)
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    pinned = None
    try:
        # the rank's host side (decay tables, record staging) on the cores next to its GPU
        import ctypes
        import gmix_amd
        from gmix_amd import _lib, shard
        buf = ctypes.create_string_buffer(64)
        if _lib.lib().gmx_device_pci_bus_id(local_rank, buf, 64) == 0:
            cpus = shard.pin_to_gpu_numa_node(buf.value.decode())
            pinned = len(cpus) if cpus else None
    except Exception as e:
        sys.stderr.write(f"[bench] rank {rank}: not pinned to its GPU's NUMA node ({e})\n")
    comm = Comm(dist, world, rank, pinned)

    t_sections, t_mark = {}, time.perf_counter()

    def section(name):   # wall seconds of the bench's own sections (rank 0's clock): where a default run's minutes go
        nonlocal t_mark
        now = time.perf_counter()
        t_sections[name] = round(t_sections.get(name, 0.0) + now - t_mark, 1)
        t_mark = now

    section("setup")
    head = run_workload(args.config, comm, local_rank, streams=args.streams, bits=args.bits, steps=args.steps,
                        warmup=args.warmup, ring_n=args.ring, ctx_mode=args.ctx_mode, ctx_mod=args.ctx_mod,
                        stock_pairs=args.stock_pairs, variant=args.variant, want_cpu=not args.no_cpu_baseline,
                        cpu_sample_bits=args.cpu_sample_bits)
    section(args.config)
    also = {}
    only = {x for x in args.only_also.split(",") if x}
    if not args.no_also:
        for name in ALSO:
            if name == args.config or (only and name not in only):
                continue
            try:
                # warm-up launches: the library's staging slots (decay tables) are allocated by the first of them
                div = max(1, args.also_streams_div)
                r = run_workload(name, comm, local_rank, warmup=2, ring_n=2,
                                 streams=max(1, WORKLOADS[name][2] // div) if div > 1 else None,
                                 want_cpu=not args.no_cpu_baseline and name not in ("stock_S1", "single_S1"))
            except Exception as e:  # a sub-result must never cost the headline line
                if dist is not None:
                    raise  # ... except across ranks, where a lone failure would leave the others in a barrier
                r = {"error": f"{type(e).__name__}: {e}"}
            if rank == 0:
                also[name] = r
            section(name)
        if rank == 0 and "cpu_baseline" in also.get("stock_held", {}) and "error" not in also.get("stock_S1", {"error": 1}):
            # one stream of the same workload: the one-core reference figure is the same measurement
            also["stock_S1"]["cpu_baseline"] = dict(also["stock_held"]["cpu_baseline"])
        if rank == 0 and "cpu_baseline" in head and args.config == "single" and "error" not in also.get("single_S1", {"error": 1}):
            also["single_S1"]["cpu_baseline"] = dict(head["cpu_baseline"])   # (likewise: the headline's own, 15 s of one core)
        if dist is None:
            # the producers in front of the mixers, timed by their own scripts' code:
            # the 41 Indirect models (bits/s) and the LSTM byte model (bytes/s), each with roofline + cpu_baseline
            # ... and whole files end to end: the reference's feature models and coder on the host running ahead of
            # the device-side models (scripts/bench_e2e.py; every output compared with the stock build's)
            for name, script, kw in (("real_trace", "bench_real_trace.py", {}),
                                     ("indirect", "bench_indirect.py", {}), ("lstm", "bench_lstm.py", {}),
                                     ("e2e_S1", "bench_e2e.py", {"streams": 1, "variant": "chain"}),
                                     ("e2e_S1_mixers", "bench_e2e.py", {"streams": 1, "variant": "mixers"}),
                                     ("e2e_S64", "bench_e2e.py", {"streams": 64, "variant": "chain"}),
                                     ("e2e_decode", "bench_e2e.py", {"decode": True})):
                if only and name not in only:
                    continue
                try:
                    mod = aux_bench(script)
                    if kw.get("decode"):
                        if not hasattr(mod, "measure_decode"):
                            continue
                        also[name] = mod.measure_decode(streams=args.decode_streams, n_bytes=args.decode_bytes)
                    elif script == "bench_e2e.py":
                        also[name] = mod.measure(n_bytes=args.e2e_many_bytes if kw["streams"] > 1 else args.e2e_bytes, **kw)
                    else:
                        also[name] = mod.measure(**kw)
                    if args.no_cpu_baseline:
                        also[name].pop("cpu_baseline", None)
                    rf = also[name].get("roofline")
                    if rf and rf.get("traffic") is None and name in ("indirect", "lstm"):   # the committed PMC summary of the same launch shape
                        c = also[name]["config"]
                        rf.update(pmc_traffic(rf["kernel"], c.get("streams", c.get("streams_per_gpu")), c.get("bits_per_stream_per_step",
                                              c.get("bytes_per_stream_per_step")), 0, rf.get("build")))
                except Exception as e:
                    also[name] = {"error": f"{type(e).__name__}: {e}"}
                section(name)
        elif "e2e_S64" in only:
            # (only when asked for: --only-also e2e_S64.)  Whole files on every GPU at once (weak scaling: a process per
            # GPU, no exchange): every rank compresses its own files on its own device -- as many as the host's CPU
            # quota and memory, shared by all ranks, carry (scripts/bench_e2e.py files_that_fit).  The ranks' clocks start
            # together, behind a barrier, and `value` = all bits / the slowest rank's whole-process time.  A rank that
            # fails reports -1 and still takes part in the collectives below.
            mod = aux_bench("bench_e2e.py")
            n_files = mod.files_that_fit(64, world)
            mine = {"value": -1.0, "seconds": -1.0, "bits": 0.0}
            try:
                r = mod.measure(streams=n_files, n_bytes=args.e2e_many_bytes, variant="chain", device=local_rank,
                                verify=(rank == 0), cpu=False, before=comm.barrier)
                mine = {"value": r["value"], "seconds": r["seconds"], "bits": 8.0 * n_files * args.e2e_many_bytes}
            except Exception as e:
                r = {"error": f"{type(e).__name__}: {e}"}
            comm.barrier()
            rates, secs, bits = comm.gather(mine["value"]), comm.gather(mine["seconds"]), comm.gather(mine["bits"])
            if rank == 0:
                ok = all(v > 0 for v in rates)
                e = dict(r)
                e.update({"n_gpus": world, "files_per_gpu": n_files, "per_rank": {"bits_per_s": rates, "seconds": secs},
                          "value": (sum(bits) / max(secs)) if ok else None, "scaling": "weak"})
                if not ok:
                    e["error"] = "a rank failed: " + str(rates)
                also["e2e_S64"] = e
    if rank == 0:
        out = {"metric": "mixer bits/sec (synthetic 256-input mixer streams, forward+update)",
               "value": head["value"], "unit": "bits/s", "n_gpus": world, "steps": head["steps"],
               "warmup": head["warmup"], "ms_per_step": head["ms_per_step"], "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": head["config"], "roofline": head["roofline"]}
        if "per_rank" in head:
            out["per_rank"] = head["per_rank"]
        if "cpu_baseline" in head:
            out["cpu_baseline"] = head["cpu_baseline"]
        # every kernel's fraction of its roofline in one short object AHEAD of the long sub-results (a truncated
        # tail still shows it); whole-file entries give bits/s, cold, and their ratio to the CPU's own CLI
        fracs = {args.config: round(head["roofline"]["frac"], 4)}
        for name, r in also.items():
            if isinstance(r, dict) and isinstance(r.get("roofline"), dict) and r["roofline"].get("frac") is not None:
                fracs[name] = round(r["roofline"]["frac"], 4)
        out["fracs"] = fracs
        e2e = {name: {"bits_per_s_cold": r.get("value"), "vs_cpu": r.get("vs_cpu"), "identical": r.get("identical_to_stock")}
               for name, r in also.items() if name.startswith("e2e_") and isinstance(r, dict) and "error" not in r}
        if also:
            out["also"] = also
        # ... and once more as the line's LAST object: a reader who keeps only the end of a long line (the driver's
        # record holds the last 2 000 characters) still sees every fraction and the whole-file figures
        out["tail_summary"] = {"fracs": fracs, "e2e": e2e, "bench_seconds": t_sections}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
