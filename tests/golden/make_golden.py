#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the REAL reference (build container only).

  python tests/golden/make_golden.py            # all synthetic cases + the english.dic trace

Needs oracle/_ref/ (make -C oracle/ref_build full) and /root/reference.  Each fixture holds
inputs-by-seed and the reference's outputs: every mixer output and probability of the dumped
bits (as uint32 bit patterns), running checksums over all bits, the 3 x u64 short state and
the serialised mixer section (or its sha256 when large).  The reference never travels; these
files do.
"""
import hashlib
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from golden.cases import CASES, IND_CASES, LSTM_CASES  # noqa: E402
from oracle import gmxo  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
LONG_INLINE_LIMIT = 96 * 1024


def topo_spec(topo):
    return ",".join(f"{l}:{t}:{lr!r}" for l, t, lr in topo.mixers)


def run_case(name):
    mk, T, dump, kw = CASES[name]
    topo = mk()
    kw = dict(kw)
    args = [os.path.join(REF, "ref_mixer_harness"), "--n", str(topo.n_inputs), "--topo", topo_spec(topo),
            "--skip", ",".join(map(str, topo.skip)) if topo.skip else "none", "--bits", str(T),
            "--dump", str(dump)]
    for k, flag in (("seed", "--seed"), ("ctx_mode", "--ctx-mode"), ("ctx_mod", "--ctx-mod"),
                    ("zero_mod", "--zero-mod"), ("bit_mode", "--bit-mode"), ("nolearn_from", "--nolearn-from")):
        if k in kw:
            args += [flag, str(kw[k])]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "d.bin")
        subprocess.run(args + ["--out", out], check=True, stdout=subprocess.DEVNULL)
        d = gmxo.read_dump(out)
    long_b = d["long"]
    meta = dict(name=name, n=topo.n_inputs, mixers=topo.mixers, skip=topo.skip, T=T, dump=dump, synth=kw,
                h32=int(d["h32"]), acc=float(d["acc"]), h64=int(d["h64"]), long_len=len(long_b),
                long_sha256=hashlib.sha256(long_b).hexdigest(), short_hex=d["short"].hex(),
                source="oracle/_ref/ref_mixer_harness (reference Mixer, g++ -O2 strict)")
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), meta=json.dumps(meta),
        outs=d["outs"].view(np.uint32), p=d["p"].view(np.uint32), mem=d["mem"],
        long=np.frombuffer(long_b if len(long_b) <= LONG_INLINE_LIMIT else b"", np.uint8))
    print(f"{name}: T={T} dump={dump} h64={d['h64']:016x} long={len(long_b)}B")


def run_ind_case(name):
    mk, T, dump, kw = IND_CASES[name]
    models = mk()
    kw = dict(kw)
    args = [os.path.join(REF, "ref_indirect_harness"), "--models", ",".join(f"{t}:{lr!r}" for t, lr in models),
            "--bits", str(T), "--dump", str(dump), "--ctx-mod", ",".join(map(str, kw.get("ctx_mod", (0, 0, 0, 0))))]
    if "seed" in kw:
        args += ["--seed", str(kw["seed"])]
    if "nolearn_from" in kw:
        args += ["--nolearn-from", str(kw["nolearn_from"])]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "d.bin")
        subprocess.run(args + ["--out", out], check=True, stdout=subprocess.DEVNULL)
        d = gmxo.read_ind_dump(out)
    # the indirect section is the head of the reference's .long file; its length follows from
    # the format (long-term-memory.cpp:8-32), the rest belongs to sections that are empty here
    ob = gmxo.IndirectBank(models, d["ns_next"], d["rm_next"])
    n_ind = len(d["long"]) - 8
    ind_b = d["long"][:n_ind]
    meta = dict(name=name, models=models, T=T, dump=dump, synth=kw, h64=int(d["h64"]),
                usage=[int(u) for u in d["usage"]], long_len=len(ind_b),
                long_sha256=hashlib.sha256(ind_b).hexdigest(),
                source="oracle/_ref/ref_indirect_harness (reference Indirect, g++ -O2 strict)")
    del ob
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), meta=json.dumps(meta), ns_next=d["ns_next"], rm_next=d["rm_next"],
        pred=d["pred"].view(np.uint32), active=np.packbits(d["active"], axis=1, bitorder="little"),
        long=np.frombuffer(ind_b if len(ind_b) <= LONG_INLINE_LIMIT else b"", np.uint8))
    print(f"{name}: K={len(models)} T={T} dump={dump} h64={d['h64']:016x} long={len(ind_b)}B")


def run_lstm_case(name):
    n_bytes, dump, kw = LSTM_CASES[name]
    args = [os.path.join(REF, "ref_lstm_harness"), "--bytes", str(n_bytes), "--dump", str(dump),
            "--seed", str(kw.get("seed", 0)), "--mask", str(kw.get("mask", 255))]
    if "nolearn_from" in kw:
        args += ["--nolearn-from", str(kw["nolearn_from"])]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "d.bin")
        subprocess.run(args + ["--out", out], check=True, stdout=subprocess.DEVNULL)
        d = gmxo.read_lstm_dump(out)
    meta = dict(name=name, bytes=n_bytes, dump=dump, synth=kw, h64=int(d["h64"]),
                init_weights_hash=int(d["init_weights_hash"]), long_hash=int(d["long_hash"]),
                usage=int(d["usage"]), short_size=int(d["short_size"]), short_hash=int(d["short_hash"]),
                top=int(d["top"]), mid=int(d["mid"]), bot=int(d["bot"]),
                source="oracle/_ref/ref_lstm_harness (reference LstmModel after srand(0xDEADBEEF), g++ -O2 strict)")
    np.savez_compressed(os.path.join(HERE, name + ".npz"), meta=json.dumps(meta), pred=d["pred"].view(np.uint32),
                        active=d["active"], ctx=d["ctx"], probs=d["probs"].view(np.uint32))
    print(f"{name}: bytes={n_bytes} dump={dump} h64={d['h64']:016x} long={d['long_hash']:016x}")


def run_trace(n_bytes=300):
    """Whole reference Predictor over the first bytes of dictionary/english.dic."""
    src = "/root/reference/dictionary/english.dic"
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "t.bin")
        subprocess.run([os.path.join(REF, "ref_trace"), src, str(n_bytes), out, "0"], check=True,
                       stdout=subprocess.DEVNULL)
        b = open(out, "rb").read()
    magic, ver, n, M, L0, L1, nskip = struct.unpack_from("<7I", b, 0)
    assert magic == 0x54584D47 and ver == 1
    off = 28
    skip = list(struct.unpack_from(f"<{nskip}I", b, off))
    off += 4 * nskip
    T, = struct.unpack_from("<Q", b, off)
    off += 8
    mixers = []
    for _ in range(M):
        layer, table, lr, ws = struct.unpack_from("<iIfi", b, off)
        off += 16
        mixers.append((layer, table, float(np.float32(lr)), ws))
    rec = 4 * n + n + 4 * M + 1 + 4 * M + 4
    pred = np.zeros((T, n), np.float32)
    act = np.zeros((T, n), np.uint8)
    ctx = np.zeros((T, M), np.uint32)
    bits = np.zeros(T, np.uint8)
    outs = np.zeros((T, M), np.float32)
    p = np.zeros(T, np.float32)
    for t in range(T):
        o = off + t * rec
        pred[t] = np.frombuffer(b, np.float32, n, o); o += 4 * n
        act[t] = np.frombuffer(b, np.uint8, n, o); o += n
        ctx[t] = np.frombuffer(b, np.uint32, M, o); o += 4 * M
        bits[t] = b[o]; o += 1
        outs[t] = np.frombuffer(b, np.float32, M, o); o += 4 * M
        p[t] = np.frombuffer(b, np.float32, 1, o)[0]
    off += T * rec
    ns, = struct.unpack_from("<Q", b, off); off += 8
    short = b[off:off + ns]; off += ns
    nl, = struct.unpack_from("<Q", b, off); off += 8
    long_b = b[off:off + nl]
    meta = dict(name="trace_english", n=n, mixers=[m[:3] for m in mixers], weight_sizes=[m[3] for m in mixers],
                skip=skip, T=int(T), long_len=len(long_b), long_sha256=hashlib.sha256(long_b).hexdigest(),
                short_hex=short.hex(), text_bytes=n_bytes,
                source="oracle/_ref/ref_trace: whole reference Predictor (g++ -O2 strict, analysis off) on the "
                       "first %d bytes of dictionary/english.dic" % n_bytes)
    np.savez_compressed(os.path.join(HERE, "trace_english.npz"), meta=json.dumps(meta),
                        pred=pred.view(np.uint32), act=np.packbits(act, axis=1, bitorder="little"),
                        ctx=ctx, bits=np.packbits(bits, bitorder="little"), outs=outs.view(np.uint32),
                        p=p.view(np.uint32))
    print(f"trace_english: T={T} n={n} M={M} long={len(long_b)}B active avg={act.sum(1).mean():.1f}")


if __name__ == "__main__":
    names = sys.argv[1:] or (list(CASES) + list(IND_CASES) + list(LSTM_CASES) + ["trace"])
    for nm in names:
        if nm == "trace":
            run_trace()
        elif nm in IND_CASES:
            run_ind_case(nm)
        elif nm in LSTM_CASES:
            run_lstm_case(nm)
        else:
            run_case(nm)
