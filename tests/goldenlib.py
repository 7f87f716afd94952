"""Shared helpers for tests that replay tests/golden/ fixtures."""
import hashlib
import json
import os

import numpy as np

from gmix_amd import topology

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return meta, z


def topo_of(meta):
    return topology.Topology(meta["n"], [tuple(m) for m in meta["mixers"]], skip=meta["skip"])


def synth_kwargs(meta):
    kw = dict(meta.get("synth", {}))
    nolearn = kw.pop("nolearn_from", None)
    return kw, nolearn


def sha256(b):
    return hashlib.sha256(b).hexdigest()


def unpack_trace(z, meta):
    T, n = meta["T"], meta["n"]
    pred = z["pred"].view(np.float32)
    act = np.unpackbits(z["act"], axis=1, bitorder="little")[:, :n]
    bits = np.unpackbits(z["bits"], bitorder="little")[:T]
    return pred, act, z["ctx"], bits, z["outs"], z["p"]


def ind_case(name):
    """(meta, models, ctx, bit_context, bits, nolearn_from, z) of an Indirect fixture; the inputs
    are re-derived from the seed (oracle/gmx_ind_synth.h)."""
    from oracle import gmxo
    meta, z = load(name)
    models = [tuple(m) for m in meta["models"]]
    kw = dict(meta["synth"])
    nolearn = kw.pop("nolearn_from", None)
    ctx, bc, bits = gmxo.ind_synth(len(models), meta["T"], seed=kw.get("seed", 0), ctx_mod=kw.get("ctx_mod", (0,) * 4))
    return meta, models, ctx, bc, bits, nolearn, z
