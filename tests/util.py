"""Shared helpers for the test-suite (golden loading, error metrics)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    arrs = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, arrs


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    """||a-b|| / ||b|| per tensor (the north_star's 'rel' metric)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = float(b.norm())
    return float((a - b).norm()) / (den if den > 0 else 1.0)


def cfg_params(meta, dtype=torch.float32):
    from oracle import conformer_oracle as O
    keys = ("vocab", "n_mel", "n_blocks", "d", "n_heads", "ksize", "lstm_hidden", "seed")
    return O.make_params(**{k: meta[k] for k in keys}, dtype=dtype)


def golden_pick(t: torch.Tensor, key: str) -> torch.Tensor:
    """The element selection a golden key was stored with: the whole tensor, or the strided sample ``flat[3::p]`` for keys
    ending in ``@s<p>`` (tests/golden/make_golden_autocast.py)."""
    if "@s" in key:
        return t.detach().flatten()[3::int(key.rsplit("@s", 1)[1])]
    return t.detach()


def golden_find(g: dict, stem: str):
    """Key of ``stem`` in a golden dict, with or without a sampling suffix; None if absent."""
    if stem in g:
        return stem
    for k in g:
        if k.startswith(stem + "@s"):
            return k
    return None
