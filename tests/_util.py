"""Shared helpers for the test-suite (fixture loading, tolerances)."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Stated floating-point tolerances (north_star: nodal u and dL/dkappa within 1e-10 rel).
RTOL_U = 1e-10
RTOL_GRAD = 1e-10


def golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def golden_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def rel_err(a, b):
    """max |a-b| / max(|b|_inf, tiny): the 'relative nodal error' of the north star."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(float(np.max(np.abs(b))) if b.size else 0.0, 1e-300)
    return float(np.max(np.abs(a - b))) / scale if b.size else 0.0


def loss_grad(kind, u, data=None):
    """gbar = dL/du for the loss kinds used by the fixtures."""
    if kind == "sum":
        return np.ones_like(u)
    if kind == "sumsq":
        return 2.0 * u
    if kind == "mse":
        return 2.0 * (u - data) / u.size
    raise ValueError(kind)
