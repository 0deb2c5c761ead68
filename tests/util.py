"""Shared helpers for the test-suite."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def T(a, device=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device) if device is not None else t


def sd(g, prefix="sd.", device=None):
    return {k[len(prefix):]: T(v, device) for k, v in g.items() if k.startswith(prefix)}


def rel_err(got: torch.Tensor, ref: torch.Tensor) -> float:
    """max |got-ref| / max(1, max|ref|): the north_star's '1e-4 relative fp32' bar."""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / max(1.0, float(ref.abs().max())))


REL_TOL = 1e-4  # BASELINE.json north_star: "within 1e-4 relative fp32"
