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


def grad_close(got: torch.Tensor, ref: torch.Tensor, l2: float = 1e-3, cap: float = 5e-2) -> bool:
    """Gradient comparison for LARGE problems: relative L2 within `l2` and the largest single deviation within `cap` of the
    maximum.  A max-norm bound at the 1e-4 level is only sound while the number of hidden pre-activations is small: among
    ~1e7+ of them a few lie within fp32 rounding of zero, their ReLU gate resolves differently under two summation orders
    (MFMA / library GEMM / CPU convolution), and each such unit moves one cell's gradient by 1e-3 ... 1e-2 of the maximum."""
    got, ref = got.detach().double().cpu().reshape(-1), ref.detach().double().cpu().reshape(-1)
    d = got - ref
    return float(d.norm() / ref.norm().clamp_min(1e-12)) < l2 and float(d.abs().max() / ref.abs().max().clamp_min(1e-12)) < cap


# relative margin below which a ReLU gate may resolve differently under two fp32 summation orders: 16 ulp of the term-magnitude
# bound (a 49..97-term fp32 dot product differs between orders by a few ulp of that bound)
GATE_K = 1e-6


def grads_match_outside(got: torch.Tensor, ref: torch.Tensor, region: torch.Tensor, tol: float = 2e-4):
    """Proof hook for per-cell gradients [.., H, W]: every element OUTSIDE `region` (the influence region of the oracle's
    near-zero ReLU gates, nca_oracle.cond_gate_influence) meets the max-norm bound `tol`; returns (ok, n_outside_fail, n_inside_fail)."""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    bad = (got - ref).abs() > tol * max(float(ref.abs().max()), 1e-6)
    reg = region.expand_as(bad)
    return int((bad & ~reg).sum()) == 0, int((bad & ~reg).sum()), int((bad & reg).sum())
