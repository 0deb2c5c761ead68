"""CPU oracle for the NCA step hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement of the reference algorithm for the path named by
BASELINE.json's north_star.  It is the checker, never the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  The shipped path (``video-stylization-with-nca_amd/``) never
imports anything from ``oracle/`` and fails loudly when libncahip.so is absent.

Parity pinning: every function below is checked bit-for-bit (CPU, fp32) against
the reference's own PyTorch modules imported in the build container; the
captured inputs/outputs live in ``tests/golden/*.npz`` (generator:
``tests/golden/gen_golden.py``).  ``tests/test_oracle_golden.py`` replays them
anywhere, without the reference.

All ``file:line`` citations are relative to the reference checkout
(smehra34/Video-Stylization-with-NCA).

The restatement is written with the same aten ops, in the same order, as the
reference so that results are bit-identical on CPU.  A second, independent
restatement with explicit numpy loops (``*_np``) exists for small cases to
guard against a shared misunderstanding of an aten op's semantics.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# fixed filters (ConditioneDyNCA/models/dynca.py:67-73, :188-198; encoder.py:12-27)
# --------------------------------------------------------------------------
SOBEL_X = [[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]]
SOBEL_Y = [[-1.0, -2.0, -1.0], [0.0, 0.0, 0.0], [1.0, 2.0, 1.0]]  # == SOBEL_X transposed
LAPLACIAN = [[1.0, 2.0, 1.0], [2.0, -12.0, 2.0], [1.0, 2.0, 1.0]]

PAD_MODES = ("constant", "replicate", "circular", "reflect")  # F.pad modes, dynca.py:85


# ==========================================================================
# ConditionedNCA  (EncoderConditioning/nca.py)
# ==========================================================================
def cond_alive(x: torch.Tensor, alive_ch: int, thr: float = 0.1,
               use_living_channel: bool = True) -> torch.Tensor:
    """nca.py:152-163 -- 3x3 max-pool on the alpha channel, strict '>'."""
    if not use_living_channel:
        return torch.ones_like(x, dtype=torch.bool)
    return F.max_pool2d(x[:, alive_ch:alive_ch + 1], kernel_size=3, stride=1, padding=1) > thr


def cond_fire_mask(u: torch.Tensor, fire_rate: float) -> torch.Tensor:
    """nca.py:165-174 -- (clamp(u,0,1) < rate).float(); ``u`` is the rand_like draw."""
    return (torch.clamp(u, 0.0, 1.0).float() < fire_rate).float()


def cond_perceive(z: torch.Tensor, wp: torch.Tensor) -> torch.Tensor:
    """nca.py:99-107,177 -- learned depthwise 3x3, groups=C, zero pad, no bias.

    ``wp`` is ``perception_net.weight`` [3C,1,3,3]; output channel 3c+k belongs
    to input channel c.
    """
    return F.conv2d(z, wp, None, 1, 1, 1, z.shape[1])


def cond_update_net(p: torch.Tensor, prm: Dict[str, torch.Tensor]) -> torch.Tensor:
    """nca.py:40-46,57-58 -- 1x1 conv -> ReLU -> 1x1 conv -> ReLU -> 1x1 conv (no bias)."""
    h = F.conv2d(p, prm["update_net.out.0.weight"], prm["update_net.out.0.bias"])
    h = F.relu(h)
    h = F.conv2d(h, prm["update_net.out.2.weight"], prm["update_net.out.2.bias"])
    h = F.relu(h)
    return F.conv2d(h, prm["update_net.out.4.weight"], None)


def cond_step(x: torch.Tensor, goal_enc: torch.Tensor, u: torch.Tensor,
              prm: Dict[str, torch.Tensor], alive_ch: int, thr: float = 0.1,
              fire_rate: float = 0.5, use_living_channel: bool = True,
              return_all: bool = False):
    """One ConditionedNCA.forward, nca.py:181-195, with the uniform draw ``u`` explicit."""
    pre = cond_alive(x, alive_ch, thr, use_living_channel)          # :185
    rmask = cond_fire_mask(u, fire_rate)                            # :187
    z = x + goal_enc * pre                                          # :177
    p = cond_perceive(z, prm["perception_net.weight"])              # :177
    out = cond_update_net(p, prm)                                   # :178
    x1 = x + rmask * out                                            # :189
    post = cond_alive(x1, alive_ch, thr, use_living_channel)        # :191
    life = (pre & post).float()                                     # :192
    x2 = x1 * life                                                  # :193
    x2 = torch.clamp(x2, -10.0, 10.0)                               # :194
    if return_all:
        return dict(pre=pre, rmask=rmask, z=z, p=p, out=out, x1=x1, post=post, life=life, x2=x2)
    return x2


def cond_pad_goal(goal_enc: torch.Tensor, num_channels: int) -> torch.Tensor:
    """nca.py:199-203 -- zero-pad the encoding in FRONT up to C channels."""
    hid = goal_enc.shape[1]
    if hid == num_channels:
        return goal_enc
    return F.pad(goal_enc, (0, 0, 0, 0, num_channels - hid, 0))


def cond_grow(x: torch.Tensor, goal_enc_padded: torch.Tensor, us: Sequence[torch.Tensor],
              prm: Dict[str, torch.Tensor], alive_ch: int, thr: float = 0.1,
              fire_rate: float = 0.5, collect: bool = False):
    """nca.py:207-208 -- T x forward with the per-step draws ``us`` explicit."""
    states = []
    for u in us:
        x = cond_step(x, goal_enc_padded, u, prm, alive_ch, thr, fire_rate)
        if collect:
            states.append(x)
    return (x, states) if collect else x


def cond_grow_rng(x: torch.Tensor, goal_enc_padded: torch.Tensor, num_steps: int,
                  prm: Dict[str, torch.Tensor], alive_ch: int, thr: float = 0.1,
                  fire_rate: float = 0.5) -> torch.Tensor:
    """Same as cond_grow but draws u with the reference's own call
    (``torch.rand_like(x[:, 0:1])``, nca.py:172) so CPU global-RNG streams line up."""
    for _ in range(num_steps):
        u = torch.rand_like(x[:, 0:1])
        x = cond_step(x, goal_enc_padded, u, prm, alive_ch, thr, fire_rate)
    return x


def cond_generate_seed(n: int, num_channels: int, alive_ch: int, size: int) -> torch.Tensor:
    """nca.py:130-150 -- zeros, alpha AND all later channels = 1 at the centre cell."""
    seed = torch.zeros(n, num_channels, size, size)
    seed[:, alive_ch:, size // 2, size // 2] = 1.0
    return seed


def gaussian_kernel_5x5(sigma: float = 1.0) -> torch.Tensor:
    """encoder.py:60-64 -- evaluated in float64 python/numpy then normalised, as there."""
    size = 5
    k = torch.tensor([[(1 / (2 * np.pi * sigma ** 2)) *
                       np.exp(-((i - size // 2) ** 2 + (j - size // 2) ** 2) / (2 * sigma ** 2))
                       for j in range(size)] for i in range(size)])
    k = k / torch.sum(k)
    return k.view(1, 1, size, size).float()


def image_encoder(img: torch.Tensor, prm: Dict[str, torch.Tensor], prefix: str = "encoder.") -> torch.Tensor:
    """encoder.py:37-57 -- gray -> sobel/laplacian, per-channel 5x5 blur, 3x3 conv-ReLU-3x3 conv."""
    ch = img.shape[1]
    gray = torch.mean(img, dim=1, keepdim=True)
    sx = F.conv2d(gray, torch.tensor([[SOBEL_X]]), padding=1)
    sy = F.conv2d(gray, torch.tensor([[SOBEL_Y]]), padding=1)
    lp = F.conv2d(gray, torch.tensor([[LAPLACIAN]]), padding=1)
    gk = gaussian_kernel_5x5(1.0)
    blurred = torch.cat([F.conv2d(img[:, i:i + 1], gk, padding=2) for i in range(ch)], dim=1)
    feat = torch.cat((sx, sy, lp, blurred), dim=1)
    h = F.conv2d(feat, prm[prefix + "embed.0.weight"], prm[prefix + "embed.0.bias"], padding=1)
    h = F.relu(h)
    return F.conv2d(h, prm[prefix + "embed.2.weight"], None, padding=1)


# ==========================================================================
# DyNCA  (ConditioneDyNCA/models/dynca.py, ExtraChannels/models/dynca.py)
# ==========================================================================
def _depthwise_fixed(z: torch.Tensor, filt, pad_mode: str) -> torch.Tensor:
    """dynca.py:83-86 -- F.pad(z,[1,1,1,1],mode) then grouped conv with one repeated 3x3."""
    c = z.shape[1]
    w = torch.tensor(filt, dtype=torch.float32).reshape(1, 1, 3, 3).repeat(c, 1, 1, 1)
    z = F.pad(z, [1, 1, 1, 1], pad_mode)
    return F.conv2d(z, w, groups=c)


def dynca_perceive(x: torch.Tensor, pad_mode: str = "replicate", scale: int = 0) -> torch.Tensor:
    """dynca.py:75-100 -- [x | Sx*x | Sy*x | L*x]; scale>0: bilinear down, perceive, bilinear up."""
    if scale != 0:
        _, _, h, w = x.shape
        x = F.interpolate(x, size=(int(h // 2 ** scale), int(w // 2 ** scale)),
                          mode="bilinear", align_corners=False)
    y1 = _depthwise_fixed(x, SOBEL_X, pad_mode)
    y2 = _depthwise_fixed(x, [list(r) for r in zip(*SOBEL_X)], pad_mode)  # sobel_x.T, dynca.py:69
    y3 = _depthwise_fixed(x, LAPLACIAN, pad_mode)
    y = torch.cat([x, y1, y2, y3], dim=1)
    if scale != 0:
        y = F.interpolate(y, size=(h, w), mode="bilinear", align_corners=False)
    return y


def dynca_perceive_multiscale(x: torch.Tensor, pad_mode: str, scales: Sequence[int] = (0,),
                              cond: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dynca.py:102-115 -- mean over scales, then cat the conditioning channels."""
    y = sum(dynca_perceive(x, pad_mode, s) for s in scales)
    y = y / len(scales)
    if cond is not None:
        y = torch.cat([y, cond], dim=1)
    return y


def edge_extractor(img: torch.Tensor, transform: Optional[str] = "tanh") -> torch.Tensor:
    """dynca.py:182-213 -- sobel_x, sobel_y, laplacian of a 1-channel image, ZERO pad, optional tanh."""
    sx = F.conv2d(img, torch.tensor([[SOBEL_X]]), padding=1)
    sy = F.conv2d(img, torch.tensor([[SOBEL_Y]]), padding=1)
    lp = F.conv2d(img, torch.tensor([[LAPLACIAN]]), padding=1)
    out = torch.cat((sx, sy, lp), dim=1)
    return torch.tanh(out) if transform == "tanh" else out


def cpe2d(b: int, h: int, w: int) -> torch.Tensor:
    """dynca.py:226-253 -- Cartesian positional encoding, [b,2,h,w]."""
    xs = torch.arange(h) / h
    ys = torch.arange(w) / w
    xs = 2.0 * (xs - 0.5 + 0.5 / h)
    ys = 2.0 * (ys - 0.5 + 0.5 / w)
    emb = torch.zeros((2, h, w))
    emb[:1] = xs[None, :, None]
    emb[1:2] = ys[None, None, :]
    return emb.unsqueeze(0).repeat(b, 1, 1, 1)


def dynca_step(x: torch.Tensor, cond: Optional[torch.Tensor], u: torch.Tensor,
               prm: Dict[str, torch.Tensor], pad_mode: str = "replicate",
               update_rate: float = 0.5, scales: Sequence[int] = (0,), return_all: bool = False):
    """One DyNCA.forward, dynca.py:117-138; ``cond`` is the already-extracted
    conditioning [B,c_cond,H,W] (edges or CPE) or None; ``u`` the torch.rand draw (:131)."""
    y = dynca_perceive_multiscale(x, pad_mode, scales, cond)            # :125
    h = F.relu(F.conv2d(y, prm["w1.weight"], prm["w1.bias"]))          # :128
    dx = F.conv2d(h, prm["w2.weight"], prm["w2.bias"])                 # :128
    m = (u + update_rate).floor()                                       # :131
    xn = x + dx * m                                                     # :133
    if return_all:
        return dict(y=y, dx=dx, m=m, x=xn)
    return xn


def dynca_nsteps(x: torch.Tensor, cond: Optional[torch.Tensor], us: Sequence[torch.Tensor],
                 prm: Dict[str, torch.Tensor], pad_mode: str = "replicate", update_rate: float = 0.5,
                 scales: Sequence[int] = (0,), collect: bool = False):
    """dynca.py:168-178."""
    states = []
    for u in us:
        x = dynca_step(x, cond, u, prm, pad_mode, update_rate, scales)
        if collect:
            states.append(x)
    return (x, states) if collect else x


def dynca_nsteps_rng(x, cond, step_n: int, prm, pad_mode="replicate", update_rate=0.5, scales=(0,)):
    """As dynca_nsteps, drawing ``torch.rand(b,1,h,w)`` per step like dynca.py:131."""
    b, _, h, w = x.shape
    for _ in range(step_n):
        u = torch.rand(b, 1, h, w)
        x = dynca_step(x, cond, u, prm, pad_mode, update_rate, scales)
    return x


def dynca_to_rgb(x: torch.Tensor, c_out: int) -> torch.Tensor:
    """dynca.py:140-141."""
    return x[:, :c_out] * 2.0


def dynca_seed(n: int, c: int, size, mode: str = "zeros", random_seed: int = 42) -> torch.Tensor:
    """dynca.py:143-166 -- note size=(x,y) -> [n,C,size_y,size_x].  'random' reseeds torch (as there)."""
    sx, sy = (size, size) if isinstance(size, int) else size
    if mode == "zeros":
        return torch.zeros(n, c, sy, sx)
    if mode == "center_on":
        sd = torch.zeros(n, c, sy, sx)
        sd[:, :, sy // 2, sx // 2] = 1.0
        return sd
    np.random.seed(random_seed)
    torch.manual_seed(random_seed)
    sd = torch.rand(1, c, sy, sx) - 0.5
    return torch.cat([sd.clone() for _ in range(n)])


# ==========================================================================
# Backward of one step (autograd on the restatement == autograd on the reference)
# ==========================================================================
def cond_grow_loss_grads(x0, goal_enc_padded, us, prm, alive_ch, thr, fire_rate, cot):
    """d<cot, grow(x0)>/d{x0, goal_enc, weights} through the restatement (eager autograd)."""
    x0 = x0.clone().requires_grad_(True)
    g = goal_enc_padded.clone().requires_grad_(True)
    p = {k: v.clone().requires_grad_(True) for k, v in prm.items()
         if k.startswith("perception_net") or k.startswith("update_net")}
    xT = cond_grow(x0, g, us, p, alive_ch, thr, fire_rate)
    (xT * cot).sum().backward()
    grads = {k: v.grad for k, v in p.items()}
    return xT.detach(), x0.grad, g.grad, grads


def cond_gate_margin(x0, goal_enc_padded, us, prm, alive_ch, thr=0.1, fire_rate=0.5, use_living_channel=True):
    """Proof hook for gradient comparisons: per batch item, the smallest RELATIVE margin of a ReLU gate that carries
    gradient along the oracle trajectory -- min over steps, over cells that fired and live (r * life = 1: nca.py:189-193) and
    over the 128 hidden units of |pre-activation| / (sum_k |w_k| |in_k| + |b|).  A gate whose margin is below the rounding
    spread of an fp32 dot product (n terms: at worst n * 2^-24, typically sqrt(n) * 2^-24) may resolve differently under
    another summation order (MFMA vs CPU convolution): ONE such unit moves the gradients through it by O(1e-3) of the maximum
    while the forward agrees to 1e-6.  Items whose margin is well above that have no such excuse."""
    w1, b1 = prm["update_net.out.0.weight"], prm["update_net.out.0.bias"]
    w2, b2 = prm["update_net.out.2.weight"], prm["update_net.out.2.bias"]
    x = x0
    best = torch.full((x0.shape[0],), float("inf"))
    with torch.no_grad():
        for u in us:
            d = cond_step(x, goal_enc_padded, u, prm, alive_ch, thr, fire_rate, use_living_channel, return_all=True)
            carries = (d["rmask"] * d["life"][:, :1]) > 0     # [B,1,H,W] (life has C channels when use_living_channel is off)
            pre1 = F.conv2d(d["p"], w1, b1)
            bnd1 = F.conv2d(d["p"].abs(), w1.abs(), b1.abs())
            h1 = F.relu(pre1)
            pre2 = F.conv2d(h1, w2, b2)
            bnd2 = F.conv2d(h1, w2.abs(), b2.abs())
            for pre, bnd in ((pre1, bnd1), (pre2, bnd2)):
                m = (pre.abs() / bnd.clamp_min(1e-30)).masked_fill(~carries.expand_as(pre), float("inf"))
                best = torch.minimum(best, m.flatten(1).min(dim=1).values)
            x = d["x2"]
    return best


def cond_gate_influence(x0, goal_enc_padded, us, prm, alive_ch, k, thr=0.1, fire_rate=0.5, use_living_channel=True):
    """Where may dL/dx0 and dL/dgoal legitimately differ between two fp32 evaluations?  A gradient-carrying gate with relative
    margin < k at step t (0-based), cell c, perturbs dL/dz_t on c's 3x3 neighbourhood, and every earlier step's stencil
    adjoint widens that by one cell: dL/dx0 and dL/dgoal can move within Chebyshev distance t + 1 of c, nowhere else.
    Returns (region [B,1,H,W] bool, number of such gates per item [B])."""
    w1, b1 = prm["update_net.out.0.weight"], prm["update_net.out.0.bias"]
    w2, b2 = prm["update_net.out.2.weight"], prm["update_net.out.2.bias"]
    x = x0
    region = torch.zeros(x0.shape[0], 1, x0.shape[2], x0.shape[3], dtype=torch.bool)
    count = torch.zeros(x0.shape[0], dtype=torch.long)
    with torch.no_grad():
        for t, u in enumerate(us):
            d = cond_step(x, goal_enc_padded, u, prm, alive_ch, thr, fire_rate, use_living_channel, return_all=True)
            carries = (d["rmask"] * d["life"][:, :1]) > 0
            pre1 = F.conv2d(d["p"], w1, b1)
            bnd1 = F.conv2d(d["p"].abs(), w1.abs(), b1.abs())
            h1 = F.relu(pre1)
            pre2 = F.conv2d(h1, w2, b2)
            bnd2 = F.conv2d(h1, w2.abs(), b2.abs())
            amb = ((pre1.abs() < k * bnd1).any(1, keepdim=True) | (pre2.abs() < k * bnd2).any(1, keepdim=True)) & carries
            count += ((pre1.abs() < k * bnd1) & carries).flatten(1).sum(1) + ((pre2.abs() < k * bnd2) & carries).flatten(1).sum(1)
            r = t + 1
            region |= F.max_pool2d(amb.float(), 2 * r + 1, 1, r) > 0
            x = d["x2"]
    return region, count


def dynca_gate_margin(x0, cond, us, prm, pad_mode, update_rate=0.5, scales=(0,)):
    """As cond_gate_margin for the DyNCA step (one hidden layer, dynca.py:126-133): per batch item, min over steps, updated
    cells (m = 1) and hidden units of |w1 y + b1| / (|w1| |y| + |b1|)."""
    x = x0
    best = torch.full((x0.shape[0],), float("inf"))
    with torch.no_grad():
        for u in us:
            r = dynca_step(x, cond, u, prm, pad_mode, update_rate, scales, return_all=True)
            pre = F.conv2d(r["y"], prm["w1.weight"], prm["w1.bias"])
            bnd = F.conv2d(r["y"].abs(), prm["w1.weight"].abs(), prm["w1.bias"].abs())
            m = (pre.abs() / bnd.clamp_min(1e-30)).masked_fill(~(r["m"] > 0).expand_as(pre), float("inf"))
            best = torch.minimum(best, m.flatten(1).min(dim=1).values)
            x = r["x"]
    return best


def dynca_gate_influence(x0, cond, us, prm, pad_mode, k, update_rate=0.5, scales=(0,)):
    """As cond_gate_influence for the DyNCA step: (region [B,1,H,W] of dL/dx0 that a hidden pre-activation of an UPDATED cell within
    relative margin k of zero can move -- Chebyshev distance t + 1 of the cell for a gate at step t, one more ring per coarser
    perception scale; circular padding wraps --, number of such gates per item)."""
    x = x0
    region = torch.zeros(x0.shape[0], 1, x0.shape[2], x0.shape[3], dtype=torch.bool)
    count = torch.zeros(x0.shape[0], dtype=torch.long)
    reach = 1 if tuple(scales) == (0,) else 2 ** max(scales) * 2
    with torch.no_grad():
        for t, u in enumerate(us):
            r = dynca_step(x, cond, u, prm, pad_mode, update_rate, scales, return_all=True)
            pre = F.conv2d(r["y"], prm["w1.weight"], prm["w1.bias"])
            bnd = F.conv2d(r["y"].abs(), prm["w1.weight"].abs(), prm["w1.bias"].abs())
            near = (pre.abs() < k * bnd) & (r["m"] > 0)
            count += near.flatten(1).sum(1)
            amb = near.any(1, keepdim=True).float()
            rad = (t + 1) * reach
            if pad_mode == "circular":
                padded = F.pad(amb, (rad, rad, rad, rad), mode="circular") if min(amb.shape[2:]) > rad else amb.new_ones(amb.shape[0], 1, amb.shape[2] + 2 * rad, amb.shape[3] + 2 * rad) * amb.amax((2, 3), keepdim=True)
                region |= F.max_pool2d(padded, 2 * rad + 1, 1, 0) > 0
            else:
                region |= F.max_pool2d(amb, 2 * rad + 1, 1, rad) > 0
            x = r["x"]
    return region, count


def dynca_nsteps_loss_grads(x0, cond, us, prm, pad_mode, update_rate, cot, scales=(0,)):
    x0 = x0.clone().requires_grad_(True)
    p = {k: prm[k].clone().requires_grad_(True) for k in ("w1.weight", "w1.bias", "w2.weight", "w2.bias")}
    xT = dynca_nsteps(x0, cond, us, p, pad_mode, update_rate, scales)
    (xT * cot).sum().backward()
    return xT.detach(), x0.grad, {k: v.grad for k, v in p.items()}


# ==========================================================================
# Counter-based fire-mask RNG used by the HIP kernels in throughput mode.
# Not part of the reference (its mask comes from the device's global torch
# generator, nca.py:172 / dynca.py:131); this is the framework's own contract,
# restated here so tests can reproduce the in-kernel masks exactly.
#   Philox4x32-10 (Salmon et al., SC'11): key=(seed_lo, seed_hi),
#   counter=(cell_group, step_lo, step_hi, 0x4e4341); cell_group = linear cell
#   index (b*H*W + y*W + x) >> 2, lane = index & 3;  u = (word >> 8) * 2^-24.
# ==========================================================================
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)
PHILOX_DOMAIN = 0x4E4341  # 'NCA'


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0 = np.asarray(c0, dtype=np.uint32); c1 = np.asarray(c1, dtype=np.uint32)
    c2 = np.asarray(c2, dtype=np.uint32); c3 = np.asarray(c3, dtype=np.uint32)
    k0 = np.uint32(k0); k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _PHILOX_M0 * c0.astype(np.uint64)
            p1 = _PHILOX_M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32); lo0 = p0.astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32); lo1 = p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + _PHILOX_W0); k1 = np.uint32(k1 + _PHILOX_W1)
    return c0, c1, c2, c3


def philox_uniform(seed: int, step: int, B: int, H: int, W: int) -> np.ndarray:
    """The [B,1,H,W] float32 uniforms the kernels draw for (seed, step)."""
    n = B * H * W
    idx = np.arange(n, dtype=np.uint64)
    grp = (idx >> np.uint64(2)).astype(np.uint32)
    lane = (idx & np.uint64(3)).astype(np.int64)
    ones = np.ones_like(grp)
    r = philox4x32_10(grp, ones * np.uint32(step & 0xFFFFFFFF), ones * np.uint32((step >> 32) & 0xFFFFFFFF),
                      ones * np.uint32(PHILOX_DOMAIN), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(r, axis=1)[np.arange(n), lane]
    u = (words >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return u.reshape(B, 1, H, W)


# ==========================================================================
# Independent numpy restatements (explicit loops; small cases only)
# ==========================================================================
def _pad_index(i: int, n: int, mode: str) -> int:
    """Index into a length-n axis for padded position i in [-1, n]; -1 => contributes zero."""
    if 0 <= i < n:
        return i
    if mode == "constant":
        return -1
    if mode == "replicate":
        return min(max(i, 0), n - 1)
    if mode == "circular":
        return i % n
    if mode == "reflect":
        return -i if i < 0 else 2 * (n - 1) - i
    raise ValueError(mode)


def dynca_perceive_np(x: np.ndarray, pad_mode: str) -> np.ndarray:
    B, C, H, W = x.shape
    out = np.zeros((B, 4 * C, H, W), dtype=np.float64)
    filts = [None, np.array(SOBEL_X), np.array(SOBEL_X).T, np.array(LAPLACIAN)]
    out[:, :C] = x
    for f in range(1, 4):
        for dy in range(3):
            for dx in range(3):
                wgt = filts[f][dy, dx]
                if wgt == 0:
                    continue
                for yy in range(H):
                    sy = _pad_index(yy + dy - 1, H, pad_mode)
                    if sy < 0:
                        continue
                    for xx in range(W):
                        sx = _pad_index(xx + dx - 1, W, pad_mode)
                        if sx < 0:
                            continue
                        out[:, f * C:(f + 1) * C, yy, xx] += wgt * x[:, :, sy, sx]
    return out


def cond_step_np(x, goal_enc, u, prm, alive_ch, thr=0.1, fire_rate=0.5):
    """Loop-level restatement of nca.py:181-195 in float64 (tolerance check, not bit-exact)."""
    x = np.asarray(x, dtype=np.float64); g = np.asarray(goal_enc, dtype=np.float64)
    B, C, H, W = x.shape
    wp = np.asarray(prm["perception_net.weight"], dtype=np.float64).reshape(3 * C, 3, 3)
    w1 = np.asarray(prm["update_net.out.0.weight"], dtype=np.float64).reshape(-1, 3 * C)
    b1 = np.asarray(prm["update_net.out.0.bias"], dtype=np.float64)
    w2 = np.asarray(prm["update_net.out.2.weight"], dtype=np.float64).reshape(w1.shape[0], -1)
    b2 = np.asarray(prm["update_net.out.2.bias"], dtype=np.float64)
    w3 = np.asarray(prm["update_net.out.4.weight"], dtype=np.float64).reshape(C, -1)

    def alive(a):
        m = np.zeros((B, H, W), dtype=bool)
        for yy in range(H):
            for xx in range(W):
                y0, y1 = max(yy - 1, 0), min(yy + 2, H)
                x0, x1 = max(xx - 1, 0), min(xx + 2, W)
                m[:, yy, xx] = a[:, y0:y1, x0:x1].reshape(B, -1).max(axis=1) > thr
        return m

    pre = alive(np.asarray(x[:, alive_ch], dtype=np.float32))
    z = x + g * pre[:, None]
    zp = np.pad(z, ((0, 0), (0, 0), (1, 1), (1, 1)))
    p = np.zeros((B, 3 * C, H, W))
    for c in range(C):
        for k in range(3):
            for dy in range(3):
                for dx in range(3):
                    p[:, 3 * c + k] += wp[3 * c + k, dy, dx] * zp[:, c, dy:dy + H, dx:dx + W]
    h1 = np.maximum(np.einsum("oc,bchw->bohw", w1, p) + b1[None, :, None, None], 0)
    h2 = np.maximum(np.einsum("oc,bchw->bohw", w2, h1) + b2[None, :, None, None], 0)
    o = np.einsum("oc,bchw->bohw", w3, h2)
    r = (np.clip(np.asarray(u, dtype=np.float32), 0, 1) < np.float32(fire_rate)).astype(np.float64)
    x1 = x + r * o
    post = alive(np.asarray(x1[:, alive_ch], dtype=np.float32))
    life = (pre & post).astype(np.float64)[:, None]
    return np.clip(x1 * life, -10.0, 10.0)


# ---------------------------------------------------------------------------------------------------------
# bf16-storage restatement of the ConditionedNCA step (what ncahip_cond_*_bf16 computes; there is no reference
# code for it -- the reference is fp32 only, nca.py:181-195 -- so this states the rounding points of include/ncahip.h:
# bf16 state/goal, f32 perception, bf16 matrix operands with f32 accumulation, bf16 store).  "parity unpinned" by
# reference fixtures by construction; it is pinned to the fp32 oracle through the bound tested in tests/.
def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def cond_step_bf16(x, goal_pad, u, prm, alive_ch=3, thr=0.1, fire_rate=0.5, lo=-10.0, hi=10.0):
    """x, goal_pad: float32 tensors holding bf16-representable values ([B,C,H,W]; goal_pad already padded to C channels).
    Returns (x_next, pre, x_pending): x_next = resolved next state (bf16-representable float32)."""
    pre = cond_alive(x, alive_ch, thr) if alive_ch >= 0 else torch.ones_like(x[:, :1], dtype=torch.bool)
    z = x + goal_pad * pre.float()
    p = cond_perceive(z, prm["perception_net.weight"])
    w1 = _bf(prm["update_net.out.0.weight"].flatten(1)); w2 = _bf(prm["update_net.out.2.weight"].flatten(1))
    w3 = _bf(prm["update_net.out.4.weight"].flatten(1))
    B, K, H, W = p.shape
    pm = _bf(p).permute(0, 2, 3, 1).reshape(-1, K).double()
    h1 = torch.relu(pm @ w1.double().t() + prm["update_net.out.0.bias"].double())
    h2 = torch.relu(_bf(h1.float()).double() @ w2.double().t() + prm["update_net.out.2.bias"].double())
    out = (_bf(h2.float()).double() @ w3.double().t()).float().reshape(B, H, W, -1).permute(0, 3, 1, 2)
    mask = (u.clamp(0.0, 1.0) < fire_rate).float().reshape(B, 1, H, W)
    x_pend = _bf(x + mask * out)
    post = cond_alive(x_pend, alive_ch, thr) if alive_ch >= 0 else torch.ones_like(pre)
    x_next = (x_pend * (pre & post).float()).clamp(lo, hi)
    return x_next, pre, x_pend


def _bf_ste(t):
    """round to bf16 in the forward, identity in the backward (straight-through)"""
    return t + (_bf(t) - t).detach()


def cond_grow_bf16_loss_grads(x0, goal_pad, us, prm, alive_ch, thr, fire_rate, cot, lo=-10.0, hi=10.0):
    """Autograd through T bf16-storage steps with the rounding points of include/ncahip.h (bf16 state / goal, f32 perception,
    bf16 matrix operands and weights, f32 accumulation, bf16 store), every rounding straight-through.  This is the function
    whose gradient ncahip_cond_grow_bwd_bf16 evaluates (its own extra rounding: the gradient operands d out, d2, d1 of the
    backward products are bf16 as well).  Returns (x_T, dL/dx0, dL/dgoal_pad, {param: grad}) for L = <cot, x_T>."""
    x = x0.clone().requires_grad_(True)
    g = goal_pad.clone().requires_grad_(True)
    p = {k: v.clone().requires_grad_(True) for k, v in prm.items() if k.startswith(("perception_net", "update_net"))}
    w1, w2, w3 = (_bf_ste(p[f"update_net.out.{i}.weight"]) for i in (0, 2, 4))
    cur = x
    for u in us:
        pre = cond_alive(cur, alive_ch, thr) if alive_ch >= 0 else torch.ones_like(cur[:, :1], dtype=torch.bool)
        z = cur + g * pre.float()
        pc = cond_perceive(z, p["perception_net.weight"])
        h1 = F.relu(F.conv2d(_bf_ste(pc), w1, p["update_net.out.0.bias"]))
        h2 = F.relu(F.conv2d(_bf_ste(h1), w2, p["update_net.out.2.bias"]))
        out = F.conv2d(_bf_ste(h2), w3, None)
        x1 = _bf_ste(cur + cond_fire_mask(u, fire_rate) * out)
        post = cond_alive(x1, alive_ch, thr) if alive_ch >= 0 else torch.ones_like(pre)
        cur = torch.clamp(x1 * (pre & post).float(), lo, hi)
    (cur * cot).sum().backward()
    return cur.detach(), x.grad, g.grad, {k: v.grad for k, v in p.items()}
