"""torch.autograd.Function wrappers: T fused NCA steps as ONE autograd node, backward by recomputation
from the saved per-step states, so PyTorch-ROCm autograd, the optimiser
and the style loss sit on top unchanged."""
from typing import Optional

import torch

from . import ops


def _needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


# ------------------------------------------------------------------------------------ ConditionedNCA
class _CondGrow(torch.autograd.Function):
    """x float32 or bfloat16 (a bf16 pool, BASELINE configs[2]: the history ring -- the saved-for-backward set -- is then
    bf16 too, half the bytes).  goal is the encoder's float32 output; for a bf16 state it is rounded to bf16 for the kernels
    and its gradient comes back float32.  Gradients are float32 except dL/dx0, which takes x's dtype."""

    @staticmethod
    def forward(ctx, x, goal, wp, w1, b1, w2, b2, w3, cfg):
        T, us = cfg["T"], cfg["us"]
        w = ops.CondWeights(wp, w1, b1, w2, b2, w3, x)
        gk = goal if (goal is None or goal.dtype == x.dtype) else goal.to(x.dtype)
        out, states, pre = ops.cond_grow(x, T, gk, us, w, cfg["alive_ch"], cfg["thr"], cfg["fire_rate"], cfg["lo"],
                                         cfg["hi"], cfg["seed"], cfg["step0"], keep_history=True)
        ctx.cfg, ctx.w = cfg, w
        ctx.goal_dtype = None if goal is None else goal.dtype
        ctx.save_for_backward(states, pre, gk if gk is not None else x.new_empty(0))
        return out

    @staticmethod
    def backward(ctx, g_out):
        states, pre, goal = ctx.saved_tensors
        cfg, w = ctx.cfg, ctx.w
        g = ops.cond_grow_backward(states, pre, goal if goal.numel() else None, cfg["us"], w, g_out.float().contiguous(),
                                   cfg["T"], cfg["alive_ch"], cfg["thr"], cfg["fire_rate"], cfg["lo"], cfg["hi"],
                                   cfg["seed"], cfg["step0"])
        ggoal = g["goal"] if g["goal"] is None else g["goal"].to(ctx.goal_dtype)
        return (g["x0"].to(states.dtype), ggoal, g["wp"].view_as(ctx.w.wp).reshape(-1, 1, 3, 3), g["w1"][:, :, None, None],
                g["b1"], g["w2"][:, :, None, None], g["b2"], g["w3"][:, :, None, None], None)


class _HipCondPerceive(torch.autograd.Function):
    """perception_net (nca.py:99-107) on the HIP stencil; differentiated through the same map written as a grouped conv."""

    @staticmethod
    def forward(ctx, z, wp):
        ctx.save_for_backward(z, wp)
        return ops.cond_perceive(z, wp)

    @staticmethod
    def backward(ctx, gp):
        z, wp = ctx.saved_tensors
        with torch.enable_grad():
            zr, wr = z.detach().requires_grad_(True), wp.detach().requires_grad_(True)
            p = torch.nn.functional.conv2d(zr, wr, None, 1, 1, 1, zr.shape[1])
            gz, gw = torch.autograd.grad(p, (zr, wr), gp)
        return gz, gw


def _cond_grow_composed(model, x, goal, T, us):
    """Differentiable grow for shapes the fused backward kernels do not cover (W % 4 != 0, or C > 32; the reference's
    DEFAULT model, C = 20, nca.py:62-94, runs the fused kernels): every step is the reference's sequence (nca.py:181-195) with the alive masks and the
    depthwise perception on the HIP kernels and the three 1x1 convolutions as library GEMMs, under PyTorch autograd."""
    a, thr, C = model._alive_ch(), model.alpha_living_threshold, model.num_channels
    gpad = None
    if goal is not None:
        gpad = torch.nn.functional.pad(goal.float(), (0, 0, 0, 0, C - goal.shape[1], 0))
    alive = (lambda t: ops.cond_alive(t.detach().contiguous(), a, thr).float()) if a >= 0 else (lambda t: torch.ones_like(t[:, :1]))
    for t in range(T):
        if us is not None and us.dtype == torch.int32:      # bit-packed masks (ConditionedNCA._draw on a device)
            fire = ops.unpack_fire_mask(us[t:t + 1], x.shape[0], x.shape[2], x.shape[3])[0]
        else:
            u = us[t] if us is not None else ops.philox_uniform(x.shape[0], x.shape[2], x.shape[3], model.mask_seed,
                                                                model._mask_step - T + t, x.device)
            fire = (u.clamp(0.0, 1.0) < model.cell_fire_rate).float()
        pre = alive(x)
        z = x if gpad is None else x + gpad * pre
        out = model.update_net.out(_HipCondPerceive.apply(z.contiguous(), model.perception_net.weight))
        x1 = x + fire * out
        x = torch.clamp(x1 * (pre * alive(x1)), -10.0, 10.0)
    return x


def cond_grow_autograd(model, x: torch.Tensor, goal: Optional[torch.Tensor], T: int) -> torch.Tensor:
    if T == 0:
        return x
    bf16 = x.dtype == torch.bfloat16      # bf16 pool (BASELINE configs[2]): bf16-storage kernels, forward and backward
    u = model.update_net.out
    params = (model.perception_net.weight, u[0].weight, u[0].bias, u[2].weight, u[2].bias, u[4].weight)
    if bf16 and (x.shape[1] > 20 or x.shape[3] % 4 != 0):
        # the bf16-storage kernels (forward and backward on bf16 MFMA) cover C <= 20 -- the reference default model -- at
        # W % 4 == 0; anything else keeps its bf16 POOL but steps in fp32 (widening is exact) and returns the pool's dtype
        return cond_grow_autograd(model, x.float(), goal, T).to(torch.bfloat16)
    x = x.contiguous() if bf16 else x.float().contiguous()
    us = model._draw(x, T)
    cfg = dict(T=T, us=us, alive_ch=model._alive_ch(), thr=model.alpha_living_threshold, fire_rate=model.cell_fire_rate,
               lo=-10.0, hi=10.0, seed=model.mask_seed, step0=model._mask_step)
    model._mask_step += T
    if _needs_grad(x, goal, *params):
        if x.shape[1] > 32 or x.shape[3] % 4 != 0:     # beyond the fused backward kernels (include/ncahip.h): composed pass
            return _cond_grow_composed(model, x.float(), goal, T, us).to(x.dtype)
        return _CondGrow.apply(x, goal, *params, cfg)
    if bf16:
        goal = None if goal is None else goal.detach().to(torch.bfloat16)
    w = ops.CondWeights(*params, x)
    out, _, _ = ops.cond_grow(x, T, goal, us, w, cfg["alive_ch"], cfg["thr"], cfg["fire_rate"], cfg["lo"], cfg["hi"],
                              cfg["seed"], cfg["step0"])
    return out


# ------------------------------------------------------------------------------------ DyNCA
class _DyncaNSteps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cond, w1, b1, w2, b2, cfg):
        w = ops.DyncaWeights(w1, b1, w2, b2, x)
        out, states = ops.dynca_nsteps(x, cfg["T"], cond, cfg["us"], w, cfg["pad"], cfg["rate"], cfg["seed"],
                                       cfg["step0"], keep_history=True, two_scale=cfg.get("two_scale", False))
        ctx.cfg, ctx.w = cfg, w
        ctx.save_for_backward(states, cond if cond is not None else x.new_empty(0))
        return out.clone(), states if cfg["want_states"] else None

    @staticmethod
    def backward(ctx, g_out, g_states):
        states, cond = ctx.saved_tensors
        cfg = ctx.cfg
        gfin = g_out.contiguous()
        if g_states is not None:                       # cotangents of intermediate states (return_middle_feature)
            gfin = gfin + g_states[cfg["T"]]
        g = ops.dynca_nsteps_backward(states, cond if cond.numel() else None, cfg["us"], ctx.w, gfin.float(),
                                      g_states, cfg["T"], cfg["pad"], cfg["rate"], cfg["seed"], cfg["step0"],
                                      two_scale=cfg.get("two_scale", False))
        return g["x0"].to(states.dtype), None, g["w1"][:, :, None, None], g["b1"], g["w2"][:, :, None, None], g["b2"], None  # no grad to cond (dynca.py:123)


def dynca_nsteps_autograd(model, x, cond, T, update_rate, want_states=False, two_scale=False):
    bf16 = x.dtype == torch.bfloat16      # bf16 pool: bf16-storage entry points (storage format only), forward and backward
    x = x.contiguous() if bf16 else x.float().contiguous()
    params = (model.w1.weight, model.w1.bias, model.w2.weight, model.w2.bias)
    us = model._draw(x, T, update_rate)
    cfg = dict(T=T, us=us, pad=model.padding_mode, rate=float(update_rate), seed=model.mask_seed, step0=model._mask_step,
               want_states=want_states, two_scale=two_scale)
    model._mask_step += T
    if _needs_grad(x, *params):
        return _DyncaNSteps.apply(x, cond, *params, cfg)
    w = ops.DyncaWeights(*params, x)
    out, states = ops.dynca_nsteps(x, T, cond, us, w, cfg["pad"], cfg["rate"], cfg["seed"], cfg["step0"],
                                   keep_history=want_states, two_scale=two_scale)
    return out, (states if want_states else None)


# ------------------------------------------------------------------------------------ standalone stencil (multi-scale path)
_SOBEL_X = ((-1.0, 0.0, 1.0), (-2.0, 0.0, 2.0), (-1.0, 0.0, 1.0))
_LAPL = ((1.0, 2.0, 1.0), (2.0, -12.0, 2.0), (1.0, 2.0, 1.0))
_TORCH_PAD = {"replicate": "replicate", "circular": "circular", "reflect": "reflect", "constant": "constant", "zeros": "constant"}


def _perceive_ref(x, pad_mode):
    """The same map as the HIP stencil in torch ops on the device -- used only to differentiate it (dynca.py:84-98)."""
    c = x.shape[1]
    sx = torch.tensor(_SOBEL_X, device=x.device)
    filt = torch.stack([sx, sx.t(), torch.tensor(_LAPL, device=x.device)])
    z = torch.nn.functional.pad(x, [1, 1, 1, 1], _TORCH_PAD[pad_mode])
    ys = [torch.nn.functional.conv2d(z, f.reshape(1, 1, 3, 3).repeat(c, 1, 1, 1), groups=c) for f in filt]
    return torch.cat([x] + ys, dim=1)


class _HipPerceive(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pad_mode):
        ctx.pad_mode = pad_mode
        ctx.save_for_backward(x)
        return ops.dynca_perceive(x, pad_mode)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        with torch.enable_grad():
            xr = x.detach().requires_grad_(True)
            (gx,) = torch.autograd.grad(_perceive_ref(xr, ctx.pad_mode), xr, gy)
        return gx, None


def hip_perceive(x: torch.Tensor, pad_mode: str) -> torch.Tensor:
    x = x.float().contiguous()
    if _needs_grad(x):
        return _HipPerceive.apply(x, pad_mode)
    return ops.dynca_perceive(x, pad_mode)
