"""Tensor-level wrappers over the C ABI: marshal torch CUDA tensors to raw pointers, enqueue on the
current torch stream.  Every function here fails loudly for non-CUDA tensors -- the product path
has no CPU fallback (the CPU restatement lives in oracle/ and is test infrastructure only)."""
from typing import Optional, Sequence

import torch

from . import _capi
from ._capi import PAD_MODES, check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _dev(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _capi.NcaHipError(f"ncahip: `{name}` must be a CUDA (ROCm) tensor -- the NCA hot path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"ncahip: `{name}` must be {dtype}, got {t.dtype}")
    return t.contiguous()


_WS_CACHE = {}


def _workspace(nbytes: int, device: torch.device) -> torch.Tensor:
    """Scratch for the backward drivers, kept per (device, stream) and grown on demand: the drivers take a caller-owned
    workspace of up to ~0.5 GB (BASELINE configs[2]); allocating it afresh per call costs a torch.empty of that size per
    training step.  Work on one stream is ordered, so consecutive calls may share it."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = None
        _WS_CACHE.pop(key, None)
        ws = torch.empty(max(nbytes, 1), device=device, dtype=torch.uint8)
        _WS_CACHE[key] = ws
    return ws


_PERSIST_WS = {}


def _persist_workspace(nbytes: int, device: torch.device):
    """(workspace, epoch) for ncahip_dynca_nsteps_fwd_persist_f32: a dedicated tensor per (device, stream, size), zeroed when it is
    created and whenever its epoch counter runs out; every call gets the next epoch (the library never clears the workspace: the
    exchanged pairs are tagged epoch * 4096 + step)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream(), nbytes)
    ent = _PERSIST_WS.get(key)
    if ent is None or ent[1] >= (1 << 20) - 2:
        ent = [torch.zeros(nbytes, device=device, dtype=torch.uint8), 0]
        _PERSIST_WS[key] = ent
    ent[1] += 1
    return ent[0], ent[1]


def release_workspaces() -> None:
    """Drop the cached workspaces (they are re-created on the next use)."""
    _WS_CACHE.clear()
    _PERSIST_WS.clear()


def _state_dtype(x: torch.Tensor):
    """The fused steps exist for fp32 and for bf16 state storage (ncahip_*_bf16, see include/ncahip.h)."""
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"ncahip: the NCA state must be float32 or bfloat16, got {x.dtype}")
    return x.dtype, ("bf16" if x.dtype == torch.bfloat16 else "f32")


def _w(t: torch.Tensor, name: str, like: torch.Tensor) -> torch.Tensor:
    """weights: detach, squeeze 1x1 conv dims, move next to the state if needed."""
    t = t.detach()
    if t.device != like.device:
        t = t.to(like.device)
    return _dev(t.float(), name)


def force_generic(on) -> None:
    """Test hook (see ncahip.h): True/1 = generic any-shape kernels, 2 = symmetric wave-private ConditionedNCA
    kernel, False/0 = defaults."""
    lib().ncahip_debug_force_generic(int(on))


def set_cond_precision(mode) -> None:
    """'exact' / 0 (default) or 'bf16x3' / 1: see ncahip_cond_precision in include/ncahip.h."""
    check(lib().ncahip_cond_precision({"exact": 0, "bf16x3": 1}.get(mode, mode)), "cond_precision")


def check_errors(clear: bool = True) -> None:
    """Synchronise the current stream and raise NcaHipError if a kernel recorded a device-side failure (a producer/consumer
    hand-off poll that expired) since the last check -- see ncahip_check_errors in include/ncahip.h."""
    check(lib().ncahip_check_errors(_stream(), int(clear)), "ncahip_check_errors")


def selftest(device=None) -> None:
    scratch = torch.zeros(1024, dtype=torch.int32, device=device or "cuda")
    check(lib().ncahip_selftest(scratch.data_ptr(), _stream()), "ncahip_selftest")


# ------------------------------------------------------------------------------------ stencils
def dynca_perceive(x: torch.Tensor, pad_mode: str = "replicate") -> torch.Tensor:
    x = _dev(x, "x")
    B, C, H, W = x.shape
    y = torch.empty(B, 4 * C, H, W, device=x.device, dtype=torch.float32)
    check(lib().ncahip_dynca_perceive_f32(_p(x), _p(y), B, C, H, W, PAD_MODES[pad_mode], _stream()), "dynca_perceive")
    return y


def cond_perceive(z: torch.Tensor, wp: torch.Tensor) -> torch.Tensor:
    z = _dev(z, "z")
    B, C, H, W = z.shape
    wp = _w(wp, "wp", z)
    assert wp.numel() == 27 * C
    y = torch.empty(B, 3 * C, H, W, device=z.device, dtype=torch.float32)
    check(lib().ncahip_cond_perceive_f32(_p(z), _p(wp), _p(y), B, C, H, W, _stream()), "cond_perceive")
    return y


def image_encoder_front(img: torch.Tensor, k3: torch.Tensor, k5: torch.Tensor) -> torch.Tensor:
    """[sobel_x | sobel_y | laplacian](mean over channels) and the per-channel 5x5 blur in one pass (encoder.py:37-52)."""
    img = _dev(img, "img")
    B, ch, H, W = img.shape
    k3, k5 = _w(k3.reshape(-1), "k3", img), _w(k5.reshape(-1), "k5", img)
    assert k3.numel() == 27 and k5.numel() == 25
    feat = torch.empty(B, 3 + ch, H, W, device=img.device, dtype=torch.float32)
    check(lib().ncahip_image_encoder_front_f32(_p(img), _p(k3), _p(k5), _p(feat), B, ch, H, W, _stream()), "image_encoder_front")
    return feat


def edge_extractor(img: torch.Tensor, k3: torch.Tensor, apply_tanh: bool) -> torch.Tensor:
    """[sobel_x | sobel_y | laplacian] of a 1-channel image, zero pad, optional tanh (dynca.py:204-213)."""
    img = _dev(img, "img")
    B, one, H, W = img.shape
    assert one == 1
    k3 = _w(k3.reshape(-1), "k3", img)
    out = torch.empty(B, 3, H, W, device=img.device, dtype=torch.float32)
    check(lib().ncahip_edge_extractor_f32(_p(img), _p(k3), _p(out), B, H, W, int(apply_tanh), _stream()), "edge_extractor")
    return out


# ------------------------------------------------------------------------------------ fire masks as bits
def mask_words(B: int, H: int, W: int) -> int:
    return (B * H * W + 31) // 32


def pack_fire_mask(u: torch.Tensor, rate: float, mode: str, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """u [T,B,1,H,W] float32 draws -> int32 [T, ceil(B*H*W/32)] bit-packed fire masks with the kernels' own predicate
    (mode 'cond': clamp(u,0,1) < rate, nca.py:171-174; 'dynca': floor(u + rate) = 1, dynca.py:131; include/ncahip.h)."""
    u = _dev(u, "u")
    T, B, _, H, W = u.shape
    bits = torch.empty(T, mask_words(B, H, W), device=u.device, dtype=torch.int32) if out is None else out
    assert bits.shape == (T, mask_words(B, H, W)) and bits.dtype == torch.int32 and bits.is_contiguous()
    check(lib().ncahip_pack_fire_mask_u32(_p(u), _p(bits), T, B, H, W, float(rate), {"cond": 0, "dynca": 1}[mode], _stream()),
          "pack_fire_mask")
    return bits


def draw_fire_masks(B: int, H: int, W: int, steps: int, rate: float, mode: str, device, chunk: int = 16) -> torch.Tensor:
    """The reference's per-step uniform draws (nca.py:172 / dynca.py:131: one [B,1,H,W] float32 draw from the device's global
    torch generator per step, nothing in between) evaluated to bit-packed fire masks int32 [steps, ceil(B*H*W/32)].  The
    draws are the same generator calls in the same order as `torch.rand_like(x[:, 0:1])` per step -- `uniform_()` on a
    [B,1,H,W] float32 tensor IS what rand_like runs -- made into a reused chunk buffer (no [T,B,1,H,W] tensor, no stack
    copy) and packed by ncahip_pack_fire_mask_u32 with the kernels' own predicate."""
    bits = torch.empty(steps, mask_words(B, H, W), device=device, dtype=torch.int32)
    buf = torch.empty(min(chunk, steps), B, 1, H, W, device=device, dtype=torch.float32)
    for t0 in range(0, steps, chunk):
        k = min(chunk, steps - t0)
        for j in range(k):
            buf[j].uniform_()
        pack_fire_mask(buf[:k], rate, mode, out=bits[t0:t0 + k])
    return bits


def unpack_fire_mask(bits: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    """int32 [T, words] -> float32 {0,1} masks [T,B,1,H,W] (torch ops; the composed passes and tests)."""
    T = bits.shape[0]
    idx = torch.arange(B * H * W, device=bits.device)
    m = (bits[:, idx >> 5] >> (idx & 31)) & 1
    return m.view(T, B, 1, H, W).float()


def _u_args(us: Optional[torch.Tensor], T: int, B: int, H: int, W: int, seed: int):
    """(tensor, seed) for the C ABI: float32 uniforms [T,B,1,H,W] as they are; int32 bit-packed masks [T, words] with the seed
    that announces them (NCAHIP_SEED_U_IS_BITS); None: in-kernel Philox keyed by `seed`."""
    if us is None:
        return None, seed
    if us.dtype == torch.int32:
        us = _dev(us, "us", torch.int32)
        assert us.numel() == T * mask_words(B, H, W), (tuple(us.shape), T, B, H, W)
        return us, _capi.SEED_U_IS_BITS
    us = _dev(us, "us")
    assert us.numel() == T * B * H * W
    return us, seed


def philox_uniform(B: int, H: int, W: int, seed: int, step: int, device="cuda") -> torch.Tensor:
    u = torch.empty(B, 1, H, W, device=device, dtype=torch.float32)
    check(lib().ncahip_philox_uniform_f32(_p(u), B, H, W, seed, step, _stream()), "philox_uniform")
    return u


# ------------------------------------------------------------------------------------ DyNCA
class DyncaWeights:
    """w1 [fc,4C+c_cond,1,1], b1 [fc], w2 [C,fc,1,1], b2 [C] as contiguous fp32 device buffers."""

    def __init__(self, w1, b1, w2, b2, like: torch.Tensor):
        self.w1, self.b1 = _w(w1, "w1", like), _w(b1, "b1", like)
        self.w2, self.b2 = _w(w2, "w2", like), _w(b2, "b2", like)
        self.fc, self.k1 = self.w1.shape[0], self.w1.numel() // self.w1.shape[0]
        self.c = self.w2.shape[0]


def dynca_step(x: torch.Tensor, cond: Optional[torch.Tensor], u: Optional[torch.Tensor], w: DyncaWeights,
               pad_mode: str = "replicate", update_rate: float = 0.5, seed: int = 0, step: int = 0,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = _dev(x, "x")
    B, C, H, W = x.shape
    c_cond = 0 if cond is None else cond.shape[1]
    if cond is not None:
        cond = _dev(cond, "cond")
        assert cond.shape == (B, c_cond, H, W)
    if u is not None:
        u = _dev(u, "u")
        assert u.numel() == B * H * W
    assert w.c == C and w.k1 == 4 * C + c_cond, (w.c, w.k1, C, c_cond)
    out = torch.empty_like(x) if out is None else out
    check(lib().ncahip_dynca_step_fwd_f32(_p(x), _p(out), _p(cond), _p(u), _p(w.w1), _p(w.b1), _p(w.w2), _p(w.b2),
                                          B, C, H, W, w.fc, c_cond, PAD_MODES[pad_mode], update_rate, seed, step,
                                          _stream()), "dynca_step_fwd")
    return out


persistent_steps = True     # dynca_nsteps: use the one-launch persistent kernel where it applies (tests / A-B timing switch it off)


def two_scale_fused_ok(C: int, H: int, W: int, fc: int) -> bool:
    """Shapes ncahip_dynca_*_fwd_ms_f32 covers (perception_scales = [0, 1] fused): even sizes, C <= 16, fc <= 128."""
    return H % 2 == 0 and W % 2 == 0 and C <= 16 and fc <= 128


def dynca_nsteps(x: torch.Tensor, T: int, cond: Optional[torch.Tensor], us: Optional[torch.Tensor], w: DyncaWeights,
                 pad_mode: str = "replicate", update_rate: float = 0.5, seed: int = 0, step0: int = 0,
                 keep_history: bool = False, two_scale: bool = False):
    """T fused steps.  Returns (x_T, states) where states is the [ring,B,C,H,W] buffer (ring=T+1 when
    keep_history, else 2).  two_scale: perception_scales = [0, 1] (ncahip_dynca_nsteps_fwd_ms_f32, fp32 states)."""
    dt, sfx = _state_dtype(x)           # bfloat16 state -> bf16-storage entry points (forward only)
    x = _dev(x, "x", dt)
    B, C, H, W = x.shape
    if T == 0:
        return x.clone(), None
    c_cond = 0 if cond is None else cond.shape[1]
    if cond is not None:
        cond = _dev(cond, "cond")
    us, seed = _u_args(us, T, B, H, W, seed)
    assert w.c == C and w.k1 == 4 * C + c_cond, (w.c, w.k1, C, c_cond)
    if not keep_history and sfx == "f32" and persistent_steps:
        # small grids (B = 1 video inference): all T steps in ONE launch, one workgroup per tile (ncahip_dynca_nsteps_fwd_persist_f32);
        # NCAHIP_ERANGE = shape not covered or not every tile resident on this device -> the per-step kernels below
        nbytes = lib().ncahip_dynca_nsteps_persist_workspace(B, C, H, W, w.fc, c_cond)
        if nbytes:
            ws, epoch = _persist_workspace(nbytes, x.device)
            out = torch.empty_like(x)
            fn = lib().ncahip_dynca_nsteps_fwd_persist_ms_f32 if two_scale else lib().ncahip_dynca_nsteps_fwd_persist_f32
            rc = fn(_p(x), _p(out), T, _p(cond), _p(us), _p(w.w1), _p(w.b1), _p(w.w2), _p(w.b2), B, C,
                                                            H, W, w.fc, c_cond, PAD_MODES[pad_mode], update_rate, seed, step0, _p(ws),
                                                            nbytes, epoch, _stream())
            if rc == 0:
                return out, None
            if rc != _capi.ERANGE:
                check(rc, "dynca_nsteps_fwd_persist")
    ring = T + 1 if keep_history else 2
    states = torch.empty(ring, B, C, H, W, device=x.device, dtype=dt)
    states[0].copy_(x)
    if two_scale:
        assert sfx == "f32", "the two-scale step is an fp32 kernel"
        pc = torch.empty(B, 4 * C, H // 2, W // 2, device=x.device, dtype=torch.float32)
        check(lib().ncahip_dynca_nsteps_fwd_ms_f32(_p(states), ring, T, _p(cond), _p(us), _p(w.w1), _p(w.b1), _p(w.w2), _p(w.b2), B, C,
                                                   H, W, w.fc, c_cond, PAD_MODES[pad_mode], update_rate, seed, step0, _p(pc), _stream()),
              "dynca_nsteps_fwd_ms")
        return states[T % ring], states
    check(getattr(lib(), "ncahip_dynca_nsteps_fwd_" + sfx)(_p(states), ring, T, _p(cond), _p(us), _p(w.w1), _p(w.b1), _p(w.w2),
                                                           _p(w.b2), B, C, H, W, w.fc, c_cond, PAD_MODES[pad_mode],
                                                           update_rate, seed, step0, _stream()), "dynca_nsteps_fwd_" + sfx)
    return states[T % ring], states


# ------------------------------------------------------------------------------------ ConditionedNCA
class CondWeights:
    """perception_net.weight [3C,1,3,3]; update_net.out.{0,2,4} weights/biases (nca.py:40-46,99-107)."""

    def __init__(self, wp, w1, b1, w2, b2, w3, like: torch.Tensor):
        self.wp = _w(wp, "wp", like)
        self.w1, self.b1 = _w(w1, "w1", like), _w(b1, "b1", like)
        self.w2, self.b2 = _w(w2, "w2", like), _w(b2, "b2", like)
        self.w3 = _w(w3, "w3", like)
        self.hidden = self.w1.shape[0]
        self.c = self.w3.shape[0]
        assert self.wp.numel() == 27 * self.c and self.w1.numel() == self.hidden * 3 * self.c


def _goal_args(goal, B, C, H, W, dtype=torch.float32):
    if goal is None:
        return None, 0
    goal = _dev(goal, "goal", dtype)
    assert goal.shape[0] == B and goal.shape[2:] == (H, W) and goal.shape[1] <= C
    return goal, goal.shape[1]


def cond_step(x: torch.Tensor, pre_in: Optional[torch.Tensor], goal: Optional[torch.Tensor],
              u: Optional[torch.Tensor], w: CondWeights, alive_ch: int = 3, thr: float = 0.1,
              fire_rate: float = 0.5, lo: float = -10.0, hi: float = 10.0, seed: int = 0, step: int = 0):
    """One fused step; returns (x_pending, pre) -- resolve with cond_finalize (see include/ncahip.h).
    bfloat16 `x` (and `goal`) select the bf16-storage kernel."""
    dt, sfx = _state_dtype(x)
    x = _dev(x, "x", dt)
    B, C, H, W = x.shape
    goal, gch = _goal_args(goal, B, C, H, W, dt)
    if pre_in is not None:
        pre_in = _dev(pre_in, "pre_in", torch.uint8)
    if u is not None:
        u = _dev(u, "u")
        assert u.numel() == B * H * W
    assert w.c == C
    x_out = torch.empty_like(x)
    pre_out = torch.empty(B, H, W, device=x.device, dtype=torch.uint8)
    fn = getattr(lib(), "ncahip_cond_step_fwd_" + sfx)
    check(fn(_p(x), _p(pre_in), _p(x_out), _p(pre_out), _p(goal), gch, _p(u), _p(w.wp), _p(w.w1), _p(w.b1), _p(w.w2),
             _p(w.b2), _p(w.w3), B, C, H, W, w.hidden, alive_ch, thr, fire_rate, lo, hi, seed, step, _stream()),
          "cond_step_fwd_" + sfx)
    return x_out, pre_out


def cond_finalize(x_pend: torch.Tensor, pre: Optional[torch.Tensor], alive_ch: int = 3, thr: float = 0.1,
                  lo: float = -10.0, hi: float = 10.0) -> torch.Tensor:
    dt, sfx = _state_dtype(x_pend)
    x_pend = _dev(x_pend, "x_pend", dt)
    B, C, H, W = x_pend.shape
    if pre is not None:
        pre = _dev(pre, "pre", torch.uint8)
    out = torch.empty_like(x_pend)
    check(getattr(lib(), "ncahip_cond_finalize_" + sfx)(_p(x_pend), _p(pre), _p(out), B, C, H, W, alive_ch, thr, lo, hi,
                                                        _stream()), "cond_finalize_" + sfx)
    return out


def cond_alive(x: torch.Tensor, alive_ch: int = 3, thr: float = 0.1) -> torch.Tensor:
    x = _dev(x, "x")
    B, C, H, W = x.shape
    out = torch.empty(B, 1, H, W, device=x.device, dtype=torch.uint8)
    check(lib().ncahip_cond_alive_u8(_p(x), _p(out), B, C, H, W, alive_ch, thr, _stream()), "cond_alive")
    return out.bool()


def cond_grow(x: torch.Tensor, T: int, goal: Optional[torch.Tensor], us: Optional[torch.Tensor], w: CondWeights,
              alive_ch: int = 3, thr: float = 0.1, fire_rate: float = 0.5, lo: float = -10.0, hi: float = 10.0,
              seed: int = 0, step0: int = 0, keep_history: bool = False):
    """T fused steps + finalize (nca.py:207-208).  Returns (x_T, states, pre).  bfloat16 `x` / `goal` select the
    bf16-storage kernels (forward only)."""
    dt, sfx = _state_dtype(x)
    x = _dev(x, "x", dt)
    B, C, H, W = x.shape
    if T == 0:
        return x.clone(), None, None
    goal, gch = _goal_args(goal, B, C, H, W, dt)
    us, seed = _u_args(us, T, B, H, W, seed)
    assert w.c == C
    ring = T + 1 if keep_history else 2
    states = torch.empty(ring, B, C, H, W, device=x.device, dtype=dt)
    pre = torch.empty(ring, B, H, W, device=x.device, dtype=torch.uint8)
    states[0].copy_(x)
    out = torch.empty_like(x)
    fn = getattr(lib(), "ncahip_cond_grow_fwd_" + sfx)
    check(fn(_p(states), _p(pre), ring, T, _p(out), _p(goal), gch, _p(us), _p(w.wp), _p(w.w1), _p(w.b1), _p(w.w2),
             _p(w.b2), _p(w.w3), B, C, H, W, w.hidden, alive_ch, thr, fire_rate, lo, hi, seed, step0, _stream()),
          "cond_grow_fwd_" + sfx)
    return out, states, pre


def cond_grow_backward(states: torch.Tensor, pre: torch.Tensor, goal: Optional[torch.Tensor], us: Optional[torch.Tensor],
                       w: CondWeights, g_final: torch.Tensor, T: int, alive_ch: int = 3, thr: float = 0.1,
                       fire_rate: float = 0.5, lo: float = -10.0, hi: float = 10.0, seed: int = 0, step0: int = 0):
    """Backward of cond_grow (states/pre = the keep_history=True buffers).  Returns a dict of gradients in the
    reference parameter layouts: x0, goal, wp [3C,9], w1 [hid,3C], b1, w2 [hid,hid], b2, w3 [C,hid].  A bfloat16 history
    (and goal) selects ncahip_cond_grow_bwd_bf16; g_final and every returned gradient are float32 either way."""
    dt, sfx = _state_dtype(states)
    states, pre, g_final = _dev(states, "states", dt), _dev(pre, "pre", torch.uint8), _dev(g_final.float(), "g_final")
    assert states.shape[0] == T + 1 and pre.shape[0] == T + 1
    _, B, C, H, W = states.shape
    goal, gch = _goal_args(goal, B, C, H, W, dt)
    us, seed = _u_args(us, T, B, H, W, seed)
    dev, f32 = states.device, torch.float32
    hid = w.hidden
    g = {"x0": torch.empty(B, C, H, W, device=dev, dtype=f32),
         "goal": torch.empty(B, gch, H, W, device=dev, dtype=f32) if gch else None,
         "wp": torch.empty(3 * C, 9, device=dev, dtype=f32), "w1": torch.empty(hid, 3 * C, device=dev, dtype=f32),
         "b1": torch.empty(hid, device=dev, dtype=f32), "w2": torch.empty(hid, hid, device=dev, dtype=f32),
         "b2": torch.empty(hid, device=dev, dtype=f32), "w3": torch.empty(C, hid, device=dev, dtype=f32)}
    nbytes = lib().ncahip_cond_grow_bwd_workspace(B, C, H, W, hid)
    ws = _workspace(nbytes, dev)
    check(getattr(lib(), "ncahip_cond_grow_bwd_" + sfx)(
        _p(states), _p(pre), T, _p(goal), gch, _p(us), _p(w.wp), _p(w.w1), _p(w.b1), _p(w.w2), _p(w.b2), _p(w.w3), B, C, H, W,
        hid, alive_ch, thr, fire_rate, lo, hi, seed, step0, _p(g_final), _p(g["x0"]), _p(g["goal"]), _p(g["wp"]), _p(g["w1"]),
        _p(g["b1"]), _p(g["w2"]), _p(g["b2"]), _p(g["w3"]), _p(ws), nbytes, _stream()), "cond_grow_bwd_" + sfx)
    return g


def gram_rows(a: torch.Tensor, b1: torch.Tensor, b2: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    """(sum_cells a_i * b_j  [ma, nb],  sum_cells a_i  [ma]) over all B*H*W cells, for a [B,ma,H,W] and b = [b1 | b2] rows
    [B,nb1,H,W] / [B,nb2,H,W]: the weight and bias gradient of a 1x1 conv layer (ncahip_gram_rows_f32).  `out` (a float32
    device vector of ma*nb + ma elements) is ADDED to instead of a fresh result being returned."""
    a, b1 = _dev(a, "a"), _dev(b1, "b1")
    B, ma, H, W = a.shape
    nb1, nb2 = b1.shape[1], 0
    if b2 is not None:
        b2 = _dev(b2, "b2")
        nb2 = b2.shape[1]
    nb = nb1 + nb2
    acc = out is not None
    if acc:
        assert out.is_cuda and out.dtype == torch.float32 and out.numel() == ma * nb + ma and out.is_contiguous()
    else:
        out = torch.empty(ma * nb + ma, device=a.device, dtype=torch.float32)
    nbytes = lib().ncahip_gram_rows_workspace(ma, nb, B, H * W)
    ws = torch.empty(nbytes, device=a.device, dtype=torch.uint8)
    check(lib().ncahip_gram_rows_f32(_p(a), ma, _p(b1), nb1, _p(b2), nb2, B, H * W, _p(out), int(acc), _p(ws), nbytes,
                                     _stream()), "gram_rows")
    return out[:ma * nb].view(ma, nb), out[ma * nb:]


def _gram_fits(ma: int, nb: int) -> bool:
    return (ma <= 32 and nb <= 128) or (ma <= 128 and nb <= 144)


def dynca_nsteps_backward(states: torch.Tensor, cond: Optional[torch.Tensor], us: Optional[torch.Tensor], w: DyncaWeights,
                          g_final: torch.Tensor, g_states: Optional[torch.Tensor], T: int, pad_mode: str = "replicate",
                          update_rate: float = 0.5, seed: int = 0, step0: int = 0, two_scale: bool = False):
    """Backward of dynca_nsteps (states = the keep_history=True buffer [T+1,B,C,H,W]): ONE call into the C driver
    ncahip_dynca_nsteps_bwd_f32, which enqueues the whole T-step loop (perception, fused MLP-backward kernel per 128-wide
    slice of the hidden layer, weight-gradient products with the cell axis as K, stencil adjoint) on the current stream with
    one caller-owned workspace.  g_final must already include any cotangent of x_T itself; g_states (optional, [T+1,...])
    adds dL/dx_t cotangents of the intermediate states t < T (forward_nsteps' return_middle_feature).
    Returns dict x0, w1 [fc,4C+cc], b1, w2 [C,fc], b2."""
    dt, dsfx = _state_dtype(states)      # bfloat16 history: ncahip_dynca_nsteps_bwd_bf16 (storage format only, fp32 gradients)
    states, g = _dev(states, "states", dt), _dev(g_final.float(), "g_final")
    _, B, C, H, W = states.shape
    assert states.shape[0] == T + 1
    c_cond = 0 if cond is None else cond.shape[1]
    if cond is not None:
        cond = _dev(cond, "cond")
    us, seed = _u_args(us, T, B, H, W, seed)
    if g_states is not None:
        g_states = _dev(g_states.float(), "g_states")
        assert g_states.shape == states.shape
    if dsfx == "bf16" and w.fc > 128:
        raise _capi.NcaHipError("ncahip: the bf16-storage DyNCA steps cover fc <= 128")
    dev, f32 = states.device, torch.float32
    fc, k1 = w.fc, 4 * C + c_cond
    out = {"x0": torch.empty(B, C, H, W, device=dev, dtype=f32), "w1": torch.empty(fc, k1, device=dev, dtype=f32),
           "b1": torch.empty(fc, device=dev, dtype=f32), "w2": torch.empty(C, fc, device=dev, dtype=f32),
           "b2": torch.empty(C, device=dev, dtype=f32)}
    sfx = "_ms" if two_scale else ""       # two_scale: backward through ncahip_dynca_nsteps_fwd_ms_f32's steps
    assert not (two_scale and dsfx == "bf16"), "the two-scale step is an fp32 kernel"
    fn = "ncahip_dynca_nsteps_bwd_bf16" if dsfx == "bf16" else f"ncahip_dynca_nsteps_bwd{sfx}_f32"
    wsfn = "ncahip_dynca_nsteps_bwd_bf16_workspace" if dsfx == "bf16" else f"ncahip_dynca_nsteps_bwd{sfx}_workspace"
    nbytes = getattr(lib(), wsfn)(B, C, H, W, fc, c_cond)
    ws = _workspace(nbytes, dev)
    check(getattr(lib(), fn)(_p(states), T, _p(cond), _p(us), _p(w.w1), _p(w.b1), _p(w.w2), _p(w.b2), B, C, H, W, fc,
                                            c_cond, PAD_MODES[pad_mode], update_rate, seed, step0, _p(g), _p(g_states), _p(out["x0"]),
                                            _p(out["w1"]), _p(out["b1"]), _p(out["w2"]), _p(out["b2"]), _p(ws), nbytes, _stream()),
          "dynca_nsteps_bwd" + sfx)
    return out
