#!/usr/bin/env python3
"""Timings of every path beside bench.py's headline, one JSON line each (median of >= 10 timed iterations after 3 warm-ups;
HIP events on the launch stream).  `python tools/bench_paths.py [names...]`; without names: all.

  cond_train      ConditionedNCA grow with history + backward, B=8 T=16 (fp32 and bf16 history)
  cond_c20        the reference's DEFAULT ConditionedNCA (C = 20, 16 hidden channels): forward, forward with history + fused backward
  cond_small      the reference's default training shape (train.py: C = 20, 64 x 64, batch 8) and C = 16 at that size: launch-bound regime
  cfg3            BASELINE configs[2] shape: B=32 C=16 256^2 T=96, forward with history + backward, fp32 and bf16
  dynca_fwd       DyNCA forward steps: C=16/fc=128, C=12/fc=96, C=32/fc=128 and C=32/fc=256 at 2x512^2 (configs[4])
  dynca_train     DyNCA forward with history + backward (the C driver): C=16/fc=128, C=12/fc=96, C=32/fc=256 at 2x512^2
  big             working sets beyond the 256 MiB Infinity Cache: perception stencil and fused fp32 step at B=64
  video           B = 1 inference at 256^2 with the shipped video models' shapes, single- and two-scale
  trainer_default ConditionedNCATrainer at the reference's own defaults (C = 20, 64 x 64, batch 8, nca_steps [48, 96]): ms per iteration
  loss            the default objective (VGG16 features + batched OT + content + overflow) at 32 x 3 x 256^2, fp32 / bf16 features
"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-stylization-with-nca_amd")]
import torch

import bench
from ncahip import ops

DEV = "cuda"


def timed(fns, iters=10, warm=3):
    """fns: list of callables run back to back per iteration; returns per-callable median ms (events between them)."""
    for _ in range(warm):
        for f in fns:
            f()
    torch.cuda.synchronize()
    res = [[] for _ in fns]
    for _ in range(iters):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(fns) + 1)]
        ev[0].record()
        for i, f in enumerate(fns):
            f()
            ev[i + 1].record()
        torch.cuda.synchronize()
        for i in range(len(fns)):
            res[i].append(ev[i].elapsed_time(ev[i + 1]))
    return [statistics.median(r) for r in res], [min(r) for r in res]


def emit(**kw):
    print(json.dumps(kw), flush=True)


def wide_weights(C, gen, hidden=64):
    """bench.make_weights' distribution at another channel count"""
    return {"perception_net.weight": torch.randn(3 * C, 1, 3, 3, generator=gen) * 0.3,
            "update_net.out.0.weight": torch.randn(hidden, 3 * C, 1, 1, generator=gen) / (3 * C) ** 0.5,
            "update_net.out.0.bias": torch.randn(hidden, generator=gen) * 0.1,
            "update_net.out.2.weight": torch.randn(hidden, hidden, 1, 1, generator=gen) / hidden ** 0.5,
            "update_net.out.2.bias": torch.randn(hidden, generator=gen) * 0.1,
            "update_net.out.4.weight": torch.randn(C, hidden, 1, 1, generator=gen) * (0.02 / hidden ** 0.5)}


def cond_case(B, H=256, W=256, C=16, dtype=torch.float32):
    gen = torch.Generator().manual_seed(0)
    prm = bench.make_weights(gen) if C == 16 else wide_weights(C, gen)
    x = torch.rand(B, C, H, W, generator=gen).to(DEV, dtype)
    goal = (torch.randn(B, C - 4, H, W, generator=gen) * 0.5).to(DEV, dtype)
    cot = torch.randn(B, C, H, W, generator=gen).to(DEV)
    w = ops.CondWeights(prm["perception_net.weight"], prm["update_net.out.0.weight"], prm["update_net.out.0.bias"],
                        prm["update_net.out.2.weight"], prm["update_net.out.2.bias"], prm["update_net.out.4.weight"], x)
    return x, goal, cot, w


def cond_train(B, T, dtype, name, iters=10, C=16, HW=256):
    x, goal, cot, w = cond_case(B, H=HW, W=HW, C=C, dtype=dtype)
    box = {}

    def fwd():
        box["h"] = ops.cond_grow(x, T, goal, None, w, 3, seed=1, step0=0, keep_history=True)

    def bwd():
        _, states, pre = box.pop("h")
        ops.cond_grow_backward(states, pre, goal, None, w, cot, T, 3, seed=1, step0=0)

    (tf, tb), (mf, mb) = timed([fwd, bwd], iters=iters)
    cells = B * HW * HW * T
    flop = 2 * (27 * C + 256 * C + 4096)      # forward flop per cell-update; the backward is ~2.4x that on MFMA
    emit(path=name, C=C, HW=HW, storage=str(dtype).split(".")[-1], B=B, T=T, fwd_frac_f32_mfma=cells * flop / tf / 1e9 / 157.3, fwd_ms=tf, bwd_ms=tb, fwd_us_per_step=tf / T * 1e3,
         bwd_us_per_step=tb / T * 1e3, fwd_bwd_Gcells_s=cells / (tf + tb) / 1e6, bwd_over_fwd=tb / tf,
         history_GB=(T + 1) * x.numel() * x.element_size() / 1e9, min_fwd_ms=mf, min_bwd_ms=mb)


def dyn_weights(C, fc, cc, gen):
    k1 = 4 * C + cc
    return ops.DyncaWeights(torch.randn(fc, k1, generator=gen) * (0.5 / k1 ** 0.5), torch.randn(fc, generator=gen) * 0.1,
                            torch.randn(C, fc, generator=gen) * (0.02 / fc ** 0.5), torch.zeros(C), torch.zeros(1, device=DEV))


def dynca(B, C, fc, H, W, T, train, cc=3, pad="circular"):
    gen = torch.Generator().manual_seed(0)
    w = dyn_weights(C, fc, cc, gen)
    x = (torch.rand(B, C, H, W, generator=gen) - 0.5).to(DEV)
    cond = (torch.rand(B, cc, H, W, generator=gen) * 2 - 1).to(DEV)
    cot = torch.randn(B, C, H, W, generator=gen).to(DEV)
    flops = 2 * (27 * C + fc * (5 * C + cc))
    cells = B * H * W * T
    if not train:
        (ms,), (mn,) = timed([lambda: ops.dynca_nsteps(x, T, cond, None, w, pad, 0.5, seed=1)])
        emit(path="dynca_fwd", B=B, C=C, fc=fc, HW=[H, W], T=T, us_per_step=ms / T * 1e3, Gcells_s=cells / ms / 1e6,
             TFLOPs=cells * flops / ms / 1e9, frac_f32_mfma=cells * flops / ms / 1e9 / 157.3, min_us_per_step=mn / T * 1e3)
        return
    box = {}

    def fwd():
        box["h"] = ops.dynca_nsteps(x, T, cond, None, w, pad, 0.5, seed=1, keep_history=True)

    def bwd():
        _, states = box.pop("h")
        ops.dynca_nsteps_backward(states, cond, None, w, cot, None, T, pad, 0.5, seed=1)

    (tf, tb), (mf, mb) = timed([fwd, bwd])
    emit(path="dynca_train", B=B, C=C, fc=fc, HW=[H, W], T=T, fwd_us_per_step=tf / T * 1e3, bwd_us_per_step=tb / T * 1e3,
         fwd_bwd_Gcells_s=cells / (tf + tb) / 1e6, bwd_over_fwd=tb / tf, bwd_TFLOPs=3 * cells * flops / tb / 1e9)


def big():
    # stencil at B=64: 67 MB in + 268 MB out per launch, 3 rotating input/output pairs = 1 GB touched between re-uses
    B, C, H, W = 64, 16, 256, 256
    xs = [torch.randn(B, C, H, W, device=DEV) for _ in range(3)]
    k = [0]

    def st():
        ops.dynca_perceive(xs[k[0] % 3], "replicate")
        k[0] += 1
    (ms,), (mn,) = timed([st], iters=12)
    emit(path="stencil_B64", us=ms * 1e3, TBps=320.0 * B * H * W / ms / 1e9, frac_hbm=320.0 * B * H * W / ms / 1e9 / 8.0, min_us=mn * 1e3,
         working_set_MB=3 * (B * C * H * W * 4 * 5) / 1e6)
    # calibration on the same buffers: what this box's HBM delivers for a write-only stream, for a 1:1 copy and for the stencil's own 1 read :
    # 4 writes shape done by torch's elementwise kernel (a broadcast copy) -- the roof the 8 TB/s figure should be read against
    ys = [torch.empty(B, 4, C, H, W, device=DEV) for _ in range(3)]

    def fill():
        ys[k[0] % 3].fill_(1.0)
        k[0] += 1

    def copy11():
        ys[k[0] % 3].copy_(ys[(k[0] + 1) % 3])
        k[0] += 1

    def bcast():
        ys[k[0] % 3].copy_(xs[k[0] % 3].unsqueeze(1))
        k[0] += 1
    (mf, mc, mb), _ = timed([fill, copy11, bcast], iters=12)
    nb = B * C * H * W * 4
    emit(path="hbm_calibration_B64", fill_TBps=4 * nb / mf / 1e9, copy_TBps=8 * nb / mc / 1e9, read1_write4_TBps=5 * nb / mb / 1e9,
         stencil_over_read1_write4=mb / ms)
    del xs, ys
    x, goal, cot, w = cond_case(B)
    T = 8
    (ms,), (mn,) = timed([lambda: ops.cond_grow(x, T, goal, None, w, 3, seed=1)], iters=10)
    cells = B * H * W * T
    emit(path="cond_fwd_B64", us_per_step=ms / T * 1e3, Gcells_s=cells / ms / 1e6, frac_f32_mfma=cells * 17248 / ms / 1e9 / 157.3,
         state_MB=x.numel() * 4 / 1e6)


def loss_leg():
    """The reference's default objective at BASELINE configs[2]'s batch (32 x 3 x 256^2): VGG16 features (seeded random weights:
    the ImageNet weights cannot be fetched here, the WORK is the same) + OT appearance (batched) + content + overflow, forward and
    backward to the generated images, fp32 and bf16 features."""
    import warnings
    import numpy as np
    from ncahip.loss import Loss
    dev = torch.device(DEV)
    style = (np.random.RandomState(0).rand(256, 256, 3) * 255).astype(np.uint8)
    if os.environ.get("NCAHIP_MIOPEN_BENCHMARK") == "1":      # probe: MIOpen's exhaustive find instead of its immediate-mode pick
        torch.backends.cudnn.benchmark = True
    import time as _time
    _t0 = _time.perf_counter()
    for name, dt, cl in (("float32", torch.float32, False), ("bfloat16", torch.bfloat16, False), ("bfloat16 NHWC", torch.bfloat16, True)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            L = Loss(dev, target_style_image=style, feature_dtype=dt, channels_last=cl)
        gen = torch.rand(32, 3, 256, 256, device=dev, requires_grad=True)
        d = {"generated_images": gen, "nca_state": torch.rand(32, 16, 256, 256, device=dev) * 3 - 1.5,
             "target_images": torch.rand(32, 3, 256, 256, device=dev)}

        def f():
            gen.grad = None
            L(d)[0].backward()
        (ms,), (mn,) = timed([f], iters=10)
        emit(path="cfg3_loss", features=name, B=32, ms_fwd_bwd=ms, min_ms=mn, objective="overflow + OT appearance (batched) + content, VGG16 random weights",
             miopen_benchmark=bool(torch.backends.cudnn.benchmark), wall_s_so_far=_time.perf_counter() - _t0)


def trainer_default_leg():
    """ConditionedNCATrainer at the reference's OWN defaults (EncoderConditioning/train.py:30-50: default ConditionedNCA = C 20, 64 x 64
    targets, batch 8, nca_steps [48, 96] drawn per batch, pool 512, lr 2e-3): wall time per trainer iteration (two train_batch calls:
    grow with history, objective, fused backward, per-tensor normalisation, Adam, pool write-back) with the default objective (OT +
    content + overflow on seeded-random VGG16 features) and with a stand-in objective, fp32 and bf16 pool; host overhead included."""
    import tempfile, time, warnings
    import numpy as np
    from ncahip.conditioned_trainer import ConditionedNCATrainer, PhaseTimer
    from ncahip.loss import Loss
    from ncahip.nca import ConditionedNCA
    dev = torch.device(DEV)
    S = 64

    class Targets:
        target_size = (3, S, S)

        def __init__(self):
            self.data = torch.rand(16, 3, S, S, device=dev, generator=torch.Generator(device=dev).manual_seed(3))

        def __len__(self):
            return self.data.shape[0]

        def __getitem__(self, idx):
            return self.data[torch.as_tensor(idx, device=dev)]

    class StandIn(torch.nn.Module):
        def forward(self, d):
            s = d["nca_state"].float()
            return [(d["generated_images"].float() - d["target_images"]).square().mean() + (s - s.clamp(-1.0, 1.0)).abs().mean(), {}]

    style = (np.random.RandomState(0).rand(S, S, 3) * 255).astype(np.uint8)
    for name, dt, default_obj in (("default objective", torch.float32, True), ("stand-in objective", torch.float32, False),
                                  ("stand-in objective", torch.bfloat16, False)):
        torch.manual_seed(0)
        nca = ConditionedNCA(target_shape=(3, S, S)).to(dev)          # the reference's default arguments: C = 20
        if default_obj:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                objective = Loss(dev, target_style_image=style, feature_dtype=torch.bfloat16)
        else:
            objective = StandIn()
        tr = ConditionedNCATrainer(nca, Targets(), None, nca_steps=[48, 96], pool_size=512, loss=objective, device=dev,
                                   log_base_path=tempfile.mkdtemp(prefix="ncahip_bench_"), pool_dtype=dt)
        for i in range(3):
            tr._iteration(i, 8)
        torch.cuda.synchronize()
        tr.phase_timer = PhaseTimer()
        times, steps, phases = [], [], []
        for i in range(12):
            tr.phase_timer.reset()
            s0 = nca._mask_step
            t0 = time.perf_counter()
            tr._iteration(3 + i, 8)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            steps.append(nca._mask_step - s0)
            phases.append(tr.phase_timer.summary())
        med = statistics.median(times)
        per_step = statistics.median([t / max(1, n) for t, n in zip(times, steps)])
        ph = {k: round(float(np.median([p.get(k, 0.0) for p in phases])), 3) for k in sorted({k for p in phases for k in p})}
        emit(path="trainer_default", objective=name, pool=str(dt).split(".")[-1], C=nca.num_channels, HW=S, batch=8, ms_per_iteration=med * 1e3,
             nca_steps_per_iteration_median=statistics.median(steps), wall_us_per_nca_step=per_step * 1e6, phase_ms_per_iteration=ph)
        tr.phase_timer = None
        del tr, objective


def video_leg():
    """B = 1 inference as the video loop issues it (utils/misc/video_utils.py:66-83: forward_nsteps(h, step_n, cond_img=frame) per
    frame) with the shipped video models' shapes (C = 12 / fc = 96 and C = 16 / fc = 128, pos_emb, two-scale perception), 256^2."""
    for C, fc in ((12, 96), (16, 128)):
        for two in (False, True):
            gen = torch.Generator().manual_seed(0)
            w = dyn_weights(C, fc, 2, gen)
            x = (torch.rand(1, C, 256, 256, generator=gen) - 0.5).to(DEV)
            cond = (torch.rand(1, 2, 256, 256, generator=gen) * 2 - 1).to(DEV)
            T = int(os.environ.get("NCAHIP_VIDEO_T", "32"))      # steps per frame (one forward_nsteps call)
            res = {}
            for persist in (True, False):     # the one-launch persistent kernel vs one launch per step
                ops.persistent_steps = persist
                (ms,), (mn,) = timed([lambda: ops.dynca_nsteps(x, T, cond, None, w, "circular", 0.5, seed=1, two_scale=two)], iters=20)
                res["persistent" if persist else "per_step"] = (ms, mn)
            ops.persistent_steps = True
            ops.check_errors()
            ms, mn = res.get("persistent", res["per_step"])
            flops = 2 * (27 * C + fc * (5 * C + 2))
            emit(path="video_B1", C=C, fc=fc, two_scale=two, HW=[256, 256], us_per_step=ms / T * 1e3, min_us_per_step=mn / T * 1e3,
                 steps_per_call=T, frames_per_s_at_32_steps=1e3 / (ms * 32 / T), kernel="persistent (one launch for T steps)" if "persistent" in res else "per-step launches",
                 per_step_launch_us_per_step=res["per_step"][0] / T * 1e3, frac_f32_mfma=256 * 256 * T * flops / ms / 1e9 / 157.3)


def main(names):
    allp = not names
    if allp or "cond_train" in names:
        cond_train(8, 16, torch.float32, "cond_train")
        cond_train(8, 16, torch.bfloat16, "cond_train")
    if allp or "cond_c20" in names:
        cond_train(8, 16, torch.float32, "cond_c20", C=20)
        cond_train(8, 16, torch.bfloat16, "cond_c20", C=20)     # bf16 pool: bf16-MFMA forward, bf16 history, exact-f32 products in the backward
        cond_train(8, 16, torch.float32, "cond_c32", C=32)
    if allp or "cond_small" in names:
        # the reference's own DEFAULT training shape (EncoderConditioning/train.py:36-43: 16 hidden channels -> C = 20, 64 x 64, batch 8)
        cond_train(8, 64, torch.float32, "cond_small_default", C=20, HW=64)
        cond_train(8, 64, torch.float32, "cond_small", C=16, HW=64)
        cond_train(8, 64, torch.bfloat16, "cond_small", C=16, HW=64)
    if allp or "cfg3" in names:
        cond_train(32, 96, torch.float32, "cfg3", iters=10)
        cond_train(32, 96, torch.bfloat16, "cfg3", iters=10)
    if allp or "dynca_fwd" in names:
        dynca(8, 16, 128, 256, 256, 32, False)
        dynca(8, 12, 96, 256, 256, 32, False)
        dynca(2, 32, 128, 512, 512, 16, False)
        dynca(2, 32, 256, 512, 512, 16, False)
    if allp or "dynca_train" in names:
        dynca(8, 16, 128, 256, 256, 16, True)
        dynca(8, 12, 96, 256, 256, 16, True)
        dynca(2, 32, 256, 512, 512, 8, True)
    if allp or "big" in names:
        big()
    if allp or "loss" in names:
        loss_leg()
    if allp or "video" in names:
        video_leg()
    if allp or "trainer_default" in names:
        trainer_default_leg()


if __name__ == "__main__":
    main(sys.argv[1:])
