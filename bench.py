#!/usr/bin/env python3
"""bench.py -- NCA cell-updates/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One bench "step" = one pass of the hot path over one batch: ConditionedNCA.grow's loop
(EncoderConditioning/nca.py:207-208) of T=64 fused NCA steps + finalize on BASELINE configs[1]
(B=8, C=16, 256x256, fp32, forward) per GPU, state / goal encoding / weights resident in HBM,
fire mask drawn in-kernel (Philox).  value = N*B*H*W*T*K / t, t = max over ranks of the barrier-
bracketed wall time of the K steps.  Each rank owns an independent shard of the sample pool
(weak scaling, no data-path collective -- SURVEY.md 8e).

Extra objects on the JSON line:
  roofline          fused step kernel (dominant): exact-f32 MFMA bound, algorithmic flops/launch over
                    the HIP-event launch time (events on the launch stream).
  roofline_stencil  standalone DyNCA perception stencil: HBM bound, 20*C bytes/cell.
  bf16_storage      the same grow loop on the bf16-storage kernels (informational; `value` stays the fp32 path).
  backward          the recompute-based backward of the same loop (fp32 / bf16 history), per step (informational).
  default_model_c20 the reference's DEFAULT ConditionedNCA (C = 20) at the same grid: forward, forward with history + backward, fp32 and
                    bf16 pool (informational).
  f32_bf16x3        the same loop with ncahip_cond_precision(1) (opt-in bf16-pair emulation of the fp32 products).
  train             ConditionedNCATrainer iterations at BASELINE configs[2] as written (B=32, T=96, bf16 pool, the default
                    objective on seeded-random VGG16) + stand-in objective lines, with per-phase device times; on every rank
                    (contains the RCCL all-reduce when N > 1).  Informational.
  cpu_baseline      the CPU oracle (pure-PyTorch restatement == the reference's CPU path, bit-identical)
                    timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "video-stylization-with-nca_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

B, C, H, W, T = 8, 16, 256, 256, 64         # BASELINE configs[1]
HIDDEN, GOAL_CH, ALIVE_CH = 64, 12, 3
FLOPS_PER_CELL = 2 * (27 * C + 3 * C * HIDDEN + HIDDEN * HIDDEN + HIDDEN * C)   # 17 248
BYTES_PER_CELL_STEP = 2 * C * 4 + GOAL_CH * 4 + 2                               # x in/out + goal + pre masks
STENCIL_BYTES_PER_CELL = 20 * C                                                 # read C, write 4C floats
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0


def make_weights(gen):
    return {"perception_net.weight": torch.randn(3 * C, 1, 3, 3, generator=gen) * 0.3,
            "update_net.out.0.weight": torch.randn(HIDDEN, 3 * C, 1, 1, generator=gen) / (3 * C) ** 0.5,
            "update_net.out.0.bias": torch.randn(HIDDEN, generator=gen) * 0.1,
            "update_net.out.2.weight": torch.randn(HIDDEN, HIDDEN, 1, 1, generator=gen) / HIDDEN ** 0.5,
            "update_net.out.2.bias": torch.randn(HIDDEN, generator=gen) * 0.1,
            # small last layer: 64 steps perturb rand(0,1) states without killing or saturating them, so the
            # grid stays ~100 % alive and full-range random (no zero-data clock bonus)
            "update_net.out.4.weight": torch.randn(C, HIDDEN, 1, 1, generator=gen) * (0.02 / HIDDEN ** 0.5)}


def event_ms(fn, iters):
    """Average device time of fn() over `iters` back-to-back calls, HIP events on the launch stream."""
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def csrc_sha16():
    """Hash of the kernel sources this run was built from (tools/collect_traffic.py stamps the same hash into its output)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "video-stylization-with-nca_amd", "csrc", "*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(*kernel_substrs):
    """HBM bytes/launch of a kernel from a committed PMC pass (profiles/*_traffic.json: separate `rocprofv3 --pmc FETCH_SIZE` /
    `--pmc WRITE_SIZE` runs of this same command, tools/collect_traffic.py).  PMC collection cannot run inside the timed
    bench, so this is a LOOK-UP, and it is only reported when the file carries the hash of the kernel sources this run was
    built from; otherwise `traffic` is null and the stale file is named.  Returns (bytes or None, provenance dict)."""
    import glob
    sha, hit, stale = csrc_sha16(), None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for k, v in d.items():
            if k != "_meta" and all(sub in k for sub in kernel_substrs):
                if d.get("_meta", {}).get("csrc_sha16") == sha:
                    hit = (v["hbm_bytes_per_launch"], os.path.basename(f))
                else:
                    stale = os.path.basename(f)
    if hit:
        return hit[0], {"source": "profiles/" + hit[1], "csrc_sha16": sha, "measured_in_this_run": False}
    return None, {"source": None, "stale_file": stale and "profiles/" + stale, "csrc_sha16": sha, "measured_in_this_run": False}


def cpu_baseline(prm, x0, goal):
    """Oracle (kind 'port': the reference's PyTorch CPU op sequence) on a bounded sample."""
    from oracle import nca_oracle as O
    gpad = O.cond_pad_goal(goal, C)
    ncpu = os.cpu_count() or 1
    with torch.no_grad():
        # pick the thread count that serves the reference's op sequence best on this host
        # (all cores oversubscribes these small convs badly); then time the bounded sample with it
        best = (None, 1e30)
        for th in sorted({t for t in (8, 16, 32, 64, ncpu) if t <= ncpu}):
            torch.set_num_threads(th)
            O.cond_grow_rng(x0, gpad, 1, prm, ALIVE_CH)
            t0 = time.perf_counter()
            O.cond_grow_rng(x0, gpad, 1, prm, ALIVE_CH)
            d = time.perf_counter() - t0
            if d < best[1]:
                best = (th, d)
        torch.set_num_threads(best[0])
        est = best[1]
        steps = int(min(T, max(2, round(12.0 / max(est, 1e-3)))))
        t0 = time.perf_counter()
        O.cond_grow_rng(x0, gpad, steps, prm, ALIVE_CH)
        dt = time.perf_counter() - t0
    return {"value": B * H * W * steps / dt, "unit": "cell-updates/s", "cores": torch.get_num_threads(),
            "kind": "port", "host_cpus": ncpu, "sample": f"oracle cond_grow, same (B={B},C={C},{H}x{W}) grid, {steps} of {T} NCA steps, "
                                      f"{dt:.1f}s, torch CPU fp32, {torch.get_num_threads()} threads"}


def train_leg(dev, world, iters):
    """ConditionedNCATrainer iterations at BASELINE configs[2] AS WRITTEN, per GPU (= configs[3]'s per-rank shape: 32 grids of
    16 x 256 x 256 per rank, pool sharded over the ranks): sample -> 2 x {grow T = 96 steps with history, objective, backward
    through the fused backward kernels, ONE flat-bucket gradient all-reduce (RCCL when N > 1), per-tensor normalisation, Adam}
    -> pool write-back (conditioned_trainer.py:122-171).  Lines:
      cfg3_bf16_loss     bf16 pool + the reference's default objective (loss/loss.py:61-76: OT appearance + content + overflow)
                         on VGG16 features with seeded-random weights in bf16 (the ImageNet weights cannot be fetched here: the
                         WORK is the same, the loss VALUE is not comparable) -- configs[2];
      bf16_standin       the same loop with a stand-in objective (MSE to the target + overflow): what the NCA kernels cost alone;
      f32_standin        fp32 pool, stand-in objective.
    Each line carries per-phase device times (HIP events on the stream, ms per ITERATION = two train_batch calls): grow forward,
    objective forward+backward, grow backward (+ encoder), all-reduce, normalise + Adam, report (the step's one host sync)."""
    import gc
    import tempfile
    import warnings
    import numpy as np
    from ncahip.conditioned_trainer import ConditionedNCATrainer, PhaseTimer
    from ncahip.loss import Loss
    from ncahip.nca import ConditionedNCA

    class Targets:
        target_size = (3, H, W)

        def __init__(self):
            self.data = torch.rand(8, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(3))

        def __len__(self):
            return self.data.shape[0]

        def __getitem__(self, idx):
            return self.data[torch.as_tensor(idx, device=dev)]

    class StandIn(torch.nn.Module):
        def forward(self, d):
            s = d["nca_state"]
            l = (d["generated_images"] - d["target_images"]).square().mean() + (s - s.clamp(-1.0, 1.0)).abs().mean()
            return [l, {}]

    torch.manual_seed(0)
    nca = ConditionedNCA(target_shape=(3, H, W), num_hidden_channels=C - 4, living_channel_dim=ALIVE_CH).to(dev)
    nca.mask_rng = "philox"
    TB, TT = 32, 96
    style = (np.random.RandomState(0).rand(H, W, 3) * 255).astype(np.uint8)
    out = {}
    for name, dt, default_obj in (("cfg3_bf16_loss", torch.bfloat16, True), ("bf16_standin", torch.bfloat16, False),
                                  ("f32_standin", torch.float32, False)):
        if default_obj:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                objective = Loss(dev, target_style_image=style, feature_dtype=torch.bfloat16)
        else:
            objective = StandIn()
        tr = ConditionedNCATrainer(nca, Targets(), None, nca_steps=[TT, TT], pool_size=2 * TB * world, loss=objective, device=dev,
                                   log_base_path=tempfile.mkdtemp(prefix="ncahip_bench_"), pool_dtype=dt)
        for w_ in range(2):                               # warm-up: allocator (history ring, backward workspace), first touch of
            tr._iteration(0, TB * world)                  # every kernel
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        tr.phase_timer = PhaseTimer()
        times, phases = [], []
        for i in range(iters):                            # every iteration timed on its own (synchronised): the MEDIAN is reported
            tr.phase_timer.reset()
            t0 = time.perf_counter()
            _, _, _, loss, _ = tr._iteration(i + 1, TB * world)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            phases.append(tr.phase_timer.summary())
        tt = torch.tensor(times, device=dev, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)   # an iteration ends when its slowest rank does
        med, mean = float(tt.median().item()), float(tt.mean().item())
        ph = {k: float(np.median([p.get(k, 0.0) for p in phases])) for k in sorted({k for p in phases for k in p})}
        out[name] = {"ms_per_iteration": med * 1e3, "ms_per_iteration_mean": mean * 1e3,
                     "cell_updates_per_s_fwd_bwd": world * TB * H * W * TT * 2 / med, "last_loss": float(loss),
                     "phase_ms_per_iteration": ph, "phase_ms_sum": sum(ph.values())}
        tr.phase_timer = None
        del tr, objective                                 # the next leg's pool / history have other sizes: start it from an empty cache
        from ncahip import ops as _ops
        _ops.release_workspaces()
        gc.collect()
        torch.cuda.empty_cache()
    out.update({"B_per_gpu": TB, "nca_steps": TT, "train_batches_per_iteration": 2, "iterations": iters,
                "allreduce_floats": sum(p.numel() for p in nca.parameters() if p.requires_grad), "n_gpus": world,
                "config": "BASELINE configs[2] per GPU: B=32 C=16 256x256, 96 steps + VGG style loss backprop, bf16 pool",
                "objective": {"cfg3_bf16_loss": "ncahip.loss.Loss: OT appearance (batched) + content + overflow, VGG16 seeded-random "
                                                "weights, bf16 features (loss value not comparable with the reference)",
                              "*_standin": "MSE + overflow"}})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline + rooflines only (profiling passes)")
    ap.add_argument("--train-iters", type=int, default=5, help="iterations of the training-shaped legs (0: skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    # Rehearsal knobs (NOT the measured configuration): NCAHIP_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and NCAHIP_BENCH_BACKEND=gloo
    # replaces RCCL, so the N > 1 protocol (rendezvous, barriers, max over ranks, the trainers' all-reduce) can be run through on a
    # one-GPU box (tools/rehearse_dist.sh).  The JSON line then says so in config.parallelism.
    share_gpu = os.environ.get("NCAHIP_BENCH_SHARE_GPU", "0") == "1"
    backend = os.environ.get("NCAHIP_BENCH_BACKEND", "nccl")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from ncahip import ops
    ops.selftest(dev)

    gen = torch.Generator().manual_seed(0)
    prm = make_weights(gen)
    dgen = torch.Generator().manual_seed(1234 + rank)       # each rank: its own pool shard
    x0 = torch.rand(B, C, H, W, generator=dgen)
    goal = torch.randn(B, GOAL_CH, H, W, generator=dgen) * 0.5
    xd, gd = x0.to(dev), goal.to(dev)
    w = ops.CondWeights(prm["perception_net.weight"], prm["update_net.out.0.weight"], prm["update_net.out.0.bias"],
                        prm["update_net.out.2.weight"], prm["update_net.out.2.bias"], prm["update_net.out.4.weight"], xd)

    # preallocated ring buffers; one bench step == ncahip_cond_grow_fwd_f32 (T launches + finalize)
    states = torch.empty(2, B, C, H, W, device=dev)
    pre = torch.empty(2, B, H, W, device=dev, dtype=torch.uint8)
    out = torch.empty_like(xd)
    states[0].copy_(xd)
    L, st = ops.lib(), torch.cuda.current_stream().cuda_stream
    step_no = [0]

    def one_step():
        ops.check(L.ncahip_cond_grow_fwd_f32(states.data_ptr(), pre.data_ptr(), 2, T, out.data_ptr(), gd.data_ptr(),
                                             GOAL_CH, None, w.wp.data_ptr(), w.w1.data_ptr(), w.b1.data_ptr(),
                                             w.w2.data_ptr(), w.b2.data_ptr(), w.w3.data_ptr(), B, C, H, W, HIDDEN,
                                             ALIVE_CH, 0.1, 0.5, -10.0, 10.0, 42, step_no[0], st), "cond_grow")
        step_no[0] += T

    def one_step_with_pool():
        states[0].copy_(xd)           # pool read: every pass grows the sampled batch for T steps
        one_step()

    def barrier():
        if dist_on:
            dist.barrier()

    for _ in range(args.warmup):
        one_step_with_pool()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        one_step_with_pool()
        evs[i + 1].record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    med_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    if dist_on:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    alive_frac = float(ops.cond_alive(out, ALIVE_CH).float().mean())

    result = None
    if rank == 0:
        value = world * B * H * W * T * args.steps / dt
        # ---- roofline of the dominant kernel: step launches only, events on the launch stream
        # one event-timed region = one grow call (T step launches issued by the C driver + one finalize launch, which is
        # <0.3 % of it and counted against the step: conservative)
        states[0].copy_(xd)
        ms_launch = event_ms(one_step_with_pool, 5) / T   # pool reset per grow (33.5 MB copy, <0.2 % of the region): the grid stays alive
        xa = out
        alive_frac_roof = float(ops.cond_alive(xa, ALIVE_CH).float().mean())
        cells = B * H * W
        tflops = cells * FLOPS_PER_CELL / (ms_launch * 1e-3) / 1e12
        # ---- HBM-bound stencil
        y = torch.empty(B, 4 * C, H, W, device=dev)
        ms_st = event_ms(lambda: ops.check(L.ncahip_dynca_perceive_f32(xd.data_ptr(), y.data_ptr(), B, C, H, W, 1, st), "perceive"), 200)
        gbs = cells * STENCIL_BYTES_PER_CELL / (ms_st * 1e-3) / 1e9
        tr_step = pmc_traffic("cond_step_fwd_pc_kernel", "StF32, false")
        tr_st = pmc_traffic("dynca_perceive_rows_kernel")
        result = {
            "metric": "NCA cell-updates/sec (B*H*W*steps/s) at 256^2 C=16", "value": value, "unit": "cell-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_median": med_ms, "value_from_median": B * H * W * T / (med_ms * 1e-3),   # rank 0's per-step HIP events (median of >= 10)
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: ConditionedNCA grow loop, B=8 C=16 256x256, 64 NCA steps per bench step, fp32 forward",
                       "B_per_gpu": B, "C": C, "H": H, "W": W, "nca_steps_per_bench_step": T, "hidden": HIDDEN,
                       "mask_rng": "in-kernel philox4x32-10", "alive_fraction_at_end": round(alive_frac, 4),
                       "alive_fraction_roofline_run": round(alive_frac_roof, 4),
                       "parallelism": f"pool-shard x{world} (no data-path collective)" +
                                      (f" [REHEARSAL: backend={backend}, share_gpu={int(share_gpu)}]" if (share_gpu or backend != "nccl") else "")},
            "roofline": {"kernel": "cond_step_fwd_pc_kernel<16,*>", "bound": "mfma", "achieved": tflops,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_F32_MFMA_TFLOPS,
                         "traffic": tr_step[0], "traffic_provenance": tr_step[1], "launch_ms": ms_launch, "flops_per_cell": FLOPS_PER_CELL,
                         "algorithmic_bytes_per_cell": BYTES_PER_CELL_STEP, "cells_per_launch": cells},
            "roofline_stencil": {"kernel": "dynca_perceive_rows_kernel<8>", "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS,
                                 "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": tr_st[0], "traffic_provenance": tr_st[1], "launch_ms": ms_st,
                                 "bytes_per_cell": STENCIL_BYTES_PER_CELL, "cells_per_launch": cells,
                                 "cells_per_s": cells / (ms_st * 1e-3)},
        }
        if not args.no_extras:
            # ---- the same grow loop with bf16 state storage (ncahip_cond_grow_fwd_bf16; BASELINE configs[2]'s storage type):
            # reported beside the fp32 headline, never as `value`
            xb16, gb16 = xd.bfloat16(), gd.bfloat16()
            ms_b = event_ms(lambda: ops.cond_grow(xb16, T, gb16, None, w, ALIVE_CH, seed=42), 5) / T
            bpc = 2 * C * 2 + GOAL_CH * 2 + 2
            result["bf16_storage"] = {"kernel": "cond_step_fwd_pc_kernel<16,true,StBF16>", "traffic": pmc_traffic("cond_step_fwd_pc_kernel", "StBF16")[0], "dtype": "bf16 storage, bf16 MFMA, f32 accumulate",
                                      "value": cells / (ms_b * 1e-3), "unit": "cell-updates/s", "launch_ms": ms_b,
                                      "algorithmic_bytes_per_cell": bpc, "hbm_GBs": cells * bpc / (ms_b * 1e-3) / 1e9,
                                      "hbm_frac": cells * bpc / (ms_b * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                      "bound": "vector ALU / LDS issue (staging + perception), see DESIGN.md"}
            # ---- opt-in bf16x3 emulation of the fp32 products (ncahip_cond_precision(1)): fp32 storage, ~1e-5 relative error
            # per step -- informational, never `value`
            ops.set_cond_precision("bf16x3")
            try:
                ms_x = event_ms(lambda: ops.cond_grow(xd, T, gd, None, w, ALIVE_CH, seed=42), 5) / T
                x3, _ = ops.cond_step(xd, None, gd, None, w, ALIVE_CH, seed=7)
            finally:
                ops.set_cond_precision("exact")
            xe, _ = ops.cond_step(xd, None, gd, None, w, ALIVE_CH, seed=7)
            result["f32_bf16x3"] = {"value": cells / (ms_x * 1e-3), "unit": "cell-updates/s", "launch_ms": ms_x,
                                    "max_rel_err_vs_exact_step": float(((x3 - xe).abs() / xe.abs().clamp_min(1.0)).max()),
                                    "dtype": "f32 storage; products as 3 bf16 MFMAs (hi*hi + hi*lo + lo*hi), f32 accumulate"}
            # ---- the same loop started from ConditionedNCA.generate_seed (nca.py:130-150: one live centre cell, most of the grid
            # dead): the kernels have no data-dependent early-out, so this must match `value` (SURVEY.md 8d asks for both)
            xs = torch.zeros_like(xd)
            xs[:, ALIVE_CH:, H // 2, W // 2] = 1.0
            ms_s = event_ms(lambda: ops.cond_grow(xs, T, gd, None, w, ALIVE_CH, seed=42), 3) / T
            result["from_seed"] = {"value": cells / (ms_s * 1e-3), "unit": "cell-updates/s", "launch_ms": ms_s,
                                   "alive_fraction_at_end": float(ops.cond_alive(ops.cond_grow(xs, T, gd, None, w, ALIVE_CH, seed=42)[0], ALIVE_CH).float().mean())}

            # ---- the drop-in classes' DEFAULT mask source (mask_rng='torch': one torch.rand_like per step, the reference's RNG
            # contract, nca.py:172) -- T extra launches per grow, kept as [T, B*H*W/32] words; informational
            ms_t = event_ms(lambda: ops.cond_grow(xd, T, gd, ops.draw_fire_masks(B, H, W, T, 0.5, "cond", dev), w, ALIVE_CH), 5) / T
            result["mask_rng_torch"] = {"value": cells / (ms_t * 1e-3), "unit": "cell-updates/s", "launch_ms": ms_t,
                                        "note": "the drop-in classes' default mask source: one torch uniform draw per step (the reference's "
                                                "generator calls, nca.py:172), evaluated to bit-packed fire masks (ncahip_pack_fire_mask_u32)"}
            # ---- the recompute-based backward of the same loop at the same shape (ncahip_cond_grow_bwd_f32 / _bf16: 16 steps with
            # history, cotangent of the final state; fp32 history with exact-f32 products, bf16 history with bf16-MFMA products):
            # informational -- the kernels the `train` leg spends its time in, without the trainer around them
            TB_ = 16
            cot = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(11)).to(dev)
            result["backward"] = {"nca_steps": TB_}
            for name, xq, gq in (("f32", xd, gd), ("bf16", xb16, gb16)):
                _, st_h, pre_h = ops.cond_grow(xq, TB_, gq, None, w, ALIVE_CH, seed=42, keep_history=True)
                ms_f = event_ms(lambda: ops.cond_grow(xq, TB_, gq, None, w, ALIVE_CH, seed=42, keep_history=True), 5) / TB_
                ms_w = event_ms(lambda: ops.cond_grow_backward(st_h, pre_h, gq, None, w, cot, TB_, ALIVE_CH, seed=42), 5) / TB_
                result["backward"][name] = {"fwd_us_per_step": ms_f * 1e3, "bwd_us_per_step": ms_w * 1e3, "bwd_over_fwd": ms_w / ms_f,
                                            "fwd_bwd_cell_updates_per_s": cells / ((ms_f + ms_w) * 1e-3)}
                del st_h, pre_h
            # ---- the reference's DEFAULT ConditionedNCA (nca.py:62-94: 3 rgb + 1 alpha + 16 hidden = 20 channels) at the bench grid:
            # forward (producer/consumer kernel, wide LDS carve), forward with history + fused backward, fp32 and bf16 pool: informational
            C20, G20 = 20, 16
            g20 = torch.Generator().manual_seed(20)
            p20 = {"wp": torch.randn(3 * C20, 1, 3, 3, generator=g20) * 0.3, "w1": torch.randn(HIDDEN, 3 * C20, 1, 1, generator=g20) / (3 * C20) ** 0.5,
                   "b1": torch.randn(HIDDEN, generator=g20) * 0.1, "w2": torch.randn(HIDDEN, HIDDEN, 1, 1, generator=g20) / HIDDEN ** 0.5,
                   "b2": torch.randn(HIDDEN, generator=g20) * 0.1, "w3": torch.randn(C20, HIDDEN, 1, 1, generator=g20) * (0.02 / HIDDEN ** 0.5)}
            x20 = torch.rand(B, C20, H, W, generator=g20).to(dev)
            gl20 = (torch.randn(B, G20, H, W, generator=g20) * 0.5).to(dev)
            cot20 = torch.randn(B, C20, H, W, generator=g20).to(dev)
            w20 = ops.CondWeights(p20["wp"], p20["w1"], p20["b1"], p20["w2"], p20["b2"], p20["w3"], x20)
            flop20 = 2 * (27 * C20 + 256 * C20 + 4096)
            result["default_model_c20"] = {"C": C20, "nca_steps": TB_, "flops_per_cell": flop20}
            for name, xq, gq in (("f32", x20, gl20), ("bf16", x20.bfloat16(), gl20.bfloat16())):
                ms_i = event_ms(lambda: ops.cond_grow(xq, T, gq, None, w20, ALIVE_CH, seed=42), 3) / T
                _, st_h, pre_h = ops.cond_grow(xq, TB_, gq, None, w20, ALIVE_CH, seed=42, keep_history=True)
                ms_f = event_ms(lambda: ops.cond_grow(xq, TB_, gq, None, w20, ALIVE_CH, seed=42, keep_history=True), 3) / TB_
                ms_w = event_ms(lambda: ops.cond_grow_backward(st_h, pre_h, gq, None, w20, cot20, TB_, ALIVE_CH, seed=42), 3) / TB_
                result["default_model_c20"][name] = {"fwd_us_per_step": ms_i * 1e3, "fwd_cell_updates_per_s": cells / (ms_i * 1e-3),
                                                     "fwd_frac_f32_mfma": cells * flop20 / (ms_i * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS if name == "f32" else None,
                                                     "fwd_with_history_us_per_step": ms_f * 1e3, "bwd_us_per_step": ms_w * 1e3,
                                                     "fwd_bwd_cell_updates_per_s": cells / ((ms_f + ms_w) * 1e-3)}
                del st_h, pre_h
            del x20, gl20, cot20
    # ---- training-shaped leg on EVERY rank (the path that contains the gradient all-reduce when N > 1): informational
    train = train_leg(dev, world, args.train_iters) if (args.train_iters > 0 and not args.no_extras) else None
    if rank == 0:
        if train is not None:
            result["train"] = train
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(prm, x0, goal)
            result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
        print(json.dumps(result), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
