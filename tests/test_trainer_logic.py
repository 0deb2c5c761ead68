"""ConditionedNCATrainer inner-loop semantics (reference conditioned_trainer.py:101-181), pinned with scripted
RNG on CPU.  The NCA is a stub whose grow() is the CPU oracle (tests may use oracle/): the trainer logic is
host code and must not need a GPU."""
import random

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import nca_oracle as O


class StubNCA(nn.Module):
    """Same surface the trainer touches: grow / generate_seed / alive / parameters."""

    def __init__(self, C=8, size=12):
        super().__init__()
        self.num_channels, self.image_size, self.living_channel_dim = C, size, 3
        self.scale = nn.Parameter(torch.tensor(0.5))
        self.shift = nn.Parameter(torch.zeros(C))
        self.unused = nn.Parameter(torch.ones(2))   # never gets a gradient
        self.steps_seen = []

    def generate_seed(self, n):
        return O.cond_generate_seed(n, self.num_channels, 3, self.image_size)

    def alive(self, x):
        return O.cond_alive(x, 3, 0.1)

    def grow(self, x, num_steps, goal):
        self.steps_seen.append(num_steps)
        return x * self.scale + self.shift[None, :, None, None] + 0.01 * goal.mean()


class Targets:
    target_size = (3, 12, 12)

    def __init__(self, n=5):
        self.data = torch.rand(n, 3, 12, 12)
        self.asked = []

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        self.asked.append(list(idx))
        return self.data[idx]


class SimpleLoss(nn.Module):
    def forward(self, d):
        l = (d["generated_images"] - d["target_images"]).pow(2).mean() + 0.1 * d["nca_state"].abs().mean()
        return [l, {"mse": l.detach()}]


def make_trainer(pool_size=16):
    from ncahip.conditioned_trainer import ConditionedNCATrainer
    nca, ds = StubNCA(), Targets()
    tr = ConditionedNCATrainer(nca, ds, None, nca_steps=[3, 6], lr=1e-2, pool_size=pool_size, log_base_path="/tmp/ncahip_test",
                               loss=SimpleLoss(), device=torch.device("cpu"))
    return tr, nca, ds


def test_constructor_signature_matches_reference():
    import inspect
    from ncahip.conditioned_trainer import ConditionedNCATrainer
    names = list(inspect.signature(ConditionedNCATrainer.__init__).parameters)[1:]
    assert names[:16] == ["nca", "target_dataset", "target_style_image", "nca_steps", "lr", "pool_size", "num_damaged",
                          "log_base_path", "damage_radius", "appearance_loss_type", "appearance_loss_weight",
                          "content_loss_weight", "overflow_loss_weight", "device", "visualiser", "loss"]
    tr, _, _ = make_trainer()
    assert tr.min_steps == 3 and tr.max_steps == 6 and tr.rgb and tr.image_size == 12 and tr.pool_size == 16


def test_sample_batch_reseeds_empty_and_dead():
    tr, nca, _ = make_trainer()
    alive = torch.rand(8, 12, 12) + 0.5
    dead = torch.zeros(8, 12, 12)
    tr.pool[[2, 5]] = torch.stack([alive, dead])
    batch = tr.sample_batch([2, 5, 7], tr.pool)
    seed = nca.generate_seed(1)[0]
    assert torch.equal(batch[0], alive)           # alive sample kept
    assert torch.equal(batch[1], seed)            # dead sample (no alpha > 0.1 anywhere) -> fresh seed
    assert torch.equal(batch[2], seed)            # never-written slot -> fresh seed


def test_train_iteration_semantics_scripted_rng():
    tr, nca, ds = make_trainer()
    random.seed(7); np.random.seed(7); torch.manual_seed(7)
    # replay the reference's draws: idxs = random.sample(range(pool), B); T ~ randint(min,max) x2 (python `random`),
    # target rows = np.random.choice(N, B)
    random.seed(7); np.random.seed(7)
    exp_idxs = random.sample(range(16), 4)
    exp_T = [random.randint(3, 6), random.randint(3, 6)]
    exp_targets = list(np.random.choice(5, 4, replace=True))
    random.seed(7); np.random.seed(7)
    w0 = nca.scale.item()
    tr.train(batch_size=4, epochs=1)
    assert nca.steps_seen == exp_T                               # two train_batch calls, T from python random
    assert ds.asked == [exp_targets]                             # ONE target draw per iteration (reused for both calls)
    assert sorted(i for i in range(16) if tr.pool[i] is not None) == sorted(exp_idxs)   # outputs written back
    assert tr.lr_sched.last_epoch == 2                           # scheduler stepped once per train_batch
    assert nca.scale.item() != w0 and nca.unused.grad is None


def test_two_fresh_seeds_and_grad_normalisation():
    tr, nca, _ = make_trainer()
    seen = {}
    orig = tr.train_batch

    def spy(batch, targets):
        seen.setdefault("first", batch.clone())
        out = orig(batch, targets)
        seen["norms"] = [float(p.grad.norm()) for p in nca.parameters() if p.grad is not None]
        return out

    tr.train_batch = spy
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    tr.train(batch_size=4, epochs=1)
    seed = nca.generate_seed(2)
    assert torch.equal(seen["first"][:2], seed)                  # batch[:2] = generate_seed(2) (conditioned_trainer.py:167)
    assert all(abs(n - 1.0) < 1e-4 for n in seen["norms"])       # p.grad /= ||p.grad|| + 1e-10, per parameter


def test_lr_milestone_hits_at_half_the_iterations():
    tr, _, _ = make_trainer()
    for _ in range(4999):
        tr.optimizer.step(); tr.lr_sched.step()
    assert abs(tr.optimizer.param_groups[0]["lr"] - 1e-2) < 1e-12
    tr.optimizer.step(); tr.lr_sched.step()                     # 5000 scheduler steps = 2500 outer iterations
    assert abs(tr.optimizer.param_groups[0]["lr"] - 3e-3) < 1e-12


def test_overflow_loss_is_the_reference_formula():
    from ncahip.loss import Loss
    L = Loss(torch.device("cpu"), content_loss_weight=0.0, appearance_loss_weight=0.0, overflow_loss_weight=2.0)
    s = torch.tensor([[-3.0, -1.0, 0.2, 1.0, 2.5]])
    loss, log = L({"nca_state": s})
    assert abs(float(loss) - 2.0 * (2.0 + 0 + 0 + 0 + 1.5) / 5) < 1e-7 and set(log) == {"overflow"}
    with pytest.raises(ValueError):
        Loss(torch.device("cpu"), appearance_loss_type="bogus", target_style_image=torch.rand(3, 64, 64))


def test_ot_loss_arithmetic():
    """The 'OT' appearance term (appearance_loss.py:149-210) against an independent float64 evaluation of the same
    formulas, including the numpy-stream sub-sampling for maps larger than 32x32."""
    import numpy as np
    import torch
    from ncahip.loss import ot_loss_single
    g = torch.Generator().manual_seed(0)
    tgt = [torch.randn(1, 8, 16, 16, generator=g), torch.randn(1, 12, 40, 40, generator=g)]
    gen = [torch.randn(1, 8, 16, 16, generator=g), torch.randn(1, 12, 40, 40, generator=g)]
    np.random.seed(5)
    got = float(ot_loss_single(tgt, gen))
    np.random.seed(5)
    ref = 0.0
    for t, q in zip(tgt, gen):
        c, h, w = t.shape[1:]
        X, Y = t.reshape(c, -1).double().numpy(), q.reshape(c, -1).double().numpy()
        if h > 32:
            idx = np.sort(np.random.choice(np.arange(h * w), size=1000, replace=False))
            X, Y = X[:, idx], Y[:, idx]
        X, Y = X.T, Y.T
        d = 1.0 - (X @ Y.T) / (np.linalg.norm(X, axis=1)[:, None] + 1e-10) / (np.linalg.norm(Y, axis=1)[None, :] + 1e-10)
        ref += max(d.min(1).mean(), d.min(0).mean())
        ref += np.abs(X.mean(0) - Y.mean(0)).mean() + np.abs(np.cov(X.T) - np.cov(Y.T)).mean()
    assert abs(got - ref) < 1e-5 * max(1.0, abs(ref)), (got, ref)


def test_loss_accepts_the_reference_image_inputs_and_default_trainer_objective():
    """train.py hands the trainer a PIL RGB image (utils.py:28-31 -> transforms.ToTensor): PIL, H x W x C uint8 and C x H x W
    float inputs must all build the default objective, and give the same style tensor."""
    import warnings
    from PIL import Image
    from ncahip.conditioned_trainer import ConditionedNCATrainer
    from ncahip.loss import Loss, _to_nchw
    rgb = (np.random.RandomState(0).rand(36, 40, 3) * 255).astype(np.uint8)
    pil = Image.fromarray(rgb, "RGB")
    a, b = _to_nchw(pil), _to_nchw(rgb)
    c = _to_nchw(torch.from_numpy(rgb).permute(2, 0, 1).float() / 255.0)
    assert a.shape == (1, 3, 36, 40) and torch.equal(a, b) and torch.equal(a, c) and float(a.max()) <= 1.0
    assert _to_nchw(rgb[:, :, 0]).shape == (1, 1, 36, 40)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        nca, ds = StubNCA(), Targets()
        tr = ConditionedNCATrainer(nca, ds, pil, nca_steps=[3, 6], pool_size=8, log_base_path="/tmp/ncahip_test",
                                   device=torch.device("cpu"))          # loss=None: the reference's default objective
        assert isinstance(tr.loss, Loss) and set(tr.loss.loss_weights) == {"overflow", "appearance", "content"}
        gen = torch.rand(2, 3, 40, 40, requires_grad=True)
        np.random.seed(0)
        val, log = tr.loss({"generated_images": gen, "nca_state": torch.rand(2, 8, 40, 40) * 3 - 1.5,
                            "target_images": torch.rand(2, 3, 24, 24)})      # content target of another size: resized
        val.backward()
        assert torch.isfinite(val) and set(log) == {"overflow", "appearance", "content"} and float(gen.grad.abs().max()) > 0


def test_ot_batched_equals_the_per_sample_loop():
    from ncahip.loss import ot_loss_batched, ot_loss_single
    g = torch.Generator().manual_seed(3)
    tgt = [torch.randn(1, 6, 40, 40, generator=g), torch.randn(1, 10, 16, 16, generator=g)]
    gen = [torch.randn(3, 6, 40, 40, generator=g), torch.randn(3, 10, 16, 16, generator=g)]
    np.random.seed(11)
    loop = sum(ot_loss_single(tgt, [f[b:b + 1] for f in gen]) for b in range(3)) / 3
    a = np.random.rand()
    np.random.seed(11)
    bat = ot_loss_batched(tgt, gen)
    b = np.random.rand()
    assert abs(float(loop) - float(bat)) < 1e-5 * max(1.0, abs(float(loop)))
    assert a == b                                                        # same consumption of numpy's global stream


def test_slw_and_gram_follow_the_reference_reductions():
    """'SlW' (appearance_loss.py:109-140): image + 5 levels, squared differences SUMMED; 'Gram' (:98-106): mean of squares."""
    import warnings
    from ncahip.loss import Loss, STYLE_LAYERS, _gram
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        style = torch.rand(3, 32, 32, generator=torch.Generator().manual_seed(1))
        gen = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(2))
        L = Loss(torch.device("cpu"), content_loss_weight=0.0, overflow_loss_weight=0.0, appearance_loss_type="SlW",
                 target_style_image=style)
        torch.manual_seed(5)
        got = float(L({"generated_images": gen, "nca_state": gen})[0])
        torch.manual_seed(5)            # independent evaluation with the same projection draws
        feats = lambda im: [((im - L.vgg.mean) / L.vgg.std).flatten(2)] + [L.vgg(im, STYLE_LAYERS)[l].flatten(2) for l in STYLE_LAYERS]
        ref = 0.0
        for x, y in zip(feats(gen), feats(style[None])):
            pr = torch.nn.functional.normalize(torch.randn(x.shape[1], 32), dim=0)
            xs = torch.einsum("bcn,cp->bpn", x, pr).sort()[0]
            ys = torch.nn.functional.interpolate(torch.einsum("bcn,cp->bpn", y, pr).sort()[0], x.shape[2], mode="nearest")
            ref += float((xs - ys).square().sum())
        assert abs(got - ref) < 1e-4 * abs(ref)
        G = Loss(torch.device("cpu"), content_loss_weight=0.0, overflow_loss_weight=0.0, appearance_loss_type="Gram",
                 target_style_image=style)
        gg = float(G({"generated_images": gen, "nca_state": gen})[0])
        fr = sum(float((_gram(G.style_feats[l]) - _gram(G.vgg(gen, STYLE_LAYERS)[l])).square().mean()) for l in STYLE_LAYERS)
        assert abs(gg - fr) < 1e-5 * abs(fr)


def test_traffic_stamp_hash_matches_bench():
    """bench.py reports `roofline.traffic` only from a profiles/*_traffic.json whose `_meta.csrc_sha16` equals the hash of the kernel
    sources it runs on; tools/collect_traffic.py writes that stamp.  Both must hash the same files the same way (a kernel edit
    without a new PMC pass only warns here: bench.py then reports `traffic: null` and names the stale file)."""
    import glob, hashlib, json, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "video-stylization-with-nca_amd", "csrc", "*"))):   # as tools/collect_traffic.py
        h.update(open(f, "rb").read())
    assert bench.csrc_sha16() == h.hexdigest()[:16]
    stamps = [json.load(open(f)).get("_meta", {}).get("csrc_sha16") for f in glob.glob(os.path.join(root, "profiles", "*_traffic.json"))]
    if bench.csrc_sha16() not in stamps:   # legitimate while kernels are being edited: bench.py then reports `traffic: null`
        import warnings
        warnings.warn("no profiles/*_traffic.json was collected from the current kernel sources: run tools/gpu_round.sh")
