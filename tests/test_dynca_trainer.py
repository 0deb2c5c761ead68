"""DyNCATrainer (reference loop: ConditioneDyNCA/experiments.py:194-304) -- host logic on CPU with a stub model,
and one real iteration pair on the GPU path."""
import numpy as np
import pytest
import torch
import torch.nn as nn


class StubDyNCA(nn.Module):
    def __init__(self, c=6):
        super().__init__()
        self.c_in, self.c_out = c, 3
        self.gain = nn.Parameter(torch.tensor(0.9))
        self.bias = nn.Parameter(torch.zeros(c))
        self.seen = []

    def seed(self, n, size=(8, 8)):
        return torch.zeros(n, self.c_in, size[1], size[0])

    def forward_nsteps(self, x, step_n, cond_img=None):
        self.seen.append((tuple(x.shape), step_n, None if cond_img is None else tuple(cond_img.shape)))
        y = x * self.gain + self.bias[None, :, None, None] + 0.25
        return y, 2 * y[:, :3]


def test_loop_semantics_cpu():
    from ncahip.dynca_trainer import DyNCATrainer
    m = StubDyNCA()
    tr = DyNCATrainer(m, lambda d: d["generated_image_list"][0].pow(2).mean() + d["nca_state"].abs().mean(), pool_size=10,
                      size=(8, 6), batch_size=3, nca_steps=(4, 9), lr=1e-2, lr_decay_step=(2, 4), inject_seed_step=2,
                      device=torch.device("cpu"))
    assert tr.pool.shape == (10, 6, 6, 8)
    tr.pool += 1.0                                        # make pool entries distinguishable from a fresh seed
    cond = torch.zeros(3, 1, 6, 8)
    for i in range(3):
        np.random.seed(i + 424)                            # replay the reference's draws (experiments.py:196-224)
        idx = np.random.choice(10, 3, replace=False)
        T = int(np.random.randint(4, 9))
        before = tr.pool.clone()
        loss, t_used = tr.step(cond_img=cond)
        assert t_used == T and m.seen[-1] == ((3, 6, 6, 8), T, (3, 1, 6, 8))
        changed = sorted(int(j) for j in range(10) if not torch.equal(before[j], tr.pool[j]))
        assert changed == sorted(int(j) for j in idx)      # only the sampled slots were written back
        assert float(loss) > 0
    for p in m.parameters():
        assert abs(float(p.grad.norm()) - 1.0) < 1e-4      # per-parameter normalisation (:259-263)
    assert abs(tr.optimizer.param_groups[0]["lr"] - 1e-2 * 0.5) < 1e-12   # one milestone (2) passed after 3 steps
    assert tr.iteration == 3


@pytest.mark.gpu
def test_two_iterations_on_gpu():
    from ncahip.dynca_trainer import DyNCATrainer
    from ncahip.models.dynca import DyNCA
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m = DyNCA(12, 3, fc_dim=96, padding_mode="circular", conditioning="edges", edge_transform="tanh", device=dev)
    w0 = m.w1.weight.detach().clone()
    target = torch.rand(1, 3, 32, 32, device=dev)
    tr = DyNCATrainer(m, lambda d: (d["generated_image_list"][0] - target).pow(2).mean(), pool_size=8, size=(32, 32),
                      batch_size=4, nca_steps=(4, 8), lr=1e-3, device=dev)
    cond = torch.rand(4, 1, 32, 32, device=dev) * 2 - 1
    l0, _ = tr.step(cond_img=cond)
    l1, _ = tr.step(cond_img=cond)
    assert torch.isfinite(l0) and torch.isfinite(l1) and not torch.equal(m.w1.weight.detach(), w0)
    assert tr.pool.is_cuda and float(tr.pool.abs().max()) > 0
