import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-stylization-with-nca_amd")]
import torch, bench
from ncahip import ops
B, C, H, W = 8, 16, 256, 256
gen = torch.Generator().manual_seed(0)
prm = bench.make_weights(gen)
x = torch.rand(B, C, H, W, generator=gen).cuda()
goal = (torch.randn(B, 12, H, W, generator=gen) * 0.5).cuda()
w = ops.CondWeights(prm["perception_net.weight"], prm["update_net.out.0.weight"], prm["update_net.out.0.bias"],
                    prm["update_net.out.2.weight"], prm["update_net.out.2.bias"], prm["update_net.out.4.weight"], x)
def t(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - w0) * 1e3 / n
for T in (16, 64):
    for kh in (False, True):
        ev, wall = t(lambda: ops.cond_grow(x, T, goal, None, w, 3, seed=1, keep_history=kh))
        print(json.dumps({"T": T, "keep_history": kh, "event_us_per_step": ev * 1e3 / T, "wall_us_per_step": wall * 1e3 / T}), flush=True)
# alive fraction along the way
out, st, pre = ops.cond_grow(x, 16, goal, None, w, 3, seed=1, keep_history=True)
print([round(float(ops.cond_alive(st[i], 3).float().mean()), 3) for i in (0, 1, 2, 4, 8, 16)])
