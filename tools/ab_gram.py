#!/usr/bin/env python3
"""Same-process A/B of gram_rows_kernel's two row-sum forms (NCAHIP_GRAM_ONES=0/1), interleaved rounds (guide rule 24)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-stylization-with-nca_amd")]
import torch
from ncahip import ops
for (ma, nb1, nb2, B, H, W) in ((128, 64, 3, 8, 256, 256), (96, 48, 3, 8, 256, 256), (128, 128, 3, 2, 512, 512)):
    a = torch.randn(B, ma, H, W, device="cuda"); b1 = torch.randn(B, nb1, H, W, device="cuda"); b2 = torch.randn(B, nb2, H, W, device="cuda")
    res = {"0": [], "1": []}
    for rnd in range(6):
        for mode in ("0", "1"):
            os.environ["NCAHIP_GRAM_ONES"] = mode
            for _ in range(3): ops.gram_rows(a, b1, b2)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.gram_rows(a, b1, b2)
            e1.record(); torch.cuda.synchronize()
            res[mode].append(e0.elapsed_time(e1) / 10 * 1e3)
    print("ma=%d nb=%d  %dx%dx%d:  adds %.1f us   ones-column %.1f us (median of 6 interleaved rounds, incl. the reduce launch)" %
          (ma, nb1 + nb2, B, H, W, statistics.median(res["0"]), statistics.median(res["1"])))
