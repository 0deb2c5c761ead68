"""Times the standalone DyNCA perception stencil on buffers that stay resident (same input / output every launch: the warm, cache-assisted
figure; the cold one is `tools/bench_paths.py big`) and prints a checksum of its output, so that two builds of the library can be compared
bit for bit (NCAHIP_LIB selects the build).  usage: python tools/stencil_probe.py [B ...]"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-stylization-with-nca_amd"))
import torch
from ncahip import ops, _capi

L = _capi.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for B in [int(a) for a in sys.argv[1:]] or [8, 64]:
    for (C, H, W, pad) in [(16, 256, 256, 1), (12, 100, 132, 3)] if B == 8 else [(16, 256, 256, 1)]:
        g = torch.Generator(device="cpu").manual_seed(5)
        x = torch.randn(B, C, H, W, generator=g).to(dev)
        y = torch.empty(B, 4 * C, H, W, device=dev)
        run = lambda: ops.check(L.ncahip_dynca_perceive_f32(x.data_ptr(), y.data_ptr(), B, C, H, W, pad, st), "perceive")
        for _ in range(20): run()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 100)
        ms = sorted(ts)[len(ts) // 2]
        gbs = B * H * W * 20 * C / (ms * 1e-3) / 1e9
        h = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
        print(json.dumps({"lib": os.path.basename(os.environ.get("NCAHIP_LIB", "libncahip.so")), "B": B, "C": C, "H": H, "W": W, "pad": pad,
                          "us": round(ms * 1e3, 2), "GBs": round(gbs, 1), "frac": round(gbs / 8000, 3), "sha": h}))
