"""decode per call over class counts and both layouts (N = 32, 128 x 128, k = 100, nms 3): us and TB/s of the heat map's bytes."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "centernet-lightning_amd"))
from centernet_lightning_amd import decode as D
N, H, W, k = 32, 128, 128, 100
g = torch.Generator(device="cuda").manual_seed(0)
for C in (1, 2, 3, 4, 5, 10, 16, 20, 21, 80, 81, 90):
    heat = torch.randn(N, H, W, C, device="cuda", generator=g).sub_(2.19).sigmoid_().permute(0, 3, 1, 2)
    box = (torch.rand(N, H, W, 4, device="cuda", generator=g) * 16).permute(0, 3, 1, 2)
    out = []
    for layout in ("nhwc", "nchw"):
        h, b = (heat, box) if layout == "nhwc" else (heat.contiguous(), box.contiguous())
        for _ in range(3): D.decode(h, b, None, k, 3)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): D.decode(h, b, None, k, 3)
            e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 20)
        out.append("%s %6.1f us (%4.2f TB/s)" % (layout, best * 1e3, N * C * H * W * 4 / (best * 1e-3) / 1e12))
    print("C = %3d  " % C + "   ".join(out))
