import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "centernet-lightning_amd"))
from centernet_lightning_amd import decode as D
N, C, H, W, k = 32, 80, 128, 128, 100
g = torch.Generator(device="cuda").manual_seed(0)
heat = torch.randn(N, H, W, C, device="cuda", generator=g).sub_(2.19).sigmoid_().permute(0, 3, 1, 2)
box = (torch.rand(N, H, W, 4, device="cuda", generator=g) * 16).permute(0, 3, 1, 2)
for layout in ("nhwc", "nchw"):
    h, b = (heat, box) if layout == "nhwc" else (heat.contiguous(), box.contiguous())
    for nms in (1, 3, 5, 7):
        for _ in range(3): D.decode(h, b, None, k, nms)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): D.decode(h, b, None, k, nms)
            e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 20)
        print(layout, "nms", nms, "decode %.4f ms" % best)
