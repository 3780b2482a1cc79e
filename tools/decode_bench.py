"""Decode micro-benchmark: single-call p50 and GPU time per call; used under rocprofv3 for the per-kernel averages.
python tools/decode_bench.py [c1|c4] [nchw]   (C1: 32 x 80 x 128 x 128, k = 100; C4: 32 x 2 x 152 x 272 + 64-d embeddings, k = 300; nchw: contiguous NCHW maps — the
reference's own layout, stage 1 then runs peaks_generic_kernel — instead of the channel-minor storage the network's out_convs write)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "centernet-lightning_amd"))
from centernet_lightning_amd import decode as D
which = sys.argv[1] if len(sys.argv) > 1 else "c1"
N, C, H, W, k, E = (32, 80, 128, 128, 100, 0) if which == "c1" else (32, 2, 152, 272, 300, 64)
g = torch.Generator(device="cuda").manual_seed(0)
heat = torch.randn(N, H, W, C, device="cuda", generator=g).sub_(2.19).sigmoid_().permute(0, 3, 1, 2)
box = (torch.rand(N, H, W, 4, device="cuda", generator=g) * 16).permute(0, 3, 1, 2)
emb = torch.randn(N, H, W, E, device="cuda", generator=g).permute(0, 3, 1, 2) if E else None
if len(sys.argv) > 2 and sys.argv[2] == "nchw":
    heat, box, emb = heat.contiguous(), box.contiguous(), (emb.contiguous() if emb is not None else None)
for _ in range(3): D.decode(heat, box, emb, k, 3)
torch.cuda.synchronize()
ts = []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); D.decode(heat, box, emb, k, 3); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort(); print("%s decode single-call p50 %.4f ms" % (which, ts[15]))
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): D.decode(heat, box, emb, k, 3)
    e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 50)
print("%s decode GPU time (50 calls back to back, best of 5) %.4f ms" % (which, best))
