"""Decode micro-benchmark at C1 (32 x 80 x 128 x 128, k = 100): single-call p50 and GPU time per call; used under rocprofv3 for the per-kernel averages."""
import sys, os, torch
sys.path.insert(0, "centernet-lightning_amd")
from centernet_lightning_amd import decode as D
g = torch.Generator(device="cuda").manual_seed(0)
heat = torch.randn(32,128,128,80, device="cuda", generator=g).sub_(2.19).sigmoid_().permute(0,3,1,2)
box = (torch.rand(32,128,128,4, device="cuda", generator=g)*16).permute(0,3,1,2)
for _ in range(3): D.decode(heat, box, None, 100, 3)
torch.cuda.synchronize()
ts=[]
for _ in range(30):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); D.decode(heat, box, None, 100, 3); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort(); print("decode single-call p50 %.4f ms" % ts[15])
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): D.decode(heat, box, None, 100, 3)
e1.record(); torch.cuda.synchronize(); print("decode GPU time (50 calls back to back) %.4f ms" % (e0.elapsed_time(e1)/50))
