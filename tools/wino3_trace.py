#!/usr/bin/env python
"""Debug (trace build only): per-wave cycle sums of the phases of winograd3.hip's work items.
    make -C centernet-lightning_amd/csrc trace
    CNL_WINO=3 CENTERNET_GFX950_LIB=tools/_trace/libcenternet_gfx950_w3trace.so python tools/wino3_trace.py head256"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from conv_bench import SHAPES  # noqa: E402
from centernet_lightning_amd import _lib  # noqa: E402
from centernet_lightning_amd._lib import ConvParams  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "head256"
N, H, W, Cin, Cout, k, stride, flags, res = SHAPES[name]
lib = _lib.load()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(N, H, W, Cin, device="cuda")
w = torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.02
b = torch.randn(Cout, device="cuda")
y = torch.empty(N, H, W, Cout, device="cuda")
u = torch.empty(lib.cnl_winograd_weight_floats(Cin, Cout), device="cuda")
lib.cnl_winograd_transform_weights_f32(w.data_ptr(), u.data_ptr(), Cin, Cout, stream)
p = ConvParams()
p.x, p.w, p.bias, p.y = x.data_ptr(), u.data_ptr(), b.data_ptr(), y.data_ptr()
p.N, p.H_in, p.W_in, p.Cin, p.Cout, p.KH, p.KW, p.stride, p.pad = N, H, W, Cin, Cout, 3, 3, 1, 1
p.ldx, p.ldy, p.flags = Cin, Cout, 1
trace = torch.zeros(256 * 4 * 12, dtype=torch.int64, device="cuda")
for _ in range(2):
    assert lib.cnl_conv3x3_winograd_f32(ctypes.byref(p), stream) == 0
torch.cuda.synchronize()
os.environ["CNL_TRACE_PTR"] = str(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert lib.cnl_conv3x3_winograd_f32(ctypes.byref(p), stream) == 0
e1.record()
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(256, 4, 12).astype(np.float64)
t = t[t[:, 0, 7] > 0]
items = t[..., 6]
cc = Cin // 16
names = ["acc zero", "wait patch0 + barrier", "transform chunk 0", "chunk loop", "epilogue", "(mid-chunk wait + barrier, inside loop)"]
print(f"{name}: {e0.elapsed_time(e1) * 1e3:.1f} us, {len(t)} workgroups, items/wave {items.mean():.1f}, chunks/item {cc}")
for i, nm in enumerate(names):
    per_item = (t[..., i] / items).mean()
    print(f"  {nm:45s} {per_item:10.0f} cycles/item   per wave: {np.round((t[..., i] / items).mean(axis=0)).astype(int)}")
for i, nm in ((8, "epilogue: addresses + residual loads"), (9, "epilogue: barrier before stage 1"), (10, "epilogue: stage-1 writes + barrier"), (11, "epilogue: stage 2 (reads, stores)")):
    print(f"  {nm:45s} {(t[..., i] / items).mean():10.0f} cycles/item")
loop = (t[..., 3] / items).mean()
print(f"  chunk loop per chunk: {loop / cc:.0f} cycles (96 MFMAs: ideal 3072) -> {loop / cc / 96:.1f} cycles per MFMA; mid wait per chunk {(t[..., 5] / items).mean() / cc:.0f}")
print(f"  total per item {(t[..., 7] / items).mean():.0f} cycles; clock {t[..., 7].mean() / (e0.elapsed_time(e1) * 1e-3) / 1e9:.2f} GHz (cycles / wall)")
