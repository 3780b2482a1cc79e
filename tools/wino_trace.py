#!/usr/bin/env python
"""Debug (trace build only): per-wave cycle sums of the Winograd K loop — time waiting at the per-chunk barrier (incl. vmcnt(0))
vs time in the chunk body.  Build: see DESIGN.md §3.1 (hipcc -DCNL_WTRACE winograd.hip, linked into tools/_trace/).
    CENTERNET_GFX950_LIB=tools/_trace/libcenternet_gfx950_wtrace.so python tools/wino_trace.py head256"""
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
trace = torch.zeros(256 * 8 * 5, dtype=torch.int64, device="cuda")
for _ in range(2):
    assert lib.cnl_conv3x3_winograd_f32(ctypes.byref(p), stream) == 0
torch.cuda.synchronize()
os.environ["CNL_TRACE_PTR"] = str(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert lib.cnl_conv3x3_winograd_f32(ctypes.byref(p), stream) == 0
e1.record()
torch.cuda.synchronize()
raw = trace.cpu().numpy()
t = raw[:256 * 8 * 4].reshape(256, 8, 4).astype(np.float64)
vm = raw[256 * 8 * 4:].reshape(256, 8).astype(np.float64)
t = t[t[:, 0, 3] > 0]
wait, body, n, total = t[..., 0], t[..., 1], t[..., 2], t[..., 3]
print(f"{name}: {e0.elapsed_time(e1) * 1e3:.1f} us, {len(t)} workgroups, chunks/wave {n.mean():.0f}")
print(f"cycles per chunk: body {np.mean(body / n):.0f} (ideal 2 waves x 32 MFMA x 64 = 4096), barrier+vmcnt wait {np.mean(wait / (n + n / 31)):.0f}")
print(f"share of kernel cycles: body {np.mean(body / total):.3f}  wait {np.mean(wait / total):.3f}  rest (prologue, last chunk, epilogue) {np.mean(1 - (body + wait) / total):.3f}")
print("per-wave mean wait cycles/chunk:", np.round((wait / (n + n / 31)).mean(axis=0)).astype(int))
print("per-wave mean vmcnt(0) wait cycles/chunk (part of the wait):", np.round((vm[:len(t)] / (n + n / 31)).mean(axis=0)).astype(int))
print("per-wave mean body cycles/chunk:", np.round((body / n).mean(axis=0)).astype(int))
print(f"clock: {total.mean() / (e0.elapsed_time(e1) * 1e-3) / 1e9:.2f} GHz (cycles / wall)")
