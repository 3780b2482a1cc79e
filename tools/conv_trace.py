#!/usr/bin/env python
"""Debug: per-workgroup phase timestamps of one conv launch (CNL_TRACE_PTR hook of conv_mfma.hip)."""
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
from centernet_lightning_amd._lib import CNL_UPSAMPLE_IN, ConvParams  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "head256"
WINO = len(sys.argv) > 2 and sys.argv[2] == "winograd"
N, H, W, Cin, Cout, k, stride, flags, res = SHAPES[name]
trace = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
lib = _lib.load()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
up = 2 if flags & CNL_UPSAMPLE_IN else 1
Ho, Wo = (H * up + 2 * ((k - 1) // 2) - k) // stride + 1, (W * up + 2 * ((k - 1) // 2) - k) // stride + 1
x = torch.randn(N, H, W, Cin, device="cuda")
w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.02
b = torch.randn(Cout, device="cuda")
y = torch.empty(N, Ho, Wo, Cout, device="cuda")
r = torch.randn(N, Ho, Wo, Cout, device="cuda") if res else None
p = ConvParams()
p.x, p.w, p.bias, p.y = x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr()
p.residual = r.data_ptr() if res else None
p.N, p.H_in, p.W_in, p.Cin, p.Cout = N, H, W, Cin, Cout
p.KH, p.KW, p.stride, p.pad = k, k, stride, (k - 1) // 2
p.ldx, p.ldy, p.ldr, p.flags = Cin, Cout, Cout, flags
fn = lib.cnl_conv2d_nhwc_f32
if WINO:
    u = torch.empty(lib.cnl_winograd_weight_floats(Cin, Cout), device="cuda")
    lib.cnl_winograd_transform_weights_f32(w.data_ptr(), u.data_ptr(), Cin, Cout, stream)
    p.w = u.data_ptr()
    p.flags = flags & 1
    fn = lib.cnl_conv3x3_winograd_f32
for _ in range(2):
    fn(ctypes.byref(p), stream)
torch.cuda.synchronize()
os.environ["CNL_TRACE_PTR"] = str(trace.data_ptr())
fn(ctypes.byref(p), stream)
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()


def us(c):
    return (c - t0) / 100.0          # 100 MHz wall clock


print(f"{name}: {len(t)} workgroups, kernel span {us(t[:, 4].max()):.1f} us")
pro, first = us(t[:, 1]) - us(t[:, 0]), us(t[:, 2]) - us(t[:, 1])
loop, epi = us(t[:, 3]) - us(t[:, 2]), us(t[:, 4]) - us(t[:, 3])
for nm, v in (("prologue", pro), ("first DMA landed", first), ("K loop", loop), ("epilogue(+store drain)", epi)):
    print(f"  {nm:24s} mean {v.mean():8.2f}  p10 {np.percentile(v, 10):8.2f}  p50 {np.percentile(v, 50):8.2f}  "
          f"p90 {np.percentile(v, 90):8.2f} us")
edges = np.linspace(0, us(t[:, 4].max()), 400)
inloop = np.array([np.sum((us(t[:, 2]) <= e) & (us(t[:, 3]) > e)) for e in edges])
print("  workgroups inside the K loop over time (400 samples): mean %.1f  min(mid) %d  frac of samples < 400: %.3f"
      % (inloop.mean(), inloop[5:-5].min(), np.mean(inloop[5:-5] < 400)))
starts = np.sort(us(t[:, 0]))
print("  sorted start times, every 128th of the first 1100:", np.round(starts[:1100:128], 1))
print("  sorted end times, every 512th:", np.round(np.sort(us(t[:, 4]))[511::512][:18], 1))
if t[:, 5].max() > 0:
    ghz = t[:, 5] / ((t[:, 3] - t[:, 1]) / 100e6) / 1e9
    print("  shader clock inside the K loop (clock64 / wall_clock64): mean %.3f GHz  p10 %.3f  p90 %.3f" % (ghz.mean(), np.percentile(ghz, 10), np.percentile(ghz, 90)))
