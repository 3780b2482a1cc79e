#!/usr/bin/env python
"""Per-item phase timing of winograd10_kernel (block 0, thread 0; s_memtime stamps) from the W10_TRACE build: `make -C centernet-lightning_amd/csrc w10trace
W10_TAG=trace`, then `python tools/w10_trace.py [Cin [H=W [Cout]]]` (W10N = batch, W10LIB = library under tools/ablibs)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CENTERNET_GFX950_LIB"] = os.path.join(ROOT, "tools/ablibs/" + os.environ.get("W10LIB", "libcnl_w10trace.so"))
sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
import torch
from centernet_lightning_amd import _lib
from centernet_lightning_amd._lib import CNL_RELU, ConvParams
lib = _lib.load()
arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d
N, Cin, H, Cout = int(os.environ.get("W10N", "32")), arg(1, 256), arg(2, 128), arg(3, 256)
W = H
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(N, H, W, Cin, device="cuda").clamp_min_(0)
w = torch.randn(Cout, 3, 3, Cin, device="cuda") * (1.0 / (Cin * 9)) ** 0.5
b = torch.randn(Cout, device="cuda")
y = torch.empty(N, H, W, Cout, device="cuda")
u = torch.empty(lib.cnl_winograd_weight_floats(Cin, Cout), device="cuda")
_lib.check(lib.cnl_winograd_transform_weights_f32(w.data_ptr(), u.data_ptr(), Cin, Cout, stream))
xm = _lib.absmax_pack(x.abs().amax(dim=(1, 2, 3))); ym = _lib.absmax_buffer(N)
p = ConvParams()
p.x, p.w, p.bias, p.y = x.data_ptr(), u.data_ptr(), b.data_ptr(), y.data_ptr()
p.N, p.H_in, p.W_in, p.Cin, p.Cout, p.KH, p.KW, p.stride, p.pad = N, H, W, Cin, Cout, 3, 3, 1, 1
p.ldx, p.ldy, p.ldr, p.flags, p.algo = Cin, Cout, Cout, CNL_RELU, 110
p.x_absmax, p.y_absmax = xm.data_ptr(), ym.data_ptr()
tr = torch.zeros(64 * 16 + 8 * 1024, dtype=torch.int64, device="cuda")
lib.cnl_w10_set_trace.argtypes = [ctypes.c_void_p]
lib.cnl_w10_set_trace(ctypes.c_void_p(tr.data_ptr()))
for _ in range(3):
    _lib.check(lib.cnl_conv3x3_winograd_f32(ctypes.byref(p), stream))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    lib.cnl_conv3x3_winograd_f32(ctypes.byref(p), stream)
e1.record(); torch.cuda.synchronize()
print(f"Cin {Cin} {H}x{W} Cout {Cout} N {N}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per launch (trace build)")
tb = tr.cpu()[64 * 16:].view(1024, 8)
t = tr.cpu()[:64 * 16].view(64, 16)
for it in range(8):
    r = [int(v) for v in t[it]]
    if not r[7]:
        break
    d = lambda a_, b_: r[b_] - r[a_]
    print(f"item {it:2d} start {r[0] - int(t[0][0]):8d} total {d(0, 7):7d}: requests={d(0, 1)} wait+barrier={d(1, 2)} lds_write+barrier={d(2, 3)} V01={d(3, 4)} chunk0={d(4, 8)} chunk1={d(8, 9)} "
          f"chunks2..={d(9, 5)} barrier={d(5, 6)} epilogue={d(6, 7)}")

import numpy as np
b_ = tb.numpy()
b_ = b_[b_[:, 1] != 0]
if len(b_):
    s0, r0 = b_[:, 0].min(), b_[:, 2].min()
    st, en = b_[:, 0] - s0, b_[:, 1] - s0
    rst, ren = (b_[:, 2] - r0) / 100.0, (b_[:, 3] - r0) / 100.0          # us
    print(f"{len(b_)} workgroups: s_memtime ticks per us = {(en.max()) / ren.max():.1f}; starts 0..{st.max()} ticks ({rst.max():.1f} us), ends {en.min()}..{en.max()} ticks ({ren.min():.1f}..{ren.max():.1f} us), "
          f"lifetime min/med/max = {int((en - st).min())}/{int(np.median(en - st))}/{int((en - st).max())} ticks, items per workgroup {int(b_[:, 6].min())}..{int(b_[:, 6].max())}")
    hw, xcc = b_[:, 4], b_[:, 5] & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
    per = np.bincount(np.unique(cu, return_inverse=True)[1])
    print(f"distinct (xcc, se, sh, cu) = {len(per)}, workgroups per CU: " + ", ".join(f"{k}: {int((per == k).sum())} CUs" for k in sorted(set(per))))
    for q in (0, 1, 2, 3, 255, 256, 257, 511):
        if q < len(b_):
            print(f"  wg {q}: start {int(st[q])} ({rst[q]:.1f} us) end {int(en[q])} ({ren[q]:.1f} us) xcc {int(xcc[q])} hw_id {int(hw[q]):#x}")
