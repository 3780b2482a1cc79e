#!/usr/bin/env python
"""Per-item phase timing of winograd13_kernel (block 0, thread 0; s_memtime stamps) from the W13_TRACE build: `make -C centernet-lightning_amd/csrc w13trace
W13_TAG=trace`, then `python tools/w13_trace.py [Cin [H=W [Cout]]]` (W13N = batch, W13LIB = library under tools/ablibs)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CENTERNET_GFX950_LIB"] = os.path.join(ROOT, "tools/ablibs/" + os.environ.get("W13LIB", "libcnl_w13trace.so"))
sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
import torch
from centernet_lightning_amd import _lib
from centernet_lightning_amd._lib import CNL_RELU, ConvParams
lib = _lib.load()
arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d
N, Cin, H, Cout = int(os.environ.get("W13N", "32")), arg(1, 256), arg(2, 128), arg(3, 256)
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
p.ldx, p.ldy, p.ldr, p.flags, p.algo = Cin, Cout, Cout, CNL_RELU, 113
p.x_absmax, p.y_absmax = xm.data_ptr(), ym.data_ptr()
tr = torch.zeros(64 * 16 + 8 * 1024, dtype=torch.int64, device="cuda")
lib.cnl_w13_set_trace.argtypes = [ctypes.c_void_p]
lib.cnl_w13_set_trace(ctypes.c_void_p(tr.data_ptr()))
for _ in range(3):
    _lib.check(lib.cnl_conv3x3_winograd_f32(ctypes.byref(p), stream))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    lib.cnl_conv3x3_winograd_f32(ctypes.byref(p), stream)
e1.record(); torch.cuda.synchronize()
print(f"Cin {Cin} {H}x{W} Cout {Cout} N {N}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per launch (trace build)")
t = tr.cpu()[:64 * 16].view(64, 16)
for it in range(10):
    r = [int(v) for v in t[it]]
    if not r[12]:
        break
    d = lambda a_, b_: r[b_] - r[a_]
    ghz = d(0, 12) / max(1, (r[14] - r[13])) / 10.0
    print(f"item {it:2d} start {r[0] - int(t[0][0]):8d} total {d(0, 12):7d} ({ghz:.2f} GHz): scale+barrier={d(0, 1)} lds_write+barrier={d(1, 2)} V01={d(2, 3)} chunk0={d(3, 4)} chunk1={d(4, 5)} "
          f"chunks2..={d(5, 6)} barrier={d(6, 7)} setup={d(7, 8)} pass0={d(8, 9)} pass1+requests={d(9, 10)} pass2={d(10, 11)} pass3+max={d(11, 12)}")
