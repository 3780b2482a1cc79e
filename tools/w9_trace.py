#!/usr/bin/env python
"""Per-item phase timing of winograd9_kernel (block 0, thread 0; s_memtime stamps) from the W9_TRACE build: `make -C centernet-lightning_amd/csrc w9trace`,
then `python tools/w9_trace.py [Cin [H=W [Cout]]]` (W9N = batch).  Prints the phases of six work items and the eight stamps inside a chunk."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CENTERNET_GFX950_LIB"] = os.path.join(ROOT, "tools/ablibs/" + os.environ.get("W9LIB", "libcnl_w9trace.so"))
sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
import torch
from centernet_lightning_amd import _lib
from centernet_lightning_amd._lib import CNL_RELU, ConvParams
lib = _lib.load()
N, H, W, Cin, Cout = int(os.environ.get("W9N", "32")), int(sys.argv[2]) if len(sys.argv) > 2 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 128, int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[3]) if len(sys.argv) > 3 else 256
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
UP = int(os.environ.get("W9UP", "0"))          # 1: the input is stored at H x W and upsampled 2x inside the launch (general form); 2: ... on row-pair weights (cnl_conv_params.w_up)
x = torch.randn(N, H, W, Cin, device="cuda").clamp_min_(0)
w = torch.randn(Cout, 3, 3, Cin, device="cuda") * (1.0 / (Cin * 9)) ** 0.5
b = torch.randn(Cout, device="cuda")
y = torch.empty(N, H * (2 if UP else 1), W * (2 if UP else 1), Cout, device="cuda")
u = torch.empty(lib.cnl_winograd_weight_floats(Cin, Cout), device="cuda")
_lib.check(lib.cnl_winograd_transform_weights_f32(w.data_ptr(), u.data_ptr(), Cin, Cout, stream))
xm = _lib.absmax_pack(x.abs().amax(dim=(1, 2, 3))); ym = _lib.absmax_buffer(N)
p = ConvParams()
p.x, p.w, p.bias, p.y = x.data_ptr(), u.data_ptr(), b.data_ptr(), y.data_ptr()
p.N, p.H_in, p.W_in, p.Cin, p.Cout, p.KH, p.KW, p.stride, p.pad = N, H, W, Cin, Cout, 3, 3, 1, 1
p.ldx, p.ldy, p.ldr, p.flags, p.algo = Cin, Cout, Cout, CNL_RELU | (4 if UP else 0), 109
if UP == 2:
    wu = torch.empty(lib.cnl_winograd_up_weight_floats(Cin, Cout), device="cuda")
    _lib.check(lib.cnl_winograd_transform_weights_up_f32(w.data_ptr(), wu.data_ptr(), Cin, Cout, stream))
    p.w_up = wu.data_ptr()
p.x_absmax, p.y_absmax = xm.data_ptr(), ym.data_ptr()
tr = torch.zeros(64 * 32, dtype=torch.int64, device="cuda")
lib.cnl_w9_set_trace.argtypes = [ctypes.c_void_p]
lib.cnl_w9_set_trace(ctypes.c_void_p(tr.data_ptr()))
for _ in range(3):
    _lib.check(lib.cnl_conv3x3_winograd_f32(ctypes.byref(p), stream))
torch.cuda.synchronize()
t = tr.cpu().view(64, 32)
names = ["setup0", "loop_top", "zeroed", "vmcnt0", "barrier", "preprod", "chunks", "p0.pre", "p0.bar", "p1.pre", "p1.bar", "p2.pre", "p2.bar", "p3.pre", "p3.bar", "end"]
for it in range(6):
    r = t[it]
    order = [1, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]
    nm = {1: "top", 5: "setup+zero", 6: "chunks", 7: "ep.pre", 8: "p0", 9: "p1", 10: "p2", 11: "p3", 12: "p4", 13: "p5", 14: "p6", 15: "p7+end"}
    print(f"item {it:2d} start {int(r[1]-t[0][0]):7d} total {int(r[15]-r[1]):7d}: " + " ".join(f"{nm[order[i]]}={int(r[order[i]] - r[order[i-1]])}" for i in range(1, len(order))))

r = t[0]
print("prologue (block 0, cycles from the kernel's first stamp): requests start %d, patches in LDS %d, after the barrier %d, first loop top %d" % tuple(int(r[i] - r[0]) for i in (2, 3, 4, 1)))
for it in range(1, 5):
    r = t[it]
    if int(r[16]):
        names = os.environ.get("W9PN", "s0 s28 s60 s97 s99(after-barrier) s114 s126 s143").split()
        print(f"item {it} chunk probe: " + " ".join(f"{names[i]}->{names[i+1]}={int(r[17+i]-r[16+i])}" for i in range(7)))
