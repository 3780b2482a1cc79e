"""Stem conv (7x7/2, 3 -> 64) alone at the C1 shape: median of 20 launches."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "centernet-lightning_amd"))
from centernet_lightning_amd import _lib
lib = _lib.load()
x = torch.rand(32, 3, 512, 512, device="cuda")
w = torch.randn(64, 7, 7, 3, device="cuda") * 0.1
wp = torch.empty(lib.cnl_stem_packed_weight_floats(), device="cuda")
b = torch.zeros(64, device="cuda")
y = torch.empty(32, 256, 256, 64, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.cnl_stem_pack_weights_f32(w.data_ptr(), wp.data_ptr(), st)
sn, sc, sh, sw = x.stride()
f = lambda: lib.cnl_stem_conv7x7_f32(x.data_ptr(), sn, sc, sh, sw, wp.data_ptr(), b.data_ptr(), y.data_ptr(), None, 32, 512, 512, 0, st)
for _ in range(5): f()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort(); print("stem conv7x7 (no pool) %.1f us" % ts[10])
yp = torch.empty(32, 128, 128, 64, device="cuda")
ym = _lib.absmax_buffer(32)
g = lambda: lib.cnl_stem_conv7x7_maxpool_f32(x.data_ptr(), sn, sc, sh, sw, wp.data_ptr(), b.data_ptr(), yp.data_ptr(), ym.data_ptr(), 32, 512, 512, st)
for _ in range(5): g()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort(); print("stem conv7x7 + maxpool (incl. the zeroing of y's tile seams) %.1f us" % ts[10])
