#!/usr/bin/env python
"""Experiment: does the last head block -> out_conv pair gain from running per sub-batch, so that the block's output (134 MB for 8 images
against 537 MB for 32) is still in the 256 MB memory-side cache when the 1x1 conv reads it?  Times block(256->256, row-Winograd) + out_conv
(1x1, 256->Cout) for 32 images as 1, 2, 4 and 8 sub-batches, launches back to back on one stream."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
import torch
from centernet_lightning_amd import _lib
from centernet_lightning_amd._lib import CNL_RELU, CNL_SIGMOID, ConvParams

lib = _lib.load()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
N, H, W, C = 32, 128, 128, 256
ams = _lib.absmax_stride()
x = torch.randn(N, H, W, C, device="cuda").clamp_min_(0)
w3 = torch.randn(C, 3, 3, C, device="cuda") * (1.0 / (C * 9)) ** 0.5
b3 = torch.randn(C, device="cuda") * 0.1
u = torch.empty(lib.cnl_winograd_weight_floats(C, C), device="cuda")
_lib.check(lib.cnl_winograd_transform_weights_f32(w3.data_ptr(), u.data_ptr(), C, C, st))
mid = torch.empty(N, H, W, C, device="cuda")
xm = _lib.absmax_pack(x.abs().amax(dim=(1, 2, 3)))
mm = _lib.absmax_buffer(N)
for cout, flags in ((80, CNL_SIGMOID), (4, 0)):
    w1 = torch.randn(cout, 1, 1, C, device="cuda") * 0.01
    b1 = torch.randn(cout, device="cuda")
    w1m = w1.abs().max().reshape(1).contiguous()
    out = torch.empty(N, H, W, cout, device="cuda")

    def params(n0, n):
        p = ConvParams()
        p.x, p.w, p.bias, p.y = x.data_ptr() + 4 * n0 * H * W * C, u.data_ptr(), b3.data_ptr(), mid.data_ptr() + 4 * n0 * H * W * C
        p.N, p.H_in, p.W_in, p.Cin, p.Cout = n, H, W, C, C
        p.KH, p.KW, p.stride, p.pad = 3, 3, 1, 1
        p.ldx, p.ldy, p.ldr, p.flags = C, C, C, CNL_RELU
        p.x_absmax, p.y_absmax = xm.data_ptr() + 4 * n0 * ams, mm.data_ptr() + 4 * n0 * ams
        q = ConvParams()
        q.x, q.w, q.bias, q.y = mid.data_ptr() + 4 * n0 * H * W * C, w1.data_ptr(), b1.data_ptr(), out.data_ptr() + 4 * n0 * H * W * cout
        q.N, q.H_in, q.W_in, q.Cin, q.Cout = n, H, W, C, cout
        q.KH, q.KW, q.stride, q.pad = 1, 1, 1, 0
        q.ldx, q.ldy, q.ldr, q.flags = C, cout, cout, flags
        q.x_absmax, q.w_absmax = mm.data_ptr() + 4 * n0 * ams, w1m.data_ptr()
        return p, q

    ref = None
    for parts in (1, 2, 4, 8, 1):
        n = N // parts
        pq = [params(i * n, n) for i in range(parts)]

        def step():
            mm.zero_()
            for p, q in pq:
                _lib.check(lib.cnl_conv3x3_winograd_f32(ctypes.byref(p), st))
                _lib.check(lib.cnl_conv2d_nhwc_f32(ctypes.byref(q), st))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        same = torch.equal(ref, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        # the block alone, same partition
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
        for _ in range(10):
            mm.zero_()
            for p, q in pq:
                lib.cnl_conv3x3_winograd_f32(ctypes.byref(p), st)
        f1.record()
        torch.cuda.synchronize()
        print(f"Cout {cout:3d}  {parts} x {n:2d} images: block + out_conv {e0.elapsed_time(e1) * 100:8.1f} us   block alone {f0.elapsed_time(f1) * 100:8.1f} us"
              f"   out_conv kernel {lib.cnl_conv2d_kernel(ctypes.byref(pq[0][1]))}   identical to the 1 x 32 result: {same}", flush=True)
