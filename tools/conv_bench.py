#!/usr/bin/env python
"""Micro-benchmark of cnl_conv2d_nhwc_f32 on one layer shape (HIP events on the launch stream), for rocprofv3
--pmc passes and A/B work on the conv kernel.  Usage: python tools/conv_bench.py [name ...] [--reps R]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
import torch  # noqa: E402
from centernet_lightning_amd import _lib  # noqa: E402
from centernet_lightning_amd._lib import CNL_RELU, CNL_SIGMOID, CNL_UPSAMPLE_IN, ConvParams  # noqa: E402

# name: (N, H, W, Cin, Cout, k, stride, flags, residual)
SHAPES = {
    "head256": (32, 128, 128, 256, 256, 3, 1, CNL_RELU, False),
    "head128": (32, 128, 128, 128, 256, 3, 1, CNL_RELU, False),
    "head512": (32, 128, 128, 512, 256, 3, 1, CNL_RELU, False),
    "headfirst": (32, 64, 64, 64, 512, 3, 1, CNL_RELU | CNL_UPSAMPLE_IN, False),
    "layer1": (32, 128, 128, 64, 64, 3, 1, CNL_RELU, False),
    "fpnfirst": (32, 128, 128, 64, 512, 3, 1, CNL_RELU, False),
    "c4first": (16, 152, 272, 64, 256, 3, 1, CNL_RELU, False),
    "layer1res": (32, 128, 128, 64, 64, 3, 1, CNL_RELU, True),
    "layer2": (32, 64, 64, 128, 128, 3, 1, CNL_RELU, True),
    "l1nr": (32, 128, 128, 64, 64, 3, 1, CNL_RELU, False),
    "l2nr": (32, 64, 64, 128, 128, 3, 1, CNL_RELU, False),
    "layer2n64": (64, 64, 64, 128, 128, 3, 1, CNL_RELU, True),
    "layer2n8": (8, 64, 64, 128, 128, 3, 1, CNL_RELU, True),
    "layer3": (32, 32, 32, 256, 256, 3, 1, CNL_RELU, True),
    "layer4": (32, 16, 16, 512, 512, 3, 1, CNL_RELU, True),
    "layer4n64": (64, 16, 16, 512, 512, 3, 1, CNL_RELU, True),
    "layer4h256": (32, 16, 16, 256, 512, 3, 1, CNL_RELU, True),
    "big256px": (8, 256, 256, 64, 64, 3, 1, CNL_RELU, False),
    "neck0": (32, 16, 16, 512, 256, 3, 1, CNL_RELU, False),
    "c4l4": (16, 19, 34, 512, 512, 3, 1, CNL_RELU, False),
    "c4l3": (16, 38, 68, 256, 256, 3, 1, CNL_RELU, False),
    "c4l1": (16, 152, 272, 64, 64, 3, 1, CNL_RELU, True),
    "c4l2": (16, 76, 136, 128, 128, 3, 1, CNL_RELU, False),
    "c4head": (16, 152, 272, 256, 256, 3, 1, CNL_RELU, False),
    "neckup1": (32, 16, 16, 256, 128, 3, 1, CNL_RELU | CNL_UPSAMPLE_IN, False),
    "neckup2": (32, 32, 32, 128, 64, 3, 1, CNL_RELU | CNL_UPSAMPLE_IN, False),
    "l2s2": (32, 128, 128, 64, 128, 3, 2, CNL_RELU, False),
    "l3s2": (32, 64, 64, 128, 256, 3, 2, CNL_RELU, False),
    "l4s2": (32, 32, 32, 256, 512, 3, 2, CNL_RELU, False),
    "l3down": (32, 64, 64, 128, 256, 1, 2, 0, False),
    "lat512": (32, 16, 16, 512, 256, 1, 1, 0, False),
    "out80": (32, 128, 128, 256, 80, 1, 1, CNL_SIGMOID, False),
    "out4": (32, 128, 128, 256, 4, 1, 1, 0, False),
    "out64c4": (32, 152, 272, 256, 64, 1, 1, 0, False),
    "out80ns": (32, 128, 128, 256, 80, 1, 1, 0, False),
    "out80n64": (64, 128, 128, 256, 80, 1, 1, CNL_SIGMOID, False),
    # one-image batches (BASELINE C0): the latency class
    "n1head": (1, 128, 128, 256, 256, 3, 1, CNL_RELU, False),
    "n1first": (1, 64, 64, 64, 512, 3, 1, CNL_RELU | CNL_UPSAMPLE_IN, False),
    "n1l1": (1, 128, 128, 64, 64, 3, 1, CNL_RELU, True),
    "n1l2": (1, 64, 64, 128, 128, 3, 1, CNL_RELU, True),
    "n1l3": (1, 32, 32, 256, 256, 3, 1, CNL_RELU, True),
    "n1l4": (1, 16, 16, 512, 512, 3, 1, CNL_RELU, True),
    "n1neck0": (1, 16, 16, 512, 256, 3, 1, CNL_RELU, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=["head256"])
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="override the shape's batch size")
    ap.add_argument("--winograd", action="store_true")
    ap.add_argument("--up2", action="store_true", help="3x3 convs with CNL_UPSAMPLE_IN through cnl_conv3x3_up2_nhwc_f32 (sub-pixel phases)")
    ap.add_argument("--hints", action="store_true", help="hand over x_absmax / w_absmax (fp16-split direct kernel where it applies; Winograd: no own absmax pass)")
    ap.add_argument("--algo", type=int, default=0, help="cnl_conv_params.algo: 0 auto, 1 F(2x2) only, 2 fp32 matrix cores, 100+v force Winograd variant v")
    ap.add_argument("--relu-data", action="store_true", help="post-ReLU activations (as inside the network)")
    ap.add_argument("--presplit", action="store_true", help="direct convs: weights with their fp16 split appended (CNL_W_SPLIT)")
    ap.add_argument("--zero-ymax", action="store_true", help="with --hints: zero the y_absmax array before every launch (what a plan does once per forward)")
    ap.add_argument("--no-ymax", action="store_true", help="with --hints: hand over x_absmax only (the kernel reports no max |y|)")
    ap.add_argument("--check", action="store_true", help="also print max |y - y_fp32mfma| / max |y_fp32mfma| (algo 2 on the same inputs)")
    args = ap.parse_args()
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name in args.names:
        N, H, W, Cin, Cout, k, stride, flags, res = SHAPES[name]
        N = args.batch or N
        up = 2 if flags & CNL_UPSAMPLE_IN else 1
        Ho, Wo = (H * up + 2 * ((k - 1) // 2) - k) // stride + 1, (W * up + 2 * ((k - 1) // 2) - k) // stride + 1
        x = torch.randn(N, H, W, Cin, device="cuda")
        if args.relu_data:
            x.clamp_min_(0)
        if os.environ.get("CNL_BENCH_ZEROS") == "1":            # data-dependent power check: all-zero activations
            x.zero_()
        if os.environ.get("CNL_BENCH_ZEROS") == "2":            # ... and all-zero weights too
            x.zero_()
        w = torch.randn(Cout, k, k, Cin, device="cuda") * (1.0 / (Cin * k * k)) ** 0.5
        if os.environ.get("CNL_BENCH_ZEROS") == "2":
            w.zero_()
        b = torch.randn(Cout, device="cuda")
        y = torch.empty(N, Ho, Wo, Cout, device="cuda")
        r = torch.randn(N, Ho, Wo, Cout, device="cuda") if res else None
        p = ConvParams()
        p.x, p.w, p.bias, p.y = x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr()
        p.residual = r.data_ptr() if res else None
        p.N, p.H_in, p.W_in, p.Cin, p.Cout = N, H, W, Cin, Cout
        p.KH, p.KW, p.stride, p.pad = k, k, stride, (k - 1) // 2
        p.ldx, p.ldy, p.ldr, p.flags = Cin, Cout, Cout, flags
        p.algo = args.algo
        fn = lib.cnl_conv2d_nhwc_f32
        if args.winograd:
            if k != 3 or stride != 1:
                continue
            u = torch.empty(lib.cnl_winograd_weight_floats(Cin, Cout), device="cuda")
            _lib.check(lib.cnl_winograd_transform_weights_f32(w.data_ptr(), u.data_ptr(), Cin, Cout, stream))
            p.w = u.data_ptr()
            p.flags = flags & 5            # RELU | UPSAMPLE_IN
            fn = lib.cnl_conv3x3_winograd_f32
        if args.up2:
            if not (flags & CNL_UPSAMPLE_IN) or k != 3:
                continue
            wp = torch.empty(lib.cnl_up2_weight_floats(Cin, Cout), device="cuda")
            _lib.check(lib.cnl_up2_pack_weights_f32(w.data_ptr(), wp.data_ptr(), Cin, Cout, stream))
            p.w = wp.data_ptr()
            p.flags = flags & 5
            fn = lib.cnl_conv3x3_up2_nhwc_f32
            w = wp
        if args.presplit and not args.winograd and not args.up2:
            nfl = lib.cnl_conv_split_weight_floats(Cin, Cout, k, k)
            if nfl:
                wsp = torch.empty(nfl, device="cuda")
                _lib.check(lib.cnl_conv_split_weights_f32(w.data_ptr(), wsp.data_ptr(), Cin, Cout, k, k, stream))
                p.w, p.flags = wsp.data_ptr(), p.flags | _lib.CNL_W_SPLIT
        if args.hints and args.winograd:
            xm = _lib.absmax_pack(x.abs().amax(dim=(1, 2, 3)))
            ym = _lib.absmax_buffer(N)
            p.x_absmax, p.y_absmax = xm.data_ptr(), (None if args.no_ymax else ym.data_ptr())
        if args.hints and not args.winograd:
            xm = _lib.absmax_pack(x.abs().amax(dim=(1, 2, 3)))
            wm = w.abs().max().reshape(1).contiguous()
            ym = _lib.absmax_buffer(N)
            p.x_absmax, p.w_absmax, p.y_absmax = xm.data_ptr(), wm.data_ptr(), (None if args.no_ymax else ym.data_ptr())
        for _ in range(2):
            _lib.check(fn(ctypes.byref(p), stream))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            if args.zero_ymax and args.hints:
                ym.zero_()                   # as inside a plan: the slots start from zero in every forward (nobody's peek finds a maximum yet)
            fn(ctypes.byref(p), stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        flops = 2.0 * N * Ho * Wo * Cout * k * k * Cin
        kern = lib.cnl_conv3x3_winograd_variant(ctypes.byref(p)) if args.winograd else lib.cnl_conv2d_kernel(ctypes.byref(p))
        extra = ""
        if args.check:
            got = y.clone()
            p.algo = 2
            _lib.check(fn(ctypes.byref(p), stream))
            torch.cuda.synchronize()
            extra = f"  max|y-y32|/max|y32| = {float((got - y).abs().max() / y.abs().max()):.2e}  nan={bool(torch.isnan(got).any())}"
        print(f"{name:10s} kernel {kern}  {ms * 1e3:9.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s  ({flops / 1e9:.1f} GFLOP){extra}", flush=True)


if __name__ == "__main__":
    main()
