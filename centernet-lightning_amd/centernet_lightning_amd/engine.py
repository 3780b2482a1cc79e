"""HIP launch plan for the CenterNet forward: backbone -> neck -> heads (reference models/meta.py:41-47).

Python here is orchestration only: it folds eval-mode BatchNorm into OHWI weights once per weight load
(device-side torch ops), allocates NHWC activation buffers per input shape, and replays a flat list of
C-ABI launches (include/centernet_gfx950.h) on torch's current HIP stream.  All arithmetic of the forward
happens inside libcenternet_gfx950.so.

Fusions relative to the reference's op-per-layer graph (results identical up to fp32 summation order):
  * conv + BN + ReLU (+ residual add)                     -> one launch
  * nn.Upsample(nearest, x2) feeding a conv              -> CNL_UPSAMPLE_IN gather (never materialised)
  * Fuse: project(top) -> upsample -> + skip              -> CNL_UPSAMPLE_OUT_ADD epilogue of the 1x1 conv
  * first 3x3 block of every head (same input, meta.py:46) -> one conv with concatenated Cout
  * heatmap .sigmoid() (centernet.py:205)                 -> epilogue of the heatmap out_conv
"""
import ctypes
import os
from collections import OrderedDict

import torch

from . import _lib
from ._lib import CNL_RELU, CNL_SIGMOID, CNL_UPSAMPLE_IN, CNL_UPSAMPLE_OUT_ADD, ConvParams

BN_EPS_DEFAULT = 1e-5


def fold_conv_bn(conv_w, conv_b, bn=None):
    """-> (OHWI weight [Cout,KH,KW,Cin] contiguous fp32, bias [Cout]) with eval-mode BN folded in:
    y = (conv(x) + b - mean) * gamma / sqrt(var + eps) + beta."""
    w = conv_w.detach().to(torch.float32)
    cout = w.shape[0]
    b = conv_b.detach().to(torch.float32) if conv_b is not None else torch.zeros(cout, device=w.device)
    if bn is not None:
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        w = w * scale.view(-1, 1, 1, 1)
        b = (b - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()
    return w.permute(0, 2, 3, 1).contiguous(), b.contiguous()


def winograd_enabled():
    """3x3 / stride-1 layers run through the Winograd F(2x2,3x3) kernel unless CNL_WINOGRAD=0 (then every conv takes the
    direct implicit-GEMM kernel; same results up to fp32 rounding)."""
    return os.environ.get("CNL_WINOGRAD", "1") != "0"


class _Layer:
    """One packed conv layer: folded OHWI weight + bias on the device (+ the Winograd-transformed weight for
    3x3 / stride-1 layers, produced once by cnl_winograd_transform_weights_f32)."""

    def __init__(self, w_ohwi, bias, stride=1):
        self.w, self.b = w_ohwi, bias
        self.cout, self.kh, self.kw, self.cin = w_ohwi.shape
        self.stride = stride
        self.pad = (self.kh - 1) // 2
        self.u = None
        if self.kh == 3 and self.kw == 3 and stride == 1 and self.cin % 8 == 0 and w_ohwi.is_cuda and winograd_enabled():
            lib = _lib.load()
            n = lib.cnl_winograd_weight_floats(self.cin, self.cout)
            with torch.cuda.device(w_ohwi.device):
                self.u = torch.empty((n,), device=w_ohwi.device, dtype=torch.float32)
                stream = ctypes.c_void_p(torch.cuda.current_stream(w_ohwi.device).cuda_stream)
                _lib.check(lib.cnl_winograd_transform_weights_f32(w_ohwi.data_ptr(), self.u.data_ptr(), self.cin, self.cout, stream),
                           "cnl_winograd_transform_weights_f32")


class PackedWeights:
    """Device-resident, BN-folded weights of a CenterNet model (rebuilt whenever parameters change)."""

    def __init__(self, model, device):
        bb, neck, heads = model.backbone, model.neck, model.heads
        dev = lambda t: t.to(device)
        L = lambda conv, bn=None, stride=None: _Layer(*map(dev, fold_conv_bn(conv.weight, conv.bias, bn)),
                                                      stride=stride if stride is not None else conv.stride[0])
        self.stem = L(bb.conv1, bb.bn1)
        # the stem kernel DMAs its weights into LDS verbatim: pack OHWI -> [148][64] once (cnl_stem_pack_weights_f32)
        lib = _lib.load()
        with torch.cuda.device(device):
            self.stem_packed = torch.empty((148 * 64,), device=device, dtype=torch.float32)
            _lib.check(lib.cnl_stem_pack_weights_f32(self.stem.w.data_ptr(), self.stem_packed.data_ptr(),
                                                     ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                       "cnl_stem_pack_weights_f32")
        self.blocks = []
        for li in range(4):
            for blk in getattr(bb, f"layer{li + 1}"):
                d = L(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
                self.blocks.append((L(blk.conv1, blk.bn1), L(blk.conv2, blk.bn2), d, li))
        self.neck_kind = type(neck).__name__
        if self.neck_kind == "SimpleNeck":
            self.neck_layers = [L(m.conv_module, m.bn_module) for m in neck.layers]
        else:
            self.top = L(neck.top_conv)
            self.fuse = []
            for f in neck.fuse:
                skip_p = L(f.project[0]) if isinstance(f.project[0], torch.nn.Conv2d) else None
                if isinstance(f.project[1], torch.nn.Conv2d):
                    top_p = L(f.project[1])
                else:       # no projection in the reference: an exact identity 1x1 keeps the fused epilogue path
                    c = f.output_conv.conv_module.in_channels
                    eye = torch.eye(c, device=device, dtype=torch.float32).view(c, 1, 1, c).contiguous()
                    top_p = _Layer(eye, torch.zeros(c, device=device))
                self.fuse.append((skip_p, top_p, L(f.output_conv.conv_module, f.output_conv.bn_module)))
        # heads: first blocks fused along Cout when every head has depth >= 1
        self.head_names = list(heads.keys())
        self.head_blocks = OrderedDict()
        self.head_out = OrderedDict()
        for name, h in heads.items():
            self.head_blocks[name] = [L(m.conv_module, m.bn_module) for m in h.blocks()]
            self.head_out[name] = L(h.out_conv)
        self.fused_first = None
        if all(len(b) >= 1 for b in self.head_blocks.values()) and len(self.head_names) > 1:
            w = torch.cat([self.head_blocks[n][0].w for n in self.head_names], dim=0).contiguous()
            b = torch.cat([self.head_blocks[n][0].b for n in self.head_names], dim=0).contiguous()
            self.fused_first = _Layer(w, b)


class _Launch:
    __slots__ = ("fn", "args", "what", "flops", "keep")

    def __init__(self, fn, args, what, flops=0, keep=()):
        self.fn, self.args, self.what, self.flops, self.keep = fn, args, what, flops, keep


class Plan:
    """Activation buffers + launch list for one (N, H, W, input strides, sigmoid) signature."""

    def __init__(self, weights: PackedWeights, N, H, W, device, sigmoid):
        self.lib = _lib.load()
        self.N, self.H, self.W, self.device, self.sigmoid = N, H, W, device, sigmoid
        self.launches = []
        self.buffers = []
        self.outputs = OrderedDict()      # name -> (NHWC buffer tensor [N,h,w,C])
        self.out_params = {}              # name -> ConvParams writing that output (y patched per call)
        self.stem_args = None
        self._build(weights)

    # -- helpers --
    def _buf(self, n, h, w, c):
        t = torch.empty((n, h, w, c), device=self.device, dtype=torch.float32)
        self.buffers.append(t)
        return t

    def _conv(self, layer, x, xh, xw, ldx, y, ldy, flags=0, residual=None, ldr=0, what="conv", x_off=0, y_off=0):
        """Append one cnl_conv2d_nhwc_f32 launch. x / y / residual are NHWC buffer tensors (or views sharing
        storage, addressed by element offsets x_off / y_off inside the pixel)."""
        p = ConvParams()
        p.x = x.data_ptr() + 4 * x_off
        p.w = layer.w.data_ptr()
        p.bias = layer.b.data_ptr()
        p.residual = residual.data_ptr() if residual is not None else None
        p.y = y.data_ptr() + 4 * y_off
        p.N, p.H_in, p.W_in, p.Cin, p.Cout = self.N, xh, xw, layer.cin, layer.cout
        p.KH, p.KW, p.stride, p.pad = layer.kh, layer.kw, layer.stride, layer.pad
        p.ldx, p.ldy, p.ldr = ldx, ldy, ldr
        p.flags = flags
        ho, wo = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(self.lib.cnl_conv2d_out_hw(ctypes.byref(p), ctypes.byref(ho), ctypes.byref(wo)), what)
        flops = 2 * self.N * ho.value * wo.value * layer.cout * layer.kh * layer.kw * layer.cin   # direct-conv (algorithmic) flops
        fn = self.lib.cnl_conv2d_nhwc_f32
        if layer.u is not None and not (flags & (CNL_UPSAMPLE_OUT_ADD | CNL_SIGMOID)) and (4 * x_off) % 16 == 0:
            p.w = layer.u.data_ptr()
            fn = self.lib.cnl_conv3x3_winograd_f32
            what += " [winograd]"
        self.launches.append(_Launch(fn, p, what, flops, keep=(x, y, residual, layer)))
        return p, ho.value, wo.value

    def _build(self, Wt):
        N, H, W = self.N, self.H, self.W
        self._wt_stem = Wt.stem
        self._wt_stem_packed = Wt.stem_packed
        if H % 32 or W % 32:
            raise ValueError(f"input H, W must be divisible by 32 (got {H}x{W}); docs/implementation.md:52 of the reference")
        # ---- backbone ----
        h2, w2 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        s1 = self._buf(N, h2, w2, 64)
        self.stem_out = s1
        h4, w4 = (h2 + 2 - 3) // 2 + 1, (w2 + 2 - 3) // 2 + 1
        cur = self._buf(N, h4, w4, 64)
        self.launches.append(_Launch("stem", None, "stem7x7+bn+relu", 2 * N * h2 * w2 * 64 * 147))
        self.launches.append(_Launch("maxpool", (s1, cur, N, h2, w2, 64), "maxpool3x3s2"))
        ch, cw, cc = h4, w4, 64
        feats = {}
        for bi, (c1, c2, down, li) in enumerate(Wt.blocks):
            oh, ow = (ch + 2 - 3) // c1.stride + 1, (cw + 2 - 3) // c1.stride + 1
            t = self._buf(N, oh, ow, c1.cout)
            self._conv(c1, cur, ch, cw, cc, t, c1.cout, CNL_RELU, what=f"layer{li + 1}.{bi}.conv1")
            if down is not None:
                idn = self._buf(N, oh, ow, down.cout)
                self._conv(down, cur, ch, cw, cc, idn, down.cout, 0, what=f"layer{li + 1}.{bi}.downsample")
            else:
                idn = cur
            out = self._buf(N, oh, ow, c2.cout)
            self._conv(c2, t, oh, ow, c1.cout, out, c2.cout, CNL_RELU, residual=idn, ldr=c2.cout, what=f"layer{li + 1}.{bi}.conv2")
            cur, ch, cw, cc = out, oh, ow, c2.cout
            feats[li] = (cur, ch, cw, cc)
        self.features = [(s1, h2, w2, 64)] + [feats[i] for i in range(4)]       # strides 2,4,8,16,32

        # ---- neck ----
        if Wt.neck_kind == "SimpleNeck":
            x, xh, xw, xc = self.features[-1]
            up = 0
            for i, layer in enumerate(Wt.neck_layers):
                y = self._buf(N, xh * (2 if up else 1), xw * (2 if up else 1), layer.cout)
                _, oh, ow = self._conv(layer, x, xh, xw, xc, y, layer.cout, CNL_RELU | up, what=f"neck.layers.{i}")
                x, xh, xw, xc, up = y, oh, ow, layer.cout, CNL_UPSAMPLE_IN
            neck, nh, nw, nc, neck_up = x, xh, xw, xc, CNL_UPSAMPLE_IN      # final upsample folded into the heads
            oh_, ow_ = 2 * nh, 2 * nw
        else:
            top, th, tw, tc = self.features[-1]
            top_layer = Wt.top
            for i, (skip_p, top_p, out_conv) in enumerate(Wt.fuse):
                skip, sh_, sw_, sc_ = self.features[-2 - i]
                if skip_p is not None:
                    sp = self._buf(N, sh_, sw_, skip_p.cout)
                    self._conv(skip_p, skip, sh_, sw_, sc_, sp, skip_p.cout, 0, what=f"neck.fuse.{i}.project.0")
                    skip, sc_ = sp, skip_p.cout
                if i == 0:
                    # level 0: `top = top_conv(c5)`; Fuse.project[1] of level 0 is Identity when channels match
                    lay = top_layer
                    if not _is_identity(top_p):
                        tt = self._buf(N, th, tw, top_layer.cout)
                        self._conv(top_layer, top, th, tw, tc, tt, top_layer.cout, 0, what="neck.top_conv")
                        top, tc, lay = tt, top_layer.cout, top_p
                else:
                    lay = top_p
                if lay.cout != sc_ or sh_ != 2 * th or sw_ != 2 * tw:
                    raise ValueError(f"FPN level {i}: skip {sc_}ch@{sh_}x{sw_} does not match top {lay.cout}ch@{th}x{tw} x2")
                fused = self._buf(N, sh_, sw_, lay.cout)
                self._conv(lay, top, th, tw, tc, fused, lay.cout, CNL_UPSAMPLE_OUT_ADD, residual=skip, ldr=sc_,
                           what=f"neck.fuse.{i}.project+up+sum")
                y = self._buf(N, sh_, sw_, out_conv.cout)
                self._conv(out_conv, fused, sh_, sw_, sc_, y, out_conv.cout, CNL_RELU, what=f"neck.fuse.{i}.output_conv")
                top, th, tw, tc = y, sh_, sw_, out_conv.cout
            neck, nh, nw, nc, neck_up = top, th, tw, tc, 0
            oh_, ow_ = nh, nw
        self.neck_out = (neck, nh, nw, nc, neck_up)
        self.out_hw = (oh_, ow_)

        # ---- heads ----
        first = {}
        if Wt.fused_first is not None:
            tot = Wt.fused_first.cout
            fb = self._buf(N, oh_, ow_, tot)
            self._conv(Wt.fused_first, neck, nh, nw, nc, fb, tot, CNL_RELU | neck_up, what="heads.*.block_1 (fused)")
            off = 0
            for name in Wt.head_names:
                wdt = Wt.head_blocks[name][0].cout
                first[name] = (fb, tot, off, wdt)
                off += wdt
        for name in Wt.head_names:
            blocks = Wt.head_blocks[name]
            if name in first:
                x, ldx, xoff, xc = first[name]
                rest, xh, xw, up = blocks[1:], oh_, ow_, 0
            else:
                x, ldx, xoff, xc = neck, nc, 0, nc
                rest, xh, xw, up = blocks, nh, nw, neck_up
            for bi, layer in enumerate(rest):
                y = self._buf(N, oh_, ow_, layer.cout)
                self._conv(layer, x, xh, xw, ldx, y, layer.cout, CNL_RELU | up, what=f"heads.{name}.block", x_off=xoff)
                x, ldx, xoff, xc, xh, xw, up = y, layer.cout, 0, layer.cout, oh_, ow_, 0
            outl = Wt.head_out[name]
            flags = up | (CNL_SIGMOID if (name == "heatmap" and self.sigmoid) else 0)
            # output buffer is allocated fresh per call (ownership passes to the caller); patched in run()
            p, _, _ = self._conv(outl, x, xh, xw, ldx, x, outl.cout, flags, what=f"heads.{name}.out_conv", x_off=xoff)
            self.out_params[name] = (p, outl.cout)

    def run(self, x):
        """x: [N,3,H,W] fp32 on self.device, any strides.  Returns OrderedDict name -> logical-NCHW view of a fresh
        NHWC tensor (channels_last, zero-copy; precedent models/meta.py:97-98)."""
        lib = self.lib
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        outs = OrderedDict()
        oh, ow = self.out_hw
        for name, (p, c) in self.out_params.items():
            t = torch.empty((self.N, oh, ow, c), device=self.device, dtype=torch.float32)
            p.y = t.data_ptr()
            outs[name] = t
        for L in self.launches:
            if L.fn == "stem":
                sn, sc, sh, sw = x.stride()
                rc = lib.cnl_stem_conv7x7_f32(x.data_ptr(), sn, sc, sh, sw, self._wt_stem_packed.data_ptr(), self._wt_stem.b.data_ptr(),
                                              self.stem_out.data_ptr(), self.N, self.H, self.W, stream)
            elif L.fn == "maxpool":
                src, dst, n, h, w, c = L.args
                rc = lib.cnl_maxpool3x3s2_nhwc_f32(src.data_ptr(), dst.data_ptr(), n, h, w, c, stream)
            else:
                rc = L.fn(ctypes.byref(L.args), stream)
            if rc != 0:
                _lib.check(rc, L.what)
        return OrderedDict((k, v.permute(0, 3, 1, 2)) for k, v in outs.items())

    def total_flops(self):
        return sum(L.flops for L in self.launches)


def _is_identity(layer):
    return layer.kh == 1 and layer.cin == layer.cout and bool(torch.equal(
        layer.w.view(layer.cout, layer.cin), torch.eye(layer.cout, device=layer.w.device)))


class Engine:
    """Per-model cache of packed weights and per-shape plans."""

    def __init__(self, model):
        self.model = model
        self.weights = None
        self.plans = {}

    def invalidate(self):
        self.weights = None
        self.plans.clear()

    def forward(self, x, sigmoid):
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise RuntimeError("CenterNet (MI355X) runs on HIP devices only: move the input to 'cuda' — there is no CPU "
                               "fallback (the CPU oracle lives under oracle/ and is test infrastructure)")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected input of shape [N,3,H,W], got {tuple(x.shape)}")
        if x.dtype != torch.float32:
            raise ValueError(f"expected float32 input, got {x.dtype}")
        dev = x.device
        if self.weights is None or self.weights_device != dev:
            self.weights = PackedWeights(self.model, dev)
            self.weights_device = dev
            self.plans.clear()
        N, _, H, W = x.shape
        chunk = self.max_batch(H, W)
        if N > chunk:
            # the kernels address each tensor through 32-bit buffer offsets (< 4 GiB per tensor): run contiguous sub-batches
            # (images are independent; results are byte-identical to one big batch) and concatenate
            parts = [self._run(x[i:i + chunk], sigmoid) for i in range(0, N, chunk)]
            return OrderedDict((k, torch.cat([p[k] for p in parts], dim=0)) for k in parts[0])
        return self._run(x, sigmoid)

    def max_batch(self, H, W):
        """Largest batch whose biggest activation tensor stays below the 4 GiB buffer-addressing limit of the kernels."""
        widest = max(64, sum(b[0].cout for b in self.weights.head_blocks.values() if b) if self.weights.fused_first is not None else 0,
                     max((l.cout for b in self.weights.head_blocks.values() for l in b), default=64))
        per_image = max((H // 2) * (W // 2) * 64, (H // 4) * (W // 4) * widest) * 4
        return max(1, (0xF0000000 - (1 << 24)) // per_image)

    def _run(self, x, sigmoid):
        dev = x.device
        N, _, H, W = x.shape
        key = (N, H, W, bool(sigmoid))
        plan = self.plans.get(key)
        if plan is None:
            with torch.cuda.device(dev):
                plan = Plan(self.weights, N, H, W, dev, bool(sigmoid))
            self.plans[key] = plan
        with torch.cuda.device(dev):
            return plan.run(x)
