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
import bisect
import ctypes
import threading
from collections import OrderedDict
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import (CNL_ALGO_AUTO, CNL_ALGO_F2, CNL_ALGO_F32, CNL_ALGO_F43, CNL_ALGO_LATENCY, CNL_RELU, CNL_RELU6, CNL_SIGMOID, CNL_UPSAMPLE_IN, CNL_UPSAMPLE_OUT_ADD, CNL_W_SPLIT,
                   CNL_WINO_F16X2, ConvParams, DeconvParams)

BN_EPS_DEFAULT = 1e-5


@dataclass(frozen=True)
class KernelOptions:
    """What the launch plan may use — an explicit, per-model choice (CenterNet.set_kernel_options), part of the plan key; the library
    reads nothing from the environment.
      algo             "auto": fp32-grade kernels on the fp16-split matrix cores where they pay (every kernel's error vs float64 at or
                       below the fp32 matrix core's; "f2" is a synonym);
                       "f32":  fp32 matrix cores only (no split operands, no hints).
      winograd         False: every conv on the direct implicit-GEMM kernels (A/B and parity checks)
      up2              the fused first head blocks behind a nearest upsample as four sub-pixel phase convs
      absmax_handover  per-image max |y| handed from producer to consumer launches (else each fp16-split Winograd launch makes
                       its own pass and the direct convs stay on the fp32 matrix cores)
      stem_fused_pool  the 3x3/2 max-pool inside the stem kernel
      presplit_weights the direct convs' weights carry their fp16 split (CNL_W_SPLIT): the fp16-split direct kernel reads the pieces
                       instead of splitting every chunk's weights again — same bits out, -20..-35 % on the stride-2 3x3 convs
      fuse_small_out   an out_conv with at most 4 output channels (box sizes; the heatmap of 1-2 class models) is folded into the epilogue of the
                       3x3 block before it (cnl_conv_params.fuse_w: per-32-channel partial sums + a fixed-order reduce) instead of re-reading
                       the block's 256-channel output (C1: 537 MB, 110 us)
      reuse_buffers    activation buffers share one arena by liveness; False keeps every intermediate (tests read them)
      latency          latency class for one-image batches (default off): every 3x3 / stride-1 layer the row-Winograd kernels can run
                       takes csrc/winograd10.hip's 4-row x 64-pixel x 32-cout work items (cnl_conv_params.algo = CNL_ALGO_LATENCY) — four times the
                       work items of winograd9's, two workgroups per CU.  (The default already does this for winograd9's own layers when a launch has
                       at most 128 work items — same bits; the option adds the layers whose default is another kernel.)
      up_rows          a 3x3 conv the row-Winograd kernel runs behind a folded nearest-2x upsample (the first head blocks and the stages of the simple neck)
                       gets the layer's ROW-PAIR weights (cnl_conv_params.w_up, ABI v12): the upsampled rows 2j and 2j+1 are one source row, so an output row
                       meets two distinct input rows instead of three — 96 instead of 144 matrix instructions per 16-channel chunk; within fp32 rounding of
                       the general form (False), batch-invariant
      f43              opt-in arithmetic class (default off): 3x3 / stride-1 layers with Cin >= 128 on maps that 4-row x 128-pixel work items tile
                       well (the 256 -> 256 head blocks) run as 1-D Winograd F(4,3) along x (csrc/winograd13.hip, cnl_conv_params.algo = CNL_ALGO_F43): 25 % fewer
                       matrix instructions, 2.4-4.6 x the fp32 matrix core's rounding error (inside the path's 1e-4 by two orders of magnitude, above "auto"'s promise)
      check_range      debug guard for the split arithmetic's input range (default off; VERDICT r5 #4c): one power-of-two scale per image means a value 2^-n below the
                       image's maximum keeps min(22, 38 - n) significant bits — regions whose magnitudes lie 10^6 below the image maximum are at 2-6e-5 of THEIR magnitude,
                       10^7 breaks the path's 1e-4, and nothing in the kernels notices.  With the option every fp16-split launch's input is inspected before it runs
                       (per image: the maximum over all channels of each 8 x 8-pixel tile; the smallest non-zero tile maximum against the image maximum) and a
                       SplitRangeError names the launch when the ratio exceeds 10^5 — the caller then uses algo="f32" for that model.  A full extra pass per launch
                       and a host synchronisation: a diagnostic, not a production setting
      split_small      for small batches (default off): launches whose output is too small to fill the chip (one image: the
                       16x16 .. 64x64 maps) run as direct convs with the reduction split over several workgroups per output tile and a
                       fixed-order reduce (cnl_conv_params.splitk).  The choice then depends on the batch size, so results are no longer
                       bit-identical between a shard and the full batch (they stay within the fp32-grade error bars)."""
    algo: str = "auto"
    winograd: bool = True
    up2: bool = True
    absmax_handover: bool = True
    stem_fused_pool: bool = True
    presplit_weights: bool = True
    fuse_small_out: bool = True
    reuse_buffers: bool = True
    split_small: bool = False
    latency: bool = False
    up_rows: bool = True
    f43: bool = False
    check_range: bool = False

    @property
    def algo_id(self):
        try:
            return {"auto": CNL_ALGO_AUTO, "f2": CNL_ALGO_F2, "f32": CNL_ALGO_F32}[self.algo]
        except KeyError:
            raise ValueError(f"KernelOptions.algo must be 'auto', 'f2' or 'f32', got {self.algo!r}") from None


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


class _Layer:
    """One packed conv layer: folded OHWI weight + bias on the device (+ the Winograd-transformed weight for
    3x3 / stride-1 layers, produced once by cnl_winograd_transform_weights_f32)."""

    def __init__(self, w_ohwi, bias, stride=1):
        self.w, self.b = w_ohwi, bias
        self.cout, self.kh, self.kw, self.cin = w_ohwi.shape
        self.stride = stride
        self.pad = (self.kh - 1) // 2
        self.wmax = w_ohwi.abs().max().reshape(1).contiguous()      # cnl_conv_params.w_absmax (fp16-split direct kernel)
        self.u = None
        if self.kh == 3 and self.kw == 3 and stride == 1 and self.cin % 8 == 0 and w_ohwi.is_cuda:
            self._transform()

    def _transform(self):
        lib = _lib.load()
        w = self.w
        n = lib.cnl_winograd_weight_floats(self.cin, self.cout)
        with torch.cuda.device(w.device):
            self.u = torch.empty((n,), device=w.device, dtype=torch.float32)
            stream = ctypes.c_void_p(torch.cuda.current_stream(w.device).cuda_stream)
            _lib.check(lib.cnl_winograd_transform_weights_f32(w.data_ptr(), self.u.data_ptr(), self.cin, self.cout, stream),
                       "cnl_winograd_transform_weights_f32")


def _layer_split_w(self):
    """The layer's weights for the direct conv kernels with their fp16 split appended (cnl_conv_split_weights_f32; flags |= CNL_W_SPLIT),
    built on first use; None where the split form does not apply (non-square / other kernel sizes, Cin % 32 != 0, CPU tensors)."""
    if getattr(self, "_wsplit", None) is None:
        lib = _lib.load()
        n = lib.cnl_conv_split_weight_floats(self.cin, self.cout, self.kh, self.kw) if self.w.is_cuda else 0
        if not n:
            self._wsplit = False
        else:
            with torch.cuda.device(self.w.device):
                buf = torch.empty((n,), device=self.w.device, dtype=torch.float32)
                stream = ctypes.c_void_p(torch.cuda.current_stream(self.w.device).cuda_stream)
                _lib.check(lib.cnl_conv_split_weights_f32(self.w.data_ptr(), buf.data_ptr(), self.cin, self.cout, self.kh, self.kw, stream),
                           "cnl_conv_split_weights_f32")
                torch.cuda.current_stream(self.w.device).synchronize()      # (as up_rows: one buffer for the plans of every stream)
            self._wsplit = buf
    return self._wsplit if self._wsplit is not False else None


def _layer_wants_up2(self):
    """cnl_conv3x3_up2_nhwc_f32 instead of Winograd for this layer when its input is nearest-2x upsampled: measured on 64 -> 512
    @64^2 -> 128^2 (861 vs 1062 us); the small neck layers (256 -> 128 @16^2: 130 vs 50 us) stay on Winograd.  Shape only."""
    return (self.kh == 3 and self.kw == 3 and self.stride == 1 and self.cin % 32 == 0 and self.cin <= 64 and self.cout >= 256
            and self.w.is_cuda)


def _layer_up2(self):
    if getattr(self, "_up2", None) is None:
        lib = _lib.load()
        with torch.cuda.device(self.w.device):
            self._up2 = torch.empty((lib.cnl_up2_weight_floats(self.cin, self.cout),), device=self.w.device, dtype=torch.float32)
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.w.device).cuda_stream)
            _lib.check(lib.cnl_up2_pack_weights_f32(self.w.data_ptr(), self._up2.data_ptr(), self.cin, self.cout, stream), "cnl_up2_pack_weights_f32")
            self.up2_wmax = self._up2.abs().max().reshape(1).contiguous()
    return self._up2


def _layer_up_rows(self):
    """The layer's row-pair weight sets for cnl_conv_params.w_up (cnl_winograd_transform_weights_up_f32), built on first use; None where the form does not
    apply (Cin % 32 != 0, CPU tensors)."""
    if getattr(self, "_up_rows", None) is None:
        lib = _lib.load()
        n = lib.cnl_winograd_up_weight_floats(self.cin, self.cout) if (self.w.is_cuda and self.kh == 3 and self.kw == 3 and self.stride == 1) else 0
        if not n:
            self._up_rows = False
        else:
            with torch.cuda.device(self.w.device):
                buf = torch.empty((n,), device=self.w.device, dtype=torch.float32)
                stream = ctypes.c_void_p(torch.cuda.current_stream(self.w.device).cuda_stream)
                _lib.check(lib.cnl_winograd_transform_weights_up_f32(self.w.data_ptr(), buf.data_ptr(), self.cin, self.cout, stream),
                           "cnl_winograd_transform_weights_up_f32")
                torch.cuda.current_stream(self.w.device).synchronize()      # built once, on whichever stream's plan asks first; plans of other streams read it too
            self._up_rows = buf
    return self._up_rows if self._up_rows is not False else None


_Layer.up_rows = _layer_up_rows
_Layer.split_w = _layer_split_w
_Layer.wants_up2 = _layer_wants_up2
_Layer.up2 = _layer_up2


class _SepLayer:
    """make_conv(conv_type="separable") (layers.py:56-69) packed: depthwise weight [3][3][C] (tap-major) + bias with BN folded,
    and the pointwise half as a 1x1 _Layer."""

    def __init__(self, mod, device):
        w = mod.dw.weight.detach().to(device=device, dtype=torch.float32)                    # [C,1,3,3]
        bn = mod.dw_bn
        scale = (bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)).to(device)
        self.dw_w = (w[:, 0] * scale.view(-1, 1, 1)).permute(1, 2, 0).contiguous()            # [3,3,C]
        self.dw_b = (bn.bias.detach().float().to(device) - bn.running_mean.detach().float().to(device) * scale).contiguous()
        self.pw = _Layer(*(t.to(device) for t in fold_conv_bn(mod.pw.weight, None, mod.pw_bn)))
        self.cin, self.cout = w.shape[0], self.pw.cout
        self.kh = self.kw = 3
        self.stride, self.pad = 1, 1


class _DeformLayer:
    """make_conv(conv_type="deformable") packed: ONE k x k conv producing offsets (+ mask logits) and the [Cout][k*k*Cin] GEMM
    weight (OHWI, BN folded) that multiplies the sampled columns (cnl_deform_sample_nhwc_f32 -> 1x1 cnl_conv2d_nhwc_f32)."""

    def __init__(self, mod, device):
        blk = mod.block
        ws, bs = [blk.offset_conv.weight], [blk.offset_conv.bias]
        self.has_mask = blk.mask_conv is not None
        if self.has_mask:
            ws.append(blk.mask_conv[0].weight)
            bs.append(blk.mask_conv[0].bias)
        w = torch.cat([t.detach().float() for t in ws], dim=0)
        b = torch.cat([t.detach().float() for t in bs], dim=0)
        self.om = _Layer(*(t.to(device) for t in fold_conv_bn(w, b, None)))          # offsets | mask logits, no activation
        gw, gb = fold_conv_bn(blk.deform_conv.weight, None, mod.bn)                   # OHWI [Cout, k, k, Cin]
        cout, k, _, cin = gw.shape
        self.gemm = _Layer(gw.reshape(cout, 1, 1, k * k * cin).contiguous().to(device), gb.to(device))
        self.cin, self.cout, self.k = cin, cout, k
        self.kh = self.kw = k
        self.stride, self.pad = 1, (k - 1) // 2


class _DeconvLayer:
    """make_upsample("conv_transpose") (layers.py:86-93) packed for cnl_deconv2x_nhwc_f32: BN folded, the K x K taps split into the
    four sub-pixel phase blocks [Cout][KHp][KWp][Cin] (layout: include/centernet_gfx950.h).  `gain` >= 0 is folded in as well
    (weighted fusion: relu(z) * g == relu(g * z))."""

    def __init__(self, mod, device, gain=1.0):
        lib = _lib.load()
        w = mod.deconv.weight.detach().to(device=device, dtype=torch.float32)                  # [Cin,Cout,K,K]
        bn = mod.bn
        scale = (bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)).to(device) * gain
        self.b = ((bn.bias.detach().float().to(device) * gain) - bn.running_mean.detach().float().to(device) * scale).contiguous()
        self.cin, self.cout, self.k = w.shape[0], w.shape[1], w.shape[2]
        p = (self.k + self.k % 2) // 2 - 1
        ws = w * scale.view(1, -1, 1, 1)
        blocks = []
        for dy in range(2):
            for dx in range(2):
                ty, py, tx, px = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
                _lib.check(lib.cnl_deconv_phase_geometry(self.k, dy, ctypes.byref(ty), ctypes.byref(py)), "cnl_deconv_phase_geometry")
                _lib.check(lib.cnl_deconv_phase_geometry(self.k, dx, ctypes.byref(tx), ctypes.byref(px)), "cnl_deconv_phase_geometry")
                kys = [dy + p + 2 * (py.value - j) for j in range(ty.value)]
                kxs = [dx + p + 2 * (px.value - j) for j in range(tx.value)]
                blk = ws[:, :, kys][:, :, :, kxs]                                             # [Cin,Cout,KHp,KWp]
                blocks.append(blk.permute(1, 2, 3, 0).contiguous().view(-1))                  # OHWI
        self.w = torch.cat(blocks).contiguous()
        assert self.w.numel() == lib.cnl_deconv_weight_floats(self.cin, self.cout, self.k)


def _scaled(layer, gain):
    """A copy of a (1x1, no-Winograd) _Layer with weight and bias multiplied by `gain` (weighted fusion folded into a projection)."""
    return _Layer((layer.w * gain).contiguous(), (layer.b * gain).contiguous(), stride=layer.stride)


def _identity_layer(c, device, gain=1.0):
    eye = (torch.eye(c, device=device, dtype=torch.float32) * gain).view(c, 1, 1, c).contiguous()
    return _Layer(eye, torch.zeros(c, device=device))


class _FuseNode:
    """A general Fuse node (params.FuseNode; layers.py:138-177) packed: per-input 1x1 projections, raw relu'd fusion weights + their
    denominator (layers.py:164-167), the resize of the last input and the output conv."""

    def __init__(self, mod, L, C, device):
        self.proj = [L(p) if isinstance(p, torch.nn.Conv2d) else None for p in mod.project]
        self.down = mod.resize_kind == "down"
        self.upsample_type = mod.upsample_type
        self.gains, self.den = None, 1.0
        if mod.weights is not None:
            wts = torch.relu(mod.weights.detach().float().cpu())
            self.gains = [float(v) for v in wts]
            self.den = float(wts.sum() + 1e-6)                 # fp32, like torch.sum(weights) + eps
        self.deconv = _DeconvLayer(mod.resize, device) if (not self.down and self.upsample_type == "conv_transpose") else None
        self.out_conv = C(mod.output_conv)


class PackedWeights:
    """Device-resident, BN-folded weights of a CenterNet model (rebuilt whenever parameters change)."""

    def __init__(self, model, device):
        # what the packed copy was made from: (tensor, version at packing time); Engine.forward rebuilds when a parameter or buffer was
        # replaced or written in place since (load_state_dict on a submodule, optimizer steps, manual edits)
        self._sources = [(t, t._version, t.data_ptr()) for t in list(model.parameters()) + list(model.buffers())]
        # where each of them (and each submodule) hangs: flat (dict, key, object) triples, so that the per-forward staleness check is a few
        # hundred identity comparisons instead of a recursive walk over the module tree
        self._slots = []
        for m in model.modules():
            self._slots += [(m._modules, k, c) for k, c in m._modules.items()]
            self._slots += [(m._parameters, k, t) for k, t in m._parameters.items()]
            self._slots += [(m._buffers, k, t) for k, t in m._buffers.items()]
        self._slot_sizes = [(d, len(d)) for d in {id(d): d for d, _, _ in self._slots}.values()]
        bb, neck, heads = model.backbone, model.neck, model.heads
        dev = lambda t: t.to(device)
        L = lambda conv, bn=None, stride=None: _Layer(*map(dev, fold_conv_bn(conv.weight, conv.bias, bn)),
                                                      stride=stride if stride is not None else conv.stride[0])
        self.stem = L(bb.conv1, bb.bn1)
        # the stem kernel DMAs its weights into LDS verbatim: pack OHWI -> [154][64] once (cnl_stem_pack_weights_f32)
        lib = _lib.load()
        with torch.cuda.device(device):
            self.stem_packed = torch.empty((lib.cnl_stem_packed_weight_floats(),), device=device, dtype=torch.float32)
            _lib.check(lib.cnl_stem_pack_weights_f32(self.stem.w.data_ptr(), self.stem_packed.data_ptr(),
                                                     ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                       "cnl_stem_pack_weights_f32")
        self.blocks = []
        for li in range(4):
            for blk in getattr(bb, f"layer{li + 1}"):
                d = L(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
                self.blocks.append((L(blk.conv1, blk.bn1), L(blk.conv2, blk.bn2), d, li))
        self.neck_kind = type(neck).__name__
        def C(m):
            kind = type(m).__name__
            if kind == "SeparableConvBn":
                return _SepLayer(m, device)
            if kind == "DeformableConvBn":
                return _DeformLayer(m, device)
            return L(m.conv_module, m.bn_module)
        self.upsample_type = getattr(neck, "upsample_type", "nearest")
        if self.neck_kind == "SimpleNeck":
            self.neck_layers = [C(m) for m in neck.layers]
            self.neck_ups = [_DeconvLayer(u, device) if self.upsample_type == "conv_transpose" else None for u in neck.upsamples]
        elif self.neck_kind == "IDANeck":
            self.fuse = []                                   # (the stem's stride-2 map is not read: len(fuse) < 4)
            self.stages = [[_FuseNode(f, L, C, device) for f in st] for st in neck.stages]
        elif self.neck_kind == "BiFPNNeck":
            self.fuse = []
            self.bifpn = [([_FuseNode(f, L, C, device) for f in lay.td],
                           [_FuseNode(f, L, C, device) for f in lay.bu] if lay.bu is not None else None) for lay in neck.bifpn]
        else:
            self.top = L(neck.top_conv)
            self.fuse = []
            for f in neck.fuse:
                ga = gb = 1.0
                if f.weights is not None:                    # Fuse.forward weighted branch (layers.py:164-167), folded on the host
                    wts = torch.relu(f.weights.detach().float())
                    den = float(wts.sum()) + 1e-6
                    ga, gb = float(wts[0]) / den, float(wts[1]) / den
                oc = f.output_conv
                c = {"SeparableConvBn": lambda: oc.pw.in_channels, "DeformableConvBn": lambda: oc.block.deform_conv.in_channels}.get(
                    type(oc).__name__, lambda: oc.conv_module.in_channels)()
                skip_p = L(f.project[0]) if isinstance(f.project[0], torch.nn.Conv2d) else None
                if f.weights is not None:
                    skip_p = _scaled(skip_p, ga) if skip_p is not None else _identity_layer(c, device, ga)
                top_p = L(f.project[1]) if isinstance(f.project[1], torch.nn.Conv2d) else None
                resize = None
                if self.upsample_type == "conv_transpose":
                    resize = _DeconvLayer(f.resize, device, gain=gb)        # relu(z) * gb == relu(gb * z), gb >= 0
                elif f.weights is not None:                                 # nearest / bilinear are linear: fold gb into the projection
                    top_p = _scaled(top_p, gb) if top_p is not None else _identity_layer(c, device, gb)
                self.fuse.append((skip_p, top_p, resize, C(f.output_conv), gb if f.weights is not None else None))
        # heads: first blocks fused along Cout when every head has depth >= 1
        self.head_names = list(heads.keys())
        self.head_blocks = OrderedDict()
        self.head_out = OrderedDict()
        for name, h in heads.items():
            self.head_blocks[name] = [L(m.conv_module, m.bn_module) for m in h.blocks()]
            self.head_out[name] = L(h.out_conv)
        self.fused_first = None
        self._fused_groups = {}
        if all(len(b) >= 1 for b in self.head_blocks.values()) and len(self.head_names) > 1:
            self.fused_first = self.fused_group(self.head_names)

    def fused_group(self, names):
        """The first blocks of heads `names` as ONE layer, concatenated along Cout (built once per group)."""
        key = tuple(names)
        if key not in self._fused_groups:
            w = torch.cat([self.head_blocks[n][0].w for n in key], dim=0).contiguous()
            b = torch.cat([self.head_blocks[n][0].b for n in key], dim=0).contiguous()
            self._fused_groups[key] = _Layer(w, b)
        return self._fused_groups[key]


    def stale(self, model=None):
        """A source tensor was written in place (load_state_dict on a submodule, manual edits: version counter), moved, or REPLACED
        (load_state_dict(assign=True), `module.weight = nn.Parameter(...)`, a swapped submodule: the module then holds another tensor
        object than the one captured when packing) since the weights were packed."""
        if model is not None:
            for d, k, obj in self._slots:
                if d.get(k) is not obj:
                    return True
            for d, n in self._slot_sizes:          # a parameter / buffer / submodule was added
                if len(d) != n:
                    return True
        for t, v0, p0 in self._sources:
            if t._version != v0 or t.data_ptr() != p0:
                return True
        return False


class BufferTooLarge(ValueError):
    """A plan buffer would cross the kernels' 32-bit buffer addressing (< 4 GiB per tensor): the engine retries with a smaller sub-batch."""


ADDRESS_LIMIT = 0xF0000000 - (1 << 24)

_VBASE = 1 << 60        # virtual addresses of plan buffers while the plan is being built (far above any device pointer)


class _VBuf:
    """An activation buffer of the plan: shape + a virtual address range while the launch list is assembled; bound to its slice of
    the plan's arena by Plan._bind (data_ptr() then returns the real address)."""
    __slots__ = ("shape", "nbytes", "vbase", "offset", "real", "first", "last")

    def __init__(self, shape, vbase):
        self.shape = tuple(shape)
        n = 1
        for d in shape:
            n *= d
        self.nbytes = (n * 4 + 255) // 256 * 256
        self.vbase, self.offset, self.real, self.first, self.last = vbase, None, None, None, None

    def data_ptr(self):
        return self.real if self.real is not None else self.vbase


class SplitRangeError(RuntimeError):
    """KernelOptions(check_range=True): an input of a fp16-split launch spans more than SPLIT_RANGE_LIMIT between an image's maximum and its weakest 8 x 8 tile."""


SPLIT_RANGE_LIMIT = 1e5


def split_range_ratio(x_nhwc):
    """Per image of an NHWC activation tensor: max |x| over the image / the smallest NON-ZERO maximum of an 8 x 8-pixel tile (over all channels) — how far below
    the image's one power-of-two scale the weakest populated region sits (1.0 for an all-zero image: exact zeros stay exact)."""
    a = x_nhwc.abs().amax(dim=3)                                       # [N, H, W]: per pixel, over channels
    tiles = torch.nn.functional.max_pool2d(a[:, None], 8, 8, ceil_mode=True).flatten(1)
    top = tiles.amax(dim=1)
    weakest = torch.where(tiles > 0, tiles, torch.full_like(tiles, float("inf"))).amin(dim=1)
    return torch.where(top > 0, top / weakest, torch.ones_like(top))


class _Launch:
    """One C-ABI call of the plan: `args` is a params struct (passed by reference) or a tuple of scalar / pointer arguments."""
    __slots__ = ("fn", "args", "what", "flops", "keep")

    def __init__(self, fn, args, what, flops=0, keep=()):
        self.fn, self.args, self.what, self.flops, self.keep = fn, args, what, flops, keep


class _ListSlot:
    """`y` of a launch whose arguments are a plain list (patched per call like ConvParams.y of the output convs)."""
    __slots__ = ("args", "index")

    def __init__(self, args, index):
        self.args, self.index = args, index

    @property
    def y(self):
        return self.args[self.index]

    @y.setter
    def y(self, v):
        self.args[self.index] = v


class Plan:
    """Activation arena + launch list for one (N, H, W, sigmoid, options, stream) signature.  A Plan is SINGLE-STREAM: its arena, its
    absmax slots and the output pointers patched per call are private state; Engine keeps one plan per stream, so forwards on
    different streams never share any of it."""

    def __init__(self, weights: PackedWeights, N, H, W, device, sigmoid, options: KernelOptions = KernelOptions()):
        self.lib = _lib.load()
        self.N, self.H, self.W, self.device, self.sigmoid, self.options = N, H, W, device, sigmoid, options
        self.algo = options.algo_id
        self.launches = []
        self.buffers = []
        self._vnext = _VBASE
        self.outputs = OrderedDict()      # name -> (NHWC buffer tensor [N,h,w,C])
        self.out_params = {}              # name -> ConvParams writing that output (y patched per call)
        self.stem_args = None
        self._build(weights)
        self._wire_absmax()
        self._bind()

    def _bind(self):
        """Liveness-based placement of the activation buffers in ONE arena: a buffer lives from the first to the last launch that
        mentions it; buffers whose lifetimes do not overlap share memory (launches are stream-ordered).  Then every virtual address
        baked into the launch arguments is translated.  reuse_buffers=False gives each buffer its own range (tests read
        intermediates after the run)."""
        bases = [b.vbase for b in self.buffers]

        def owner(addr):
            if not isinstance(addr, int) or addr < _VBASE:
                return None
            i = bisect.bisect_right(bases, addr) - 1
            b = self.buffers[i]
            assert b.vbase <= addr < b.vbase + b.nbytes + 256, "virtual address outside every plan buffer"
            return b

        def pointers(L):
            if isinstance(L.args, (ConvParams, DeconvParams)):
                return [(L.args, f) for f in (("x", "y", "residual", "splitk_scratch", "fuse_part") if isinstance(L.args, ConvParams) else ("x", "y", "residual"))]
            if isinstance(L.args, list):
                return [(L.args, i) for i in range(len(L.args))]
            return []

        def get(holder, key):
            return getattr(holder, key) if isinstance(key, str) else holder[key]

        fresh_y = {id(p_) for p_, _ in self.out_params.values()}          # their y is patched per call
        for i, L in enumerate(self.launches):
            used = [t for t in L.keep if isinstance(t, _VBuf)]
            for holder, key in pointers(L):
                if isinstance(key, str) and key == "y" and id(holder) in fresh_y:
                    continue
                b = owner(get(holder, key))
                if b is not None:
                    used.append(b)
            for b in used:
                b.first = i if b.first is None else b.first
                b.last = i
        self.bytes_without_reuse = sum(b.nbytes for b in self.buffers)
        live = [b for b in self.buffers if b.first is not None]
        if self.options.reuse_buffers:
            free, top, active = [], 0, []                  # free: [(offset, size)], active: buffers placed and not yet expired
            for b in sorted(live, key=lambda b_: b_.first):
                for a_ in [a_ for a_ in active if a_.last < b.first]:
                    active.remove(a_)
                    free.append((a_.offset, a_.nbytes))
                free.sort()
                merged = []
                for off, sz in free:                       # coalesce neighbours
                    if merged and merged[-1][0] + merged[-1][1] == off:
                        merged[-1] = (merged[-1][0], merged[-1][1] + sz)
                    else:
                        merged.append((off, sz))
                if merged and merged[-1][0] + merged[-1][1] == top:      # a free tail gives the arena back
                    top = merged.pop()[0]
                free = merged
                fit = [(sz, off) for off, sz in free if sz >= b.nbytes]
                if fit:
                    sz, off = min(fit)                     # best fit
                    free.remove((off, sz))
                    if sz > b.nbytes:
                        free.append((off + b.nbytes, sz - b.nbytes))
                    b.offset = off
                else:
                    b.offset = top
                    top += b.nbytes
                active.append(b)
            total = max((b.offset + b.nbytes for b in live), default=0)
        else:
            total = 0
            for b in live:
                b.offset = total
                total += b.nbytes
        self.arena_bytes = total
        self.arena = torch.empty((total // 4 + 64,), device=self.device, dtype=torch.float32)
        base = self.arena.data_ptr()
        for L in self.launches:
            for holder, key in pointers(L):
                v = get(holder, key)
                b = owner(v)
                if b is not None and b.offset is not None:
                    new = base + b.offset + (v - b.vbase)
                    if isinstance(key, str):
                        setattr(holder, key, new)
                    else:
                        holder[key] = new
        for b in live:
            b.real = base + b.offset

    def tensor(self, buf):
        """The [N, h, w, C] view of a plan buffer in the arena (valid until a later launch reuses the range: build the plan with
        KernelOptions(reuse_buffers=False) to read intermediates after a run)."""
        n = 1
        for d in buf.shape:
            n *= d
        return self.arena[buf.offset // 4: buf.offset // 4 + n].view(buf.shape)

    def _wire_absmax(self):
        """Hand the per-image maximum magnitude of a tensor from the launch that produces it to the fp16-split launches that
        consume it (cnl_conv_params.x_absmax / y_absmax): the Winograd ones then skip their own pass over the input, and the
        direct convs (which never make one) take the fp16-split kernel instead of the fp32 one.  Only where it is provably the
        whole story: the consumer's input buffer has exactly one writer in the plan, that writer runs an fp16-split kernel (the
        ones that report max |y|) and it writes every channel of the buffer.  A 3x3 direct conv without such a producer (the
        stride-2 conv after layer1) gets an explicit cnl_absmax_per_image_f32 pass: cheaper than what the split kernel saves."""
        lib, wino, direct, up2 = self.lib, self.lib.cnl_conv3x3_winograd_f32, self.lib.cnl_conv2d_nhwc_f32, self.lib.cnl_conv3x3_up2_nhwc_f32
        self.absmax = None
        if not self.options.absmax_handover or self.algo == CNL_ALGO_F32:   # every fp16-split Winograd launch makes its own pass,
            return                                                        # every direct conv stays on the fp32 matrix cores
        writers, unsafe = {}, set()
        fresh_y = {id(p) for p, _ in self.out_params.values()}      # output convs: y is a fresh tensor per call (keep[1] is a placeholder)
        for L in self.launches:
            if isinstance(L.args, ConvParams):
                if id(L.args) not in fresh_y:
                    writers.setdefault(id(L.keep[1]), []).append(L)
            else:                       # other launches: anything they hold may be written by them
                unsafe.update(id(t) for t in L.keep if isinstance(t, (torch.Tensor, _VBuf)))

        def would_split(L):             # a direct conv that takes the fp16-split kernel once it has both hints
            if (L.fn is not direct and L.fn is not up2) or not L.args.w_absmax:
                return False
            saved, L.args.x_absmax = L.args.x_absmax, L.args.w_absmax      # any non-null pointer: the choice looks at presence only
            k = (lib.cnl_conv2d_kernel if L.fn is direct else lib.cnl_conv3x3_up2_kernel)(ctypes.byref(L.args))
            L.args.x_absmax = saved
            return k == 5

        slot_of, pairs, passes, reports = {}, [], [], set()       # reports: launches that run an fp16-split kernel
        shared = []                                               # launches that take the slot of their group's explicit pass
        own_pass = {}                                             # input buffer -> fp16-split Winograd launches without a reporting producer
        stem_group = []                                           # fp16-split Winograd launches that read the stem's (pooled) output
        for L in self.launches:         # plan order: a direct conv only reports max |y| if it got its own hint
            if not isinstance(L.args, ConvParams):
                continue
            is_wino5 = L.fn is wino and lib.cnl_conv3x3_winograd_kernel(ctypes.byref(L.args)) == CNL_WINO_F16X2
            if L.fn is direct and (L.args.flags & CNL_UPSAMPLE_OUT_ADD):
                reports.add(id(L))          # the Fuse epilogue (project -> up -> + skip) reports max |y| from the fp32 kernel too: a producer only
                continue
            if not is_wino5 and not would_split(L):
                continue
            x = L.keep[0]
            ws = writers.get(id(x), [])
            P = ws[0] if len(ws) == 1 else None
            if (P is not None and P is not L and id(x) not in unsafe and id(P) in reports and P.args.y == x.data_ptr()
                    and P.args.Cout == P.args.ldy):
                pairs.append((P, L, slot_of.setdefault(id(P), len(slot_of))))
                reports.add(id(L))
            elif is_wino5 and x is self.backbone_in:
                # the stem's output: the fp16-split stem kernel reports max |y| per image (max-pooling keeps the maximum of its
                # non-negative input, so the figure holds for the pooled map whichever launch pooled it)
                reports.add(id(L))
                stem_group.append(L)
            elif is_wino5:
                # no reporting producer: an explicit cnl_absmax_per_image_f32 pass into a slot of THIS plan's absmax tensor (the library's own
                # pass parks the maxima in scratch inside the shared weight buffer: two streams running one model would race on it — ADVICE r2).
                # Launches reading the same single-writer tensor share one pass; a tensor with several writers gets a pass per reader
                reports.add(id(L))
                key = (id(x), L.args.x, L.args.Cin, L.args.ldx) if (id(x) not in unsafe and len(ws) <= 1) else ("solo", id(L))
                own_pass.setdefault(key, []).append(L)
            elif L.args.KH == 3:
                passes.append((L, slot_of.setdefault(id(L), len(slot_of))))
                reports.add(id(L))
        # a tensor read by several such launches (the per-head first blocks behind an fp32-kernel neck layer): ONE explicit pass, shared
        for key, group in own_pass.items():
            passes.append((group[0], slot_of.setdefault(("shared", key), len(slot_of))))
            shared.extend((L, key) for L in group[1:])
        if stem_group:
            slot_of[("stem",)] = len(slot_of)
        # one float per (tensor, image): an image's scale must not depend on its batch neighbours
        ams = _lib.absmax_stride()             # floats between the per-image slots: one cache line per image
        row = 4 * self.N * ams                 # bytes per tensor
        self.absmax = torch.zeros((max(len(slot_of), 1), self.N, ams), device=self.device, dtype=torch.float32) if slot_of else None
        for P, L, i in pairs:
            P.args.y_absmax = self.absmax.data_ptr() + i * row
            L.args.x_absmax = self.absmax.data_ptr() + i * row
        for L, key in shared:
            L.args.x_absmax = self.absmax.data_ptr() + slot_of[("shared", key)] * row
        if stem_group:
            self.stem_absmax = self.absmax.data_ptr() + slot_of[("stem",)] * row
            for L in stem_group:
                L.args.x_absmax = self.stem_absmax
        for L, i in passes:
            a = L.args
            L.args.x_absmax = self.absmax.data_ptr() + i * row
            self.launches.insert(self.launches.index(L), _Launch(
                lib.cnl_absmax_per_image_f32, [a.x, self.N, a.H_in * a.W_in, a.Cin, a.ldx, L.args.x_absmax], L.what + ".absmax",
                0, keep=(self.absmax, L.keep[0])))

    # -- helpers --
    def _buf(self, n, h, w, c):
        if n * h * w * c * 4 > ADDRESS_LIMIT:
            raise BufferTooLarge(f"activation [{n},{h},{w},{c}] fp32 exceeds the kernels' 4 GiB buffer addressing")
        t = _VBuf((n, h, w, c), self._vnext)
        self._vnext += t.nbytes + 4096
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
        p.algo = self.algo
        p.w_absmax = layer.wmax.data_ptr()
        ho, wo = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(self.lib.cnl_conv2d_out_hw(ctypes.byref(p), ctypes.byref(ho), ctypes.byref(wo)), what)
        flops = 2 * self.N * ho.value * wo.value * layer.cout * layer.kh * layer.kw * layer.cin   # direct-conv (algorithmic) flops
        fn = self.lib.cnl_conv2d_nhwc_f32
        split = self._split_slices(layer, flags, ho.value, wo.value)
        if split > 1:
            # small grid: direct conv, reduction split over `split` workgroups per output tile + fixed-order reduce
            p.splitk = split
            scratch = self._buf(split, self.N * ho.value, wo.value, layer.cout)
            p.splitk_scratch = scratch.data_ptr()
            p.splitk_scratch_bytes = scratch.nbytes
            self.launches.append(_Launch(fn, p, what + f" [split x{split}]", flops, keep=(x, y, residual, layer, scratch)))
            return p, ho.value, wo.value
        if self.options.winograd and layer.u is not None and not (flags & (CNL_UPSAMPLE_OUT_ADD | CNL_SIGMOID)) and (4 * x_off) % 16 == 0:
            p.w = layer.u.data_ptr()
            fn = self.lib.cnl_conv3x3_winograd_f32
            what += " [winograd]"
            if self.options.latency and self.algo in (CNL_ALGO_AUTO, CNL_ALGO_F2):
                p.algo = CNL_ALGO_LATENCY
            elif self.options.f43 and self.algo in (CNL_ALGO_AUTO, CNL_ALGO_F2):
                p.algo = CNL_ALGO_F43
        # the row-Winograd kernels (winograd9.hip / winograd10.hip) fold the upsample into their patch gather and beat the sub-pixel phases
        # below (C1: 8.93 -> 8.82 ms per forward); whether a layer takes one is the dispatcher's decision — asked, not re-derived here
        rowwino = (fn is self.lib.cnl_conv3x3_winograd_f32 and self.algo != CNL_ALGO_F32
                   and self.lib.cnl_conv3x3_winograd_variant(ctypes.byref(p)) in (9, 10, 11))
        if rowwino and self.options.up_rows and (flags & CNL_UPSAMPLE_IN) and residual is None:
            wu = layer.up_rows()                        # the row-pair form of the row-Winograd kernel: two kernel rows per output row behind the upsample
            if wu is not None:
                p.w_up = wu.data_ptr()
        if (self.options.up2 and (flags & CNL_UPSAMPLE_IN) and not (flags & ~(CNL_RELU | CNL_UPSAMPLE_IN)) and residual is None
                and layer.wants_up2() and not rowwino):
            # short channel loop, many couts, conv on the nearest-2x upsampled input (the fused first head blocks behind the simple
            # neck): four 2x2 sub-pixel phase convs on the low-resolution input (fp16-split direct kernel) beat Winograd there
            p.w = layer.up2().data_ptr()
            p.w_absmax = layer.up2_wmax.data_ptr()
            p.algo = self.algo
            fn = self.lib.cnl_conv3x3_up2_nhwc_f32
            what = what.replace(" [winograd]", "") + " [sub-pixel phases]"
        if fn is self.lib.cnl_conv2d_nhwc_f32 and self.options.presplit_weights and self.algo != CNL_ALGO_F32:
            ws = layer.split_w()
            if ws is not None:
                p.w = ws.data_ptr()                     # [fp32 weights | their fp16 split | scale]: the fp32 kernels read the first part
                p.flags |= CNL_W_SPLIT
        self.launches.append(_Launch(fn, p, what, flops, keep=(x, y, residual, layer)))
        return p, ho.value, wo.value

    def _split_slices(self, layer, flags, ho, wo):
        """KernelOptions.split_small: slices of the reduction for a launch with at most 32 output tiles of 64 x 128 (an eighth of the CUs).  Only
        where the channel loop is long enough to pay for the second launch (Cin >= 256; their inputs' maxima are always handed over)."""
        if not self.options.split_small or self.algo == CNL_ALGO_F32 or not self.options.absmax_handover:
            return 0
        if flags & (CNL_UPSAMPLE_IN | CNL_UPSAMPLE_OUT_ADD) or layer.kh != layer.kw or layer.kh not in (1, 3) or layer.cin < 256 or layer.cin % 32:
            return 0
        tiles = -(-(self.N * ho * wo) // 64) * -(-layer.cout // 128)
        kt = layer.kh * layer.kw * layer.cin // 32
        if tiles > 64 or kt < 8:            # measured at N = 1 (profiles/r02_split_small.txt): a split launch costs ~25-35 us whatever its
                                            # size; the unsplit Winograd launch it replaces ~2.7 us per 16 channels -> Cin >= 256 only
            return 0
        s_ = min(kt // 3, max(1, 512 // tiles), 64)       # two 64 x 128 workgroups are co-resident per CU; >= 3 chunks per slice
        return s_ if s_ >= 2 else 0

    def _sep(self, layer, x, xh, xw, ldx, y, ldy, what):
        """Separable conv (layers.py:56-69): depthwise 3x3 + BN + ReLU6 -> pointwise 1x1 + BN + ReLU6."""
        t = self._buf(self.N, xh, xw, layer.cin)
        self.launches.append(_Launch(self.lib.cnl_depthwise3x3_nhwc_f32,
                                     [x.data_ptr(), layer.dw_w.data_ptr(), layer.dw_b.data_ptr(), t.data_ptr(), self.N, xh, xw,
                                      layer.cin, ldx, layer.cin, CNL_RELU6], what + ".dw", 0, keep=(x, t, layer)))
        self._conv(layer.pw, t, xh, xw, layer.cin, y, ldy, CNL_RELU6, what=what + ".pw")
        return xh, xw

    def _deform(self, layer, x, xh, xw, ldx, y, ldy, what):
        """Deformable conv (layers.py:9-38, 47-54): offsets / mask conv -> deformable sampling into columns -> 1x1 GEMM + BN + ReLU."""
        no = layer.om.cout
        om = self._buf(self.N, xh, xw, no)
        self._conv(layer.om, x, xh, xw, ldx, om, no, 0, what=what + ".offset+mask_conv")
        kk = layer.k * layer.k
        col = self._buf(self.N, xh, xw, kk * layer.cin)
        self.launches.append(_Launch(self.lib.cnl_deform_sample_nhwc_f32,
                                     [x.data_ptr(), om.data_ptr(), col.data_ptr(), self.N, xh, xw, layer.cin, ldx, no, layer.k,
                                      int(layer.has_mask)], what + ".sample", 0, keep=(x, om, col)))
        self._conv(layer.gemm, col, xh, xw, kk * layer.cin, y, ldy, CNL_RELU, what=what + ".deform_conv (GEMM)")
        return xh, xw

    def _block(self, layer, x, xh, xw, ldx, y, ldy, what, up=0):
        """make_conv(...) of any type; `up` = CNL_UPSAMPLE_IN folds a pending nearest x2 into a normal conv."""
        if isinstance(layer, _DeformLayer):
            assert not up
            return self._deform(layer, x, xh, xw, ldx, y, ldy, what)
        if isinstance(layer, _SepLayer):
            assert not up
            return self._sep(layer, x, xh, xw, ldx, y, ldy, what)
        _, oh, ow = self._conv(layer, x, xh, xw, ldx, y, ldy, CNL_RELU | up, what=what)
        return oh, ow

    def _upsample(self, x, xh, xw, c, ldx, mode, what, residual=None, ldr=0):
        """nn.Upsample x2 (0 nearest / 1 bilinear) materialised, + optional Fuse sum.  Returns the new buffer."""
        y = self._buf(self.N, 2 * xh, 2 * xw, c)
        self.launches.append(_Launch(self.lib.cnl_upsample2x_nhwc_f32,
                                     [x.data_ptr(), residual.data_ptr() if residual is not None else None, y.data_ptr(), self.N, xh, xw,
                                      c, ldx, ldr, c, mode], what, 0, keep=(x, y, residual)))
        return y

    def _deconv(self, layer, x, xh, xw, ldx, what, residual=None, ldr=0):
        """ConvTranspose2d x2 + BN + ReLU (+ Fuse sum after the activation).  Returns the new buffer."""
        y = self._buf(self.N, 2 * xh, 2 * xw, layer.cout)
        p = DeconvParams()
        p.x, p.w, p.bias, p.y = x.data_ptr(), layer.w.data_ptr(), layer.b.data_ptr(), y.data_ptr()
        p.residual = residual.data_ptr() if residual is not None else None
        p.N, p.H_in, p.W_in, p.Cin, p.Cout, p.K = self.N, xh, xw, layer.cin, layer.cout, layer.k
        p.ldx, p.ldy, p.ldr, p.flags = ldx, layer.cout, ldr, CNL_RELU
        flops = 2 * self.N * xh * xw * layer.cout * layer.k * layer.k * layer.cin
        self.launches.append(_Launch(self.lib.cnl_deconv2x_nhwc_f32, p, what, flops, keep=(x, y, residual, layer)))
        return y

    def _fuse_node(self, nd, ins, what):
        """One general Fuse node (layers.py:160-177): ins = [(buffer, h, w, c), ...], the last one at half ("up") or double ("down")
        resolution.  Lowering: 1x1 projections where the node has them; then
          * two inputs, nearest "up", unweighted, last input projected: project -> upsample -> + in0 in the epilogue of that ONE 1x1
            conv (CNL_UPSAMPLE_OUT_ADD), like the FPN levels;
          * anything else: cnl_fuse_sum_nhwc_f32 (gains, up to three inputs, the last resized on the fly; a conv_transpose resize is
            materialised first by cnl_deconv2x_nhwc_f32);
        then the output conv.  Returns (buffer, h, w, c)."""
        N = self.N
        oh, ow = ins[0][1], ins[0][2]
        lh, lw = ins[-1][1], ins[-1][2]
        if (not nd.down and (oh, ow) != (2 * lh, 2 * lw)) or (nd.down and (2 * oh, 2 * ow) != (lh, lw)) or any((h, w) != (oh, ow) for _, h, w, _ in ins[:-1]):
            raise ValueError(f"{what}: input sizes {[(h, w) for _, h, w, _ in ins]} do not fit a Fuse node with resize={'down' if nd.down else 'up'}")
        fused_up = (len(ins) == 2 and not nd.down and nd.upsample_type == "nearest" and nd.gains is None and nd.proj[-1] is not None)
        cur = []
        for j, (b, h, w, c) in enumerate(ins):
            pj = nd.proj[j]
            if pj is not None and not (fused_up and j == len(ins) - 1):
                t = self._buf(N, h, w, pj.cout)
                self._conv(pj, b, h, w, c, t, pj.cout, 0, what=f"{what}.project.{j}")
                b, c = t, pj.cout
            cur.append((b, h, w, c))
        oc = cur[0][3]
        fused = self._buf(N, oh, ow, oc)
        if fused_up:
            lb, _, _, lc = cur[-1]
            lay = nd.proj[-1]
            if lay.cout != oc:
                raise ValueError(f"{what}: channel mismatch {lay.cout} vs {oc}")
            self._conv(lay, lb, lh, lw, lc, fused, oc, CNL_UPSAMPLE_OUT_ADD, residual=cur[0][0], ldr=oc, what=f"{what}.project+up+sum")
        else:
            if any(c != oc for _, _, _, c in cur):
                raise ValueError(f"{what}: channel mismatch {[c for _, _, _, c in cur]}")
            lb = cur[-1][0]
            if nd.down:
                mode = 2
            elif nd.deconv is not None:
                lb, mode = self._deconv(nd.deconv, lb, lh, lw, oc, f"{what}.resize (conv_transpose)"), 3
            else:
                mode = 1 if nd.upsample_type == "bilinear" else 0
            g = nd.gains if nd.gains is not None else [1.0] * len(cur)
            in1 = cur[1][0] if len(cur) == 3 else None
            self.launches.append(_Launch(self.lib.cnl_fuse_sum_nhwc_f32,
                                         [cur[0][0].data_ptr(), in1.data_ptr() if in1 is not None else None, lb.data_ptr(), fused.data_ptr(),
                                          N, oh, ow, oc, oc, oc, oc, oc, g[0], g[1] if len(cur) == 3 else 0.0, g[-1], nd.den, mode],
                                         f"{what}.sum ({'max-pool down' if nd.down else nd.upsample_type})", 0, keep=(cur[0][0], in1, lb, fused)))
        y = self._buf(N, oh, ow, nd.out_conv.cout)
        self._block(nd.out_conv, fused, oh, ow, oc, y, nd.out_conv.cout, f"{what}.output_conv")
        return (y, oh, ow, nd.out_conv.cout)

    def _build(self, Wt):
        N, H, W = self.N, self.H, self.W
        self._wt_stem = Wt.stem
        self._wt_stem_packed = Wt.stem_packed
        if H % 32 or W % 32:
            raise ValueError(f"input H, W must be divisible by 32 (got {H}x{W}); docs/implementation.md:52 of the reference")
        # ---- backbone ----
        h2, w2 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        h4, w4 = (h2 + 2 - 3) // 2 + 1, (w2 + 2 - 3) // 2 + 1
        cur = self._buf(N, h4, w4, 64)
        # the stride-2 feature map is only materialised when a neck consumes it (an FPN with four Fuse levels); otherwise the stem
        # kernel pools its own tile and the 64-channel map at half resolution never reaches memory (cnl_stem_conv7x7_maxpool_f32)
        need_s1 = (Wt.neck_kind != "SimpleNeck" and len(Wt.fuse) >= 4) or not self.options.stem_fused_pool or self.algo == CNL_ALGO_F32
        s1 = self._buf(N, h2, w2, 64) if need_s1 else None
        self.stem_out = s1 if need_s1 else cur
        self.stem_fused_pool = not need_s1
        self.backbone_in = cur                 # the pooled stem output: its per-image maximum comes from the stem kernel itself (y_absmax)
        self.stem_absmax = None                # device pointer of that slot once _wire_absmax has placed it
        self.launches.append(_Launch("stem", None, "stem7x7+bn+relu" + ("" if need_s1 else "+maxpool3x3s2"), 2 * N * h2 * w2 * 64 * 147,
                                     keep=(self.stem_out,)))
        if need_s1:
            self.launches.append(_Launch("maxpool", (s1, cur, N, h2, w2, 64), "maxpool3x3s2", keep=(s1, cur)))
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
        bil = 1 if Wt.upsample_type == "bilinear" else 0
        if Wt.neck_kind == "SimpleNeck":
            x, xh, xw, xc = self.features[-1]
            up = 0                                    # CNL_UPSAMPLE_IN when a nearest x2 of `x` is still pending
            for i, (layer, dec) in enumerate(zip(Wt.neck_layers, Wt.neck_ups)):
                if up and isinstance(layer, (_SepLayer, _DeformLayer)):   # only the normal conv kernels fold the upsample gather
                    x, xh, xw, up = self._upsample(x, xh, xw, xc, xc, 0, f"neck.upsample.{i - 1} (nearest)"), 2 * xh, 2 * xw, 0
                y = self._buf(N, xh * (2 if up else 1), xw * (2 if up else 1), layer.cout)
                oh, ow = self._block(layer, x, xh, xw, xc, y, layer.cout, f"neck.layers.{i}", up)
                x, xh, xw, xc, up = y, oh, ow, layer.cout, 0
                if dec is not None:
                    x, xh, xw = self._deconv(dec, x, xh, xw, xc, f"neck.upsamples.{i} (conv_transpose)"), 2 * xh, 2 * xw
                elif bil:
                    x, xh, xw = self._upsample(x, xh, xw, xc, xc, 1, f"neck.upsample.{i} (bilinear)"), 2 * xh, 2 * xw
                else:
                    up = CNL_UPSAMPLE_IN                          # nearest: folded into the consumer
            neck, nh, nw, nc, neck_up = x, xh, xw, xc, up          # a pending final nearest upsample is folded into the heads
            oh_, ow_ = (2 * nh, 2 * nw) if up else (nh, nw)
        elif Wt.neck_kind == "IDANeck":
            levels = list(self.features[1:])
            for s_, stage in enumerate(Wt.stages):
                levels = [self._fuse_node(nd, [levels[i], levels[i + 1]], f"neck.stages.{s_}.{i}") for i, nd in enumerate(stage)]
            neck, nh, nw, nc = levels[0]
            neck_up, oh_, ow_ = 0, nh, nw
        elif Wt.neck_kind == "BiFPNNeck":
            ins = list(self.features[1:])
            for l_, (tds, bus) in enumerate(Wt.bifpn):
                n_ = len(ins)
                td = [None] * (n_ - 1) + [ins[-1]]
                for i in range(n_ - 2, -1, -1):
                    td[i] = self._fuse_node(tds[i], [ins[i], td[i + 1]], f"neck.bifpn.{l_}.td.{i}")
                if bus is None:
                    ins = td
                    continue
                outs = [td[0]]
                for i in range(1, n_ - 1):
                    outs.append(self._fuse_node(bus[i - 1], [ins[i], td[i], outs[i - 1]], f"neck.bifpn.{l_}.bu.{i - 1}"))
                outs.append(self._fuse_node(bus[n_ - 2], [ins[-1], outs[-1]], f"neck.bifpn.{l_}.bu.{n_ - 2}"))
                ins = outs
            neck, nh, nw, nc = ins[0]
            neck_up, oh_, ow_ = 0, nh, nw
        else:
            top, th, tw, tc = self.features[-1]
            top_pending = Wt.top                      # level 0 starts with `top = top_conv(c5)` (not yet materialised)
            for i, (skip_p, top_p, resize, out_conv, gb) in enumerate(Wt.fuse):
                skip, sh_, sw_, sc_ = self.features[-2 - i]
                if skip_p is not None:
                    sp = self._buf(N, sh_, sw_, skip_p.cout)
                    self._conv(skip_p, skip, sh_, sw_, sc_, sp, skip_p.cout, 0, what=f"neck.fuse.{i}.project.0")
                    skip, sc_ = sp, skip_p.cout
                fuse_c = top_p.cout if top_p is not None else (top_pending.cout if top_pending is not None else tc)
                if fuse_c != sc_ or sh_ != 2 * th or sw_ != 2 * tw:
                    raise ValueError(f"FPN level {i}: skip {sc_}ch@{sh_}x{sw_} does not match top {fuse_c}ch@{th}x{tw} x2")
                if resize is None and not bil:
                    # nearest: project -> upsample -> sum in the epilogue of ONE 1x1 conv (top_conv itself at level 0 when
                    # Fuse.project[1] is the identity; an exact identity 1x1 when there is nothing to project)
                    lay = top_p
                    if top_pending is not None:
                        if top_p is None:
                            lay = top_pending
                        else:
                            tt = self._buf(N, th, tw, top_pending.cout)
                            self._conv(top_pending, top, th, tw, tc, tt, top_pending.cout, 0, what="neck.top_conv")
                            top, tc = tt, top_pending.cout
                    if lay is None:
                        lay = _identity_layer(tc, self.device)
                    fused = self._buf(N, sh_, sw_, lay.cout)
                    self._conv(lay, top, th, tw, tc, fused, lay.cout, CNL_UPSAMPLE_OUT_ADD, residual=skip, ldr=sc_,
                               what=f"neck.fuse.{i}.project+up+sum")
                else:
                    for lay, nm in ((top_pending, "neck.top_conv"), (top_p, f"neck.fuse.{i}.project.1")):
                        if lay is not None:
                            tt = self._buf(N, th, tw, lay.cout)
                            self._conv(lay, top, th, tw, tc, tt, lay.cout, 0, what=nm)
                            top, tc = tt, lay.cout
                    if resize is not None:
                        fused = self._deconv(resize, top, th, tw, tc, f"neck.fuse.{i}.resize (conv_transpose)+sum", residual=skip, ldr=sc_)
                    else:
                        fused = self._upsample(top, th, tw, tc, tc, 1, f"neck.fuse.{i}.resize (bilinear)+sum", residual=skip, ldr=sc_)
                top_pending = None
                y = self._buf(N, sh_, sw_, out_conv.cout)
                self._block(out_conv, fused, sh_, sw_, sc_, y, out_conv.cout, f"neck.fuse.{i}.output_conv")
                top, th, tw, tc = y, sh_, sw_, out_conv.cout
            neck, nh, nw, nc, neck_up = top, th, tw, tc, 0
            oh_, ow_ = nh, nw
        self.neck_out = (neck, nh, nw, nc, neck_up)
        self.out_hw = (oh_, ow_)

        # ---- heads ----
        self.head_features = {}
        first = {}
        # the first blocks of the heads run as ONE launch writing one wide buffer (same input: meta.py:46).  Where 32 images of that buffer
        # (BASELINE's batch per GPU) would cross the kernels' 4 GiB addressing (C4: 32 x 152 x 272 x 768 floats = 4.06 GB against the engine's
        # limit of 4.01) the heads are fused in GROUPS that fit, in order (C4: heatmap + box_2d = 512 couts in one launch, reid on its own) —
        # decided per IMAGE, never by N: a shard and the full batch take the same launches.
        if Wt.fused_first is not None:
            groups, cur_g, cur_c = [], [], 0
            for name in Wt.head_names:
                c = Wt.head_blocks[name][0].cout
                if cur_g and 32 * oh_ * ow_ * (cur_c + c) * 4 > ADDRESS_LIMIT:
                    groups.append(cur_g)
                    cur_g, cur_c = [], 0
                cur_g.append(name)
                cur_c += c
            groups.append(cur_g)
            for grp in groups:
                if len(grp) < 2:
                    continue
                layer = Wt.fused_group(grp)
                tot = layer.cout
                fb = self._buf(N, oh_, ow_, tot)
                self._conv(layer, neck, nh, nw, nc, fb, tot, CNL_RELU | neck_up,
                           what="heads.*.block_1 (fused)" if len(grp) == len(Wt.head_names) else f"heads.{'+'.join(grp)}.block_1 (fused)")
                off = 0
                for name in grp:
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
            p_last = None
            for bi, layer in enumerate(rest):
                y = self._buf(N, oh_, ow_, layer.cout)
                p_last, _, _ = self._conv(layer, x, xh, xw, ldx, y, layer.cout, CNL_RELU | up, what=f"heads.{name}.block", x_off=xoff)
                x, ldx, xoff, xc, xh, xw, up = y, layer.cout, 0, layer.cout, oh_, ow_, 0
            self.head_features[name] = (x, ldx, xoff, xc, xh, xw, up)      # what out_conv reads (tests: feature-level parity gate)
            outl = Wt.head_out[name]
            flags = up | (CNL_SIGMOID if (name == "heatmap" and self.sigmoid) else 0)
            # an out_conv of at most 4 channels behind a row-Winograd block: folded into that block's epilogue (partial sums per 32 channels)
            # + a fixed-order reduce — decided from the layer shapes alone
            if self._fold_small_out(p_last, outl, name, flags, oh_, ow_):
                continue
            # output buffer is allocated fresh per call (ownership passes to the caller); patched in run()
            p, _, _ = self._conv(outl, x, xh, xw, ldx, x, outl.cout, flags, what=f"heads.{name}.out_conv", x_off=xoff)
            self.out_params[name] = (p, outl.cout)

    def _fold_small_out(self, p_last, outl, name, flags, oh, ow):
        if (p_last is None or not self.options.fuse_small_out or self.algo == CNL_ALGO_F32 or outl.kh != 1 or outl.kw != 1 or outl.cout > 4
                or outl.stride != 1 or (flags & ~CNL_SIGMOID) or p_last.residual):
            return False
        L = self.launches[-1]
        if L.args is not p_last or L.fn is not self.lib.cnl_conv3x3_winograd_f32 or p_last.Cout != p_last.ldy:
            return False
        # only winograd9's epilogue has the fold, and the dispatcher keeps a launch that carries fuse_w there.  The latency class (an explicit per-model option)
        # keeps its small work items instead and its out_conv stays a launch.  The DEFAULT plan folds whatever the grid size: the folded conv is
        # another summation order of the same products, so the choice must not look at N — a one-image shard and the full batch agree bit for bit
        # (tests/test_gpu_e2e.py::test_full_size_properties) — at the price of variant 9 instead of 11 on the last head block of a small batch (63 against 56 us at N = 1)
        if self.options.latency:
            return False
        nb = (p_last.Cout + 63) // 64 * 2
        if nb * self.N * oh * ow * 16 > ADDRESS_LIMIT:
            return False
        fw = getattr(outl, "_fuse_w", None)
        if fw is None:
            with torch.cuda.device(outl.w.device):
                fw = torch.empty(((p_last.Cout + 63) // 64 * 64, 4), device=outl.w.device, dtype=torch.float32)
                _lib.check(self.lib.cnl_fused_out_pack_weights_f32(outl.w.data_ptr(), fw.data_ptr(), outl.cin, outl.cout,
                                                                   ctypes.c_void_p(torch.cuda.current_stream(outl.w.device).cuda_stream)),
                           "cnl_fused_out_pack_weights_f32")
                torch.cuda.current_stream(outl.w.device).synchronize()      # (as up_rows / split_w: built once, read by the plans of every stream — ADVICE r5)
            outl._fuse_w = fw
        p_last.fuse_w = fw.data_ptr()
        if self.lib.cnl_conv3x3_winograd_variant(ctypes.byref(p_last)) != 9:        # the dispatcher keeps a folded launch on winograd9 — if it can run there
            p_last.fuse_w = None
            return False
        part = self._buf(nb, self.N * oh, ow, 4)
        p_last.fuse_part = part.data_ptr()
        L.keep = tuple(L.keep) + (part, fw)
        L.what += f" + {name}.out_conv partials"
        args = [part.data_ptr(), nb, self.N * oh * ow, outl.cout, outl.b.data_ptr(), 0, outl.cout, flags]
        self.launches.append(_Launch(self.lib.cnl_fused_out_reduce_f32, args, f"heads.{name}.out_conv (reduce of {nb} partials)", 0, keep=(part, outl)))
        self.out_params[name] = (_ListSlot(args, 5), outl.cout)
        return True

    def run(self, x, norm=None):
        """x: [N,3,H,W] fp32 on self.device, any strides — or, with norm = (mean255, inv_std255) ctypes float[3] arrays, uint8 frames
        [N,H,W,3] that the stem normalises on the fly.  Returns OrderedDict name -> logical-NCHW view of a fresh NHWC tensor
        (channels_last, zero-copy; precedent models/meta.py:97-98)."""
        self._norm = norm
        lib = self.lib
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        outs = OrderedDict()
        oh, ow = self.out_hw
        for name, (p, c) in self.out_params.items():
            t = torch.empty((self.N, oh, ow, c), device=self.device, dtype=torch.float32)
            p.y = t.data_ptr()
            outs[name] = t
        if self.absmax is not None:
            self.absmax.zero_()                                # the producers fold max |y| into these with atomic max
        for L in self.launches:
            if self.options.check_range:
                self._check_split_range(L)
            rc = self.launch(L, x, stream)
            if rc != 0:
                _lib.check(rc, L.what)
        return OrderedDict((k, v.permute(0, 3, 1, 2)) for k, v in outs.items())

    def input_view(self, p):
        """The [N, H_in, W_in, Cin] view (pixel stride ldx) of a conv launch's input inside the plan's arena."""
        off = (p.x - self.arena.data_ptr()) // 4
        assert 0 <= off < self.arena.numel(), "not an arena buffer"
        return self.arena.as_strided((p.N, p.H_in, p.W_in, p.Cin), (p.H_in * p.W_in * p.ldx, p.W_in * p.ldx, p.ldx, 1), self.arena.storage_offset() + off)

    def _check_split_range(self, L):
        """KernelOptions(check_range=True): inspect the input of a launch that scales it per image for the fp16 split (the producer launches of this forward have
        been enqueued on the same stream, so the buffer holds THIS forward's activations when the reductions below run)."""
        p = L.args
        if not isinstance(p, ConvParams) or not p.x_absmax:
            return
        lib = self.lib
        if L.fn is lib.cnl_conv3x3_winograd_f32:
            split = lib.cnl_conv3x3_winograd_kernel(ctypes.byref(p)) == CNL_WINO_F16X2
        elif L.fn is lib.cnl_conv2d_nhwc_f32:
            split = lib.cnl_conv2d_kernel(ctypes.byref(p)) == 5
        else:
            split = L.fn is lib.cnl_conv3x3_up2_nhwc_f32
        if not split:
            return
        ratio = split_range_ratio(self.input_view(p))
        worst = float(ratio.max())
        if worst > SPLIT_RANGE_LIMIT:
            n = int(ratio.argmax())
            raise SplitRangeError(f"{L.what}: image {n} of the batch holds a populated 8 x 8 tile {worst:.3g} x below the image's maximum (limit {SPLIT_RANGE_LIMIT:g}): "
                                  f"the fp16-split kernels scale an image by ONE power of two — use KernelOptions(algo='f32') for inputs like this (include/centernet_gfx950.h, x_absmax)")

    def launch(self, L, x, stream):
        """Issue ONE launch of the plan (x: the forward's input, read by the stem only).  Returns the C ABI's return code."""
        lib = self.lib
        if L.fn == "stem":
            if x.dtype == torch.uint8:                       # [N,H,W,3] uint8: byte strides of the logical (n, c, y, x) axes
                sn, sh, sw, sc = x.stride()
                return lib.cnl_stem_conv7x7_u8(x.data_ptr(), sn, sc, sh, sw, self._norm[0], self._norm[1], self._wt_stem_packed.data_ptr(),
                                               self._wt_stem.b.data_ptr(), self.stem_out.data_ptr(), self.stem_absmax, self.N, self.H, self.W,
                                               1 if self.stem_fused_pool else 0, stream)
            sn, sc, sh, sw = x.stride()
            if self.stem_fused_pool:
                return lib.cnl_stem_conv7x7_maxpool_f32(x.data_ptr(), sn, sc, sh, sw, self._wt_stem_packed.data_ptr(), self._wt_stem.b.data_ptr(),
                                                        self.stem_out.data_ptr(), self.stem_absmax, self.N, self.H, self.W, stream)
            return lib.cnl_stem_conv7x7_f32(x.data_ptr(), sn, sc, sh, sw, self._wt_stem_packed.data_ptr(), self._wt_stem.b.data_ptr(),
                                            self.stem_out.data_ptr(), self.stem_absmax, self.N, self.H, self.W, self.algo, stream)
        if L.fn == "maxpool":
            src, dst, n, h, w, c = L.args
            return lib.cnl_maxpool3x3s2_nhwc_f32(src.data_ptr(), dst.data_ptr(), n, h, w, c, stream)
        if isinstance(L.args, list):
            return L.fn(*L.args, stream)
        return L.fn(ctypes.byref(L.args), stream)

    def total_flops(self):
        return sum(L.flops for L in self.launches)


class Engine:
    """Per-model cache of packed weights and per-shape plans."""

    def __init__(self, model):
        self.model = model
        self.weights = None
        self._lock = threading.RLock()                 # host-side enqueue of one forward at a time: a plan's per-call state (patched
                                                       # output pointers, LRU order) is not re-entrant, and threads share the default stream
        self.plans = OrderedDict()                     # least recently used first
        self._sub_override = {}                        # (H, W) -> largest sub-batch found to fit after a BufferTooLarge retry
        self.max_plans = 8                             # each plan owns an activation arena (GBs at large batches): bound what is kept
        self.options = KernelOptions()

    def invalidate(self):
        self.weights = None
        self.plans.clear()
        self._sub_override.clear()

    def set_options(self, **kw):
        """Replace kernel options (KernelOptions fields); plans are keyed by them, so nothing else needs invalidating."""
        opts = KernelOptions(**{**self.options.__dict__, **kw})
        opts.algo_id                                   # validates
        self.options = opts
        return opts

    def forward_u8(self, images, mean255, inv_std255, sigmoid):
        """uint8 frames [N,H,W,3] -> outputs, normalised inside the stem kernel (no fp32 image in HBM)."""
        if not (isinstance(images, torch.Tensor) and images.is_cuda):
            raise RuntimeError("CenterNet (MI355X) runs on HIP devices only: move the frames to 'cuda' (no CPU fallback)")
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3:
            raise ValueError(f"expected uint8 frames [N,H,W,3], got {images.dtype} {tuple(images.shape)}")
        if self.options.algo == "f32":
            raise ValueError("the uint8 stem runs on the fp16-split kernel only: use algo 'auto' / 'f2', or preprocess_uint8() + forward()")
        return self._dispatch(images, sigmoid, images.shape[0], images.shape[1], images.shape[2], (mean255, inv_std255))

    def forward(self, x, sigmoid):
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise RuntimeError("CenterNet (MI355X) runs on HIP devices only: move the input to 'cuda' — there is no CPU "
                               "fallback (the CPU oracle lives under oracle/ and is test infrastructure)")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected input of shape [N,3,H,W], got {tuple(x.shape)}")
        if x.dtype != torch.float32:
            raise ValueError(f"expected float32 input, got {x.dtype}")
        N, _, H, W = x.shape
        return self._dispatch(x, sigmoid, N, H, W, None)

    def _dispatch(self, x, sigmoid, N, H, W, norm):
        with self._lock:
            return self._dispatch_locked(x, sigmoid, N, H, W, norm)

    def _dispatch_locked(self, x, sigmoid, N, H, W, norm):
        dev = x.device
        if self.weights is None or self.weights_device != dev or self.weights.stale(self.model):
            self.weights = PackedWeights(self.model, dev)
            self.weights_device = dev
            self.plans.clear()
        chunk = self.sub_batch(N, H, W)
        while True:
            try:
                if N > chunk:
                    # the kernels address each tensor through 32-bit buffer offsets (< 4 GiB per tensor): run contiguous, equally sized
                    # sub-batches (images are independent; results are byte-identical to one big batch) and concatenate
                    parts = [self._run(x[i:i + chunk], sigmoid, H, W, norm) for i in range(0, N, chunk)]
                    return OrderedDict((k, torch.cat([p[k] for p in parts], dim=0)) for k in parts[0])
                return self._run(x, sigmoid, H, W, norm)
            except BufferTooLarge:
                # max_batch() prices the backbone / head tensors; a neck option can hold something wider (deformable columns): split further
                if chunk == 1:
                    raise
                # one more part — and strictly fewer images per part (ceil(N / (parts + 1)) stops shrinking once chunk <= ~sqrt(N):
                # N = 9, chunk = 3 -> 3, an endless rebuild of the same plan)
                smaller = min(chunk - 1, -(-N // (-(-N // chunk) + 1)))
                assert 1 <= smaller < chunk
                self._sub_override[(H, W)] = chunk = smaller

    def sub_batch(self, N, H, W):
        """Images per launch plan: N when it fits the addressing limit, else N split into the fewest equal parts that do."""
        limit = min(self.max_batch(H, W), self._sub_override.get((H, W), 1 << 30))
        parts = -(-N // limit)
        return -(-N // parts)

    def max_batch(self, H, W):
        """Largest batch whose biggest activation tensor stays below the 4 GiB buffer-addressing limit of the kernels."""
        widest = max(64, max((l.cout for b in self.weights.head_blocks.values() for l in b), default=64))      # (the fused first head
                                                                                                          # blocks fall back to per-head launches)
        per_image = max((H // 2) * (W // 2) * 64, (H // 4) * (W // 4) * widest) * 4
        return max(1, ADDRESS_LIMIT // per_image)

    def _run(self, x, sigmoid, H, W, norm):
        dev = x.device
        N = x.shape[0]
        # one plan per stream: its arena, absmax slots and patched output pointers are private to that stream's launch order
        key = (N, H, W, bool(sigmoid), self.options, torch.cuda.current_stream(dev).cuda_stream)
        plan = self.plans.get(key)
        if plan is None:
            with torch.cuda.device(dev):
                plan = Plan(self.weights, N, H, W, dev, bool(sigmoid), self.options)
            self.plans[key] = plan
            while len(self.plans) > max(1, self.max_plans):
                self.plans.popitem(last=False)         # drop the least recently used plan (and its arena)
        else:
            self.plans.move_to_end(key)
        with torch.cuda.device(dev):
            return plan.run(x, norm)

    def plan_for(self, x, sigmoid=True):
        """The (existing) plan a forward of x on the current stream uses — bench / tests introspection."""
        N, _, H, W = x.shape
        n_sub = self.sub_batch(N, H, W)
        return self.plans[(n_sub, H, W, bool(sigmoid), self.options, torch.cuda.current_stream(x.device).cuda_stream)]
