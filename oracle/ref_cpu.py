"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product package.

Plain PyTorch CPU fp32 restatement of the CenterNet forward that the HIP path must match:
  GenericModel.forward            <- centernet_lightning/models/meta.py:41-47
  GenericHead                     <- centernet_lightning/models/meta.py:21-30 (block = conv3x3 no-bias + BN + ReLU)
  make_conv / make_upsample / Fuse <- centernet_lightning/models/layers.py:40-101, :138-177 (normal + separable conv; nearest,
                                     bilinear and conv_transpose upsampling; plain and weighted fusion)
  ResNet-34 backbone              <- public torchvision topology (absent from the reference tree; contract
                                     tests/test_backbones.py:60-70)
  .sigmoid() at the forward boundary <- centernet_lightning/models/centernet.py:205

It is op-per-layer and unfused on purpose (F.conv2d, F.batch_norm in eval mode, F.relu, F.max_pool2d,
F.interpolate(nearest), explicit adds): exactly the ATen call sequence of the reference, so BN folding,
upsample folding, head fusion and the MFMA summation order of the product are all exercised by the
comparison.  The structure is read from the state_dict's key names, so this file does not import the
product's model code.

Parity status: heads/wiring PINNED by tests/golden/head_wiring.npz (outputs of the reference's
GenericHead / GenericModel); backbone + neck "parity unpinned" by the reference (its sources and tests
for them are missing — SURVEY.md §8c) and defined against this restatement.
"""
import re
from collections import OrderedDict

import torch
import torch.nn.functional as F

EPS = 1e-5


def _bn(x, sd, prefix, stats=None):
    if stats is not None:                      # calibration: measure batch stats, store them, use them
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        sd[prefix + ".running_mean"].copy_(mean)
        sd[prefix + ".running_var"].copy_(var.clamp_min(1e-3))
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], training=False, eps=EPS)


def _conv_bn_relu(x, sd, conv, bn, stride=1, relu=True, stats=None):
    w = sd[conv + ".weight"]
    y = F.conv2d(x, w, None, stride=stride, padding=(w.shape[-1] - 1) // 2)
    y = _bn(y, sd, bn, stats)
    return F.relu(y) if relu else y


def backbone_features(sd, x, stats=None):
    """-> [f/2, f/4, f/8, f/16, f/32] with channels [64,64,128,256,512]."""
    p = "backbone."
    x = _conv_bn_relu(x, sd, p + "conv1", p + "bn1", stride=2, stats=stats)
    feats = [x]
    x = F.max_pool2d(x, 3, 2, 1)
    for li in range(1, 5):
        bi = 0
        while f"{p}layer{li}.{bi}.conv1.weight" in sd:
            q = f"{p}layer{li}.{bi}."
            stride = 2 if (bi == 0 and li > 1) else 1
            out = _conv_bn_relu(x, sd, q + "conv1", q + "bn1", stride=stride, stats=stats)
            out = _conv_bn_relu(out, sd, q + "conv2", q + "bn2", relu=False, stats=stats)
            if q + "downsample.0.weight" in sd:
                idn = F.conv2d(x, sd[q + "downsample.0.weight"], None, stride=stride)
                idn = _bn(idn, sd, q + "downsample.1", stats)
            else:
                idn = x
            x = F.relu(out + idn)
            bi += 1
        feats.append(x)
    return feats


def _bilinear_zero(x, py, px):
    """torchvision deform_conv2d's bilinear_interpolate (torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp): 0 when the point is
    outside (-1, H) x (-1, W); corners outside the image contribute 0.  x [N,C,H,W]; py, px [N,Ho,Wo] -> [N,C,Ho,Wo]."""
    N, C, H, W = x.shape
    inside = (py > -1) & (py < H) & (px > -1) & (px < W)
    h_low, w_low = torch.floor(py), torch.floor(px)
    lh, lw = py - h_low, px - w_low
    hh, hw = 1 - lh, 1 - lw
    h_low, w_low = h_low.long(), w_low.long()
    h_high, w_high = h_low + 1, w_low + 1
    flat = x.reshape(N, C, H * W)

    def corner(hy, wx, ok):
        idx = (hy.clamp(0, H - 1) * W + wx.clamp(0, W - 1)).reshape(N, 1, -1).expand(N, C, -1)
        v = flat.gather(2, idx).reshape(N, C, *py.shape[1:])
        return v * (ok & inside).unsqueeze(1).to(x.dtype)

    v1 = corner(h_low, w_low, (h_low >= 0) & (w_low >= 0))
    v2 = corner(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
    v3 = corner(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
    v4 = corner(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))
    w1, w2, w3, w4 = (hh * hw).unsqueeze(1), (hh * lw).unsqueeze(1), (lh * hw).unsqueeze(1), (lh * lw).unsqueeze(1)
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4


def deform_conv2d(x, offset, weight, mask=None, padding=1):
    """torchvision.ops.deform_conv2d restated (stride 1, dilation 1, one offset group) — third-party, absent from the image
    ("parity unpinned"): out(p) = sum_k w_k . x(p + p_k + dp_k) . m_k, offsets stored (dy, dx) interleaved per tap."""
    N, C, H, W = x.shape
    O, _, kh, kw = weight.shape
    ys = torch.arange(H, dtype=x.dtype).view(1, H, 1)
    xs = torch.arange(W, dtype=x.dtype).view(1, 1, W)
    out = torch.zeros(N, O, H, W, dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            k = i * kw + j
            py = ys - padding + i + offset[:, 2 * k]
            px = xs - padding + j + offset[:, 2 * k + 1]
            val = _bilinear_zero(x, py, px)
            if mask is not None:
                val = val * mask[:, k].unsqueeze(1)
            out += torch.einsum("oc,nchw->nohw", weight[:, :, i, j], val)
    return out


def make_conv_forward(x, sd, q, stats=None):
    """layers.py:40-79 make_conv: `q`.0/.1 = conv3x3+BN+ReLU ("normal"); `q`.0/.1/.3/.4 = depthwise+BN+ReLU6, pointwise+BN+ReLU6
    ("separable", :56-69)."""
    if q + ".0.deform_conv.weight" in sd:                    # DeformableConv2dBlock (layers.py:9-38) + BN + ReLU (:47-54)
        w = sd[q + ".0.deform_conv.weight"]
        pad = (w.shape[-1] - 1) // 2
        offset = F.conv2d(x, sd[q + ".0.offset_conv.weight"], sd[q + ".0.offset_conv.bias"], padding=pad)
        mask = None
        if q + ".0.mask_conv.0.weight" in sd:               # version 2: modulated, sigmoid mask
            mask = torch.sigmoid(F.conv2d(x, sd[q + ".0.mask_conv.0.weight"], sd[q + ".0.mask_conv.0.bias"], padding=pad))
        return F.relu(_bn(deform_conv2d(x, offset, w, mask, pad), sd, q + ".1", stats))
    if q + ".3.weight" in sd:
        w = sd[q + ".0.weight"]
        y = F.conv2d(x, w, None, padding=(w.shape[-1] - 1) // 2, groups=w.shape[0])
        y = F.relu6(_bn(y, sd, q + ".1", stats))
        y = F.conv2d(y, sd[q + ".3.weight"], None)
        return F.relu6(_bn(y, sd, q + ".4", stats))
    return _conv_bn_relu(x, sd, q + ".0", q + ".1", stats=stats)


def make_upsample_forward(x, sd, q, upsample_type, stats=None):
    """layers.py:81-101 make_upsample: ConvTranspose2d(stride 2)+BN+ReLU when `q`.0.weight exists, else nn.Upsample(x2, mode)."""
    if q + ".0.weight" in sd:
        w = sd[q + ".0.weight"]
        k = w.shape[-1]
        op = k % 2
        y = F.conv_transpose2d(x, w, None, stride=2, padding=(k + op) // 2 - 1, output_padding=op)
        return F.relu(_bn(y, sd, q + ".1", stats))
    return F.interpolate(x, scale_factor=2, mode=upsample_type)


def fuse_forward(sd, q, skip, top, upsample_type="nearest", stats=None, eps=1e-6):
    """Fuse.forward (layers.py:160-177) with in_channels = [skip, top], resize="up"."""
    out = [skip, top]
    for j in range(2):
        if f"{q}project.{j}.weight" in sd:
            out[j] = F.conv2d(out[j], sd[f"{q}project.{j}.weight"], sd[f"{q}project.{j}.bias"])
    out[-1] = make_upsample_forward(out[-1], sd, q + "resize", upsample_type, stats)
    if q + "weights" in sd:
        w = F.relu(sd[q + "weights"])
        o = torch.stack([out[j] * w[j] for j in range(2)], dim=-1)
        o = torch.sum(o, dim=-1) / (torch.sum(w) + eps)
    else:
        o = torch.stack(out, dim=-1).sum(dim=-1)
    return make_conv_forward(o, sd, q + "output_conv", stats)


def fuse_forward_n(sd, q, inputs, resize="up", upsample_type="nearest", stats=None, eps=1e-6):
    """Fuse.forward (layers.py:138-177) for any number of inputs: every input projected by its 1x1 conv where one exists, the LAST one
    resized — "up": make_upsample (:154), "down": make_downsample (:156) — then the plain or weighted sum and the output conv.
    Reference quirk kept: Fuse passes `downsample=` (not `downsample_type=`) to make_downsample (:156), where it disappears into
    **kwargs (:118), so the down path is ALWAYS nn.MaxPool2d(2, 2) whatever the `downsample` argument says."""
    out = list(inputs)
    for j in range(len(out)):
        if f"{q}project.{j}.weight" in sd:
            out[j] = F.conv2d(out[j], sd[f"{q}project.{j}.weight"], sd[f"{q}project.{j}.bias"])
    if resize == "up":
        out[-1] = make_upsample_forward(out[-1], sd, q + "resize", upsample_type, stats)
    else:
        out[-1] = F.max_pool2d(out[-1], 2, 2)
    if q + "weights" in sd:
        w = F.relu(sd[q + "weights"])
        o = torch.stack([out[j] * w[j] for j in range(len(out))], dim=-1)
        o = torch.sum(o, dim=-1) / (torch.sum(w) + eps)
    else:
        o = torch.stack(out, dim=-1).sum(dim=-1)
    return make_conv_forward(o, sd, q + "output_conv", stats)


def ida_forward(sd, feats, stats=None, upsample_type="nearest"):
    """IDANeck — "iteratively fuse consecutive feature maps from backbone until there is only 1 feature map left"
    (docs/implementation.md:43; the class itself is missing from the reference tree, tests/test_necks.py:61-62 is empty).  Defined here
    on the reference's own Fuse node: levels = features at strides 4, 8, 16, 32; stage s replaces them by
    [Fuse([c_i, c_{i+1}], out = c_i, "up")(level_i, level_{i+1})]; after three stages one 64-channel map at stride 4 is left."""
    levels = list(feats[1:])
    s = 0
    while len(levels) > 1:
        levels = [fuse_forward_n(sd, f"neck.stages.{s}.{i}.", [levels[i], levels[i + 1]], "up", upsample_type, stats) for i in range(len(levels) - 1)]
        s += 1
    return levels[0]


def bifpn_forward(sd, feats, stats=None, upsample_type="nearest"):
    """BiFPNNeck (docs/implementation.md:42: EfficientDet's BiFPN; class missing from the reference tree, tests/test_necks.py:58-59 is
    empty).  Defined here on the reference's Fuse node over the features at strides 4 .. 32 (P2 .. P5), per layer:
        top-down   td_3 = in_3;  td_i = Fuse([in_i, td_{i+1}], C, "up")                        i = 2, 1, 0
        bottom-up  out_0 = td_0; out_i = Fuse([in_i, td_i, out_{i-1}], C, "down")              i = 1, 2;   out_3 = Fuse([in_3, out_2], C, "down")
    The neck's output is out_0 of the last layer, so the last layer has no bottom-up nodes (nothing reads them)."""
    ins = list(feats[1:])
    l = 0
    has = lambda prefix: any(k.startswith(prefix) for k in sd)
    while has(f"neck.bifpn.{l}.td.0."):
        q = f"neck.bifpn.{l}."
        td = [None, None, None, ins[3]]
        for i in (2, 1, 0):
            td[i] = fuse_forward_n(sd, f"{q}td.{i}.", [ins[i], td[i + 1]], "up", upsample_type, stats)
        if has(f"{q}bu.0."):
            outs = [td[0], None, None, None]
            for i in (1, 2):
                outs[i] = fuse_forward_n(sd, f"{q}bu.{i - 1}.", [ins[i], td[i], outs[i - 1]], "down", upsample_type, stats)
            outs[3] = fuse_forward_n(sd, f"{q}bu.2.", [ins[3], outs[2]], "down", upsample_type, stats)
            ins = outs
        else:
            ins = [td[0], td[1], td[2], td[3]]
        l += 1
    return ins[0]


def neck_forward(sd, feats, stats=None, upsample_type="nearest"):
    """`upsample_type` distinguishes "nearest" from "bilinear" (neither has parameters); "conv_transpose" is read off the keys."""
    if any(k.startswith("neck.stages.0.0.") for k in sd):      # IDA
        return ida_forward(sd, feats, stats, upsample_type)
    if any(k.startswith("neck.bifpn.0.td.0.") for k in sd):    # BiFPN
        return bifpn_forward(sd, feats, stats, upsample_type)
    if "neck.top_conv.weight" in sd:                           # FPN
        top = F.conv2d(feats[-1], sd["neck.top_conv.weight"], sd["neck.top_conv.bias"])
        i = 0
        while f"neck.fuse.{i}.output_conv.1.weight" in sd:
            top = fuse_forward(sd, f"neck.fuse.{i}.", feats[-2 - i], top, upsample_type, stats)
            i += 1
        return top
    x = feats[-1]                                              # simple neck: conv -> upsample per stage
    i = 0
    while f"neck.layers.{i}.1.weight" in sd:
        x = make_conv_forward(x, sd, f"neck.layers.{i}", stats)
        x = make_upsample_forward(x, sd, f"neck.upsamples.{i}", upsample_type, stats)
        i += 1
    return x


def head_names(sd):
    names = []
    for k in sd:
        m = re.match(r"heads\.([^.]+)\.out_conv\.weight", k)
        if m:
            names.append(m.group(1))
    return names


def head_forward(sd, name, x, stats=None, prefix="heads.", return_features=False):
    """GenericHead.forward (meta.py:21-30).  return_features: also the last block's output (what out_conv reads)."""
    q = f"{prefix}{name}."
    i = 1
    while f"{q}block_{i}.conv.weight" in sd:
        x = _conv_bn_relu(x, sd, f"{q}block_{i}.conv", f"{q}block_{i}.bn", stats=stats)
        i += 1
    y = F.conv2d(x, sd[q + "out_conv.weight"], sd[q + "out_conv.bias"])
    return (y, x) if return_features else y


@torch.no_grad()
def forward(sd, x, sigmoid=True, stats=None, return_intermediates=False, upsample_type="nearest"):
    """sd: state_dict (CPU fp32 tensors) with the key layout of centernet_lightning_amd.CenterNet;
    x: [N,3,H,W] CPU fp32.  Returns OrderedDict(heatmap, box_2d[, reid]) in NCHW.
    return_intermediates: (out, backbone features, neck output); "heads": additionally the dict of every head's last-block output
    (the 256-channel tensor its out_conv reads) — the feature-level parity gate of tests/test_gpu_e2e.py compares those."""
    feats = backbone_features(sd, x, stats)
    neck = neck_forward(sd, feats, stats, upsample_type)
    out, head_feats = OrderedDict(), OrderedDict()
    for name in head_names(sd):
        y, head_feats[name] = head_forward(sd, name, neck, stats, return_features=True)
        out[name] = y.sigmoid() if (name == "heatmap" and sigmoid) else y
    if return_intermediates == "heads":
        return out, feats, neck, head_feats
    if return_intermediates:
        return out, feats, neck
    return out


@torch.no_grad()
def forward_float64(sd, x, **kw):
    """The same forward in float64 (weights and input promoted): the reference against which fp32 implementations' rounding
    error is measured (CPU fp32 and the HIP kernels alike)."""
    sd64 = OrderedDict((k, v.double() if v.is_floating_point() else v) for k, v in sd.items())
    return forward(sd64, x.double(), **kw)


@torch.no_grad()
def synth_state_dict(model_state_dict, seed=0, calib_shape=(2, 3, 256, 256), calib_seed=1234, upsample_type="nearest"):
    """Synthetic weights per BASELINE.md §5 / SURVEY.md §8(d): convs keep their Kaiming init; BN gamma~U(0.5,1.5),
    beta~N(0,0.1); running stats CALIBRATED by one CPU forward on a seeded batch so activations stay O(1);
    out_conv.weight ~ N(0, 0.01^2); out_conv.bias keeps init_bias.  Returns a new CPU state_dict."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict((k, v.detach().clone().float().cpu() if v.is_floating_point() else v.detach().clone().cpu())
                     for k, v in model_state_dict.items())
    for k, v in sd.items():
        if k.endswith("running_var"):
            base = k[: -len("running_var")]
            sd[base + "weight"].copy_(torch.rand(v.shape, generator=g) + 0.5)
            sd[base + "bias"].copy_(torch.randn(v.shape, generator=g) * 0.1)
        elif k.endswith("out_conv.weight"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.01)
        elif k.endswith("weight") and v.dim() == 4:
            fan_out = v.shape[0] * v.shape[2] * v.shape[3]
            v.copy_(torch.randn(v.shape, generator=g) * (2.0 / fan_out) ** 0.5)       # kaiming_normal_(fan_out, relu)
            if k.endswith("offset_conv.weight"):
                # the reference zero-initialises the offset conv (layers.py:27); Kaiming-sized weights on O(10) activations would
                # throw the sampling points tens of pixels away and make the layer chaotically sensitive to 1e-6 input noise
                v.mul_(0.02)
        elif k.endswith(("top_conv.bias", "project.0.bias", "project.1.bias")):
            v.copy_(torch.randn(v.shape, generator=g) * 0.05)
        elif k.endswith(".weights") and v.dim() == 1:                  # Fuse fusion weights (layers.py:148)
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
    xg = torch.Generator().manual_seed(calib_seed)
    x = torch.rand(*calib_shape, generator=xg)
    forward(sd, x, stats=True, upsample_type=upsample_type)
    return sd
