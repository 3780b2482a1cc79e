"""Parameter containers for the CenterNet hot path: ResNet-34 backbone, simple / FPN neck, GenericHead.

These nn.Modules only HOLD parameters (so `state_dict()` / `load_state_dict()` / `.to()` behave like the
reference's modules and torchvision ResNet checkpoints load by key name); they are never *called* — the
forward pass is the HIP launch plan in engine.py.

Structure restated from (the Gen-A sources are missing from the reference tree):
  backbone  public torchvision ResNet-34 topology — conv1 7x7/2, bn1, maxpool 3x3/2, BasicBlock x [3,4,6,3],
            channels 64/128/256/512, 1x1/2 downsample — feature contract tests/test_backbones.py:60-70
  neck      models/layers.py:40-101 (make_conv, make_upsample), :138-177 (Fuse); contract tests/test_necks.py:23-56
  heads     models/meta.py:21-30 (GenericHead), models/fairmot.py:20 (reid defaults)
"""
from collections import OrderedDict

import torch
from torch import nn

RESNET_LAYERS = {"resnet18": [2, 2, 2, 2], "resnet34": [3, 4, 6, 3]}
BACKBONE_CHANNELS = [64, 64, 128, 256, 512]          # strides 2, 4, 8, 16, 32 (tests/test_necks.py:7-8)


class ConvBn(nn.Module):
    """conv (no bias) + BatchNorm2d (+ ReLU at run time): make_conv "normal" (layers.py:72-77) / ConvBnAct."""

    def __init__(self, cin, cout, k=3, stride=1, names=("conv", "bn")):
        super().__init__()
        self._names = names
        self.k, self.stride = k, stride
        self.add_module(names[0], nn.Conv2d(cin, cout, k, stride=stride, padding=(k - 1) // 2, bias=False))
        self.add_module(names[1], nn.BatchNorm2d(cout))
        nn.init.kaiming_normal_(self.conv_module.weight, mode="fan_out", nonlinearity="relu")   # layers.py:77

    @property
    def conv_module(self):
        return getattr(self, self._names[0])

    @property
    def bn_module(self):
        return getattr(self, self._names[1])


class SeparableConvBn(nn.Module):
    """make_conv(conv_type="separable") (layers.py:56-69): depthwise 3x3 (groups=C, no bias) + BN + ReLU6, then pointwise 1x1
    (no bias) + BN + ReLU6; state_dict keys 0, 1, 3, 4 as in the reference's nn.Sequential."""

    def __init__(self, cin, cout, k=3, depth_multiplier=1):
        super().__init__()
        if depth_multiplier != 1:
            raise ValueError("separable conv: depth_multiplier != 1 is not constructible in the reference either "
                             "(layers.py:60 sizes the BatchNorm with in_channels)")
        self.k = k
        self.add_module("0", nn.Conv2d(cin, cin, k, padding=(k - 1) // 2, groups=cin, bias=False))
        self.add_module("1", nn.BatchNorm2d(cin))
        self.add_module("3", nn.Conv2d(cin, cout, 1, bias=False))
        self.add_module("4", nn.BatchNorm2d(cout))
        nn.init.kaiming_normal_(getattr(self, "0").weight, mode="fan_out", nonlinearity="relu")     # layers.py:68-69
        nn.init.kaiming_normal_(getattr(self, "3").weight, mode="fan_out", nonlinearity="relu")

    dw = property(lambda self: getattr(self, "0"))
    dw_bn = property(lambda self: getattr(self, "1"))
    pw = property(lambda self: getattr(self, "3"))
    pw_bn = property(lambda self: getattr(self, "4"))


class _DeformableBlock(nn.Module):
    """Parameters of layers.py:9-38 DeformableConv2dBlock: offset_conv (zero-initialised), mask_conv.0 (+ Sigmoid, version 2 only),
    deform_conv.weight [out, in, k, k] (torchvision DeformConv2d, bias=False)."""

    def __init__(self, cin, cout, k=3, version=2, mask_init_bias=0.0):
        super().__init__()
        pad = (k - 1) // 2
        self.offset_conv = nn.Conv2d(cin, 2 * k * k, k, padding=pad)
        self.mask_conv = nn.Sequential(nn.Conv2d(cin, k * k, k, padding=pad), nn.Sigmoid()) if version == 2 else None
        self.deform_conv = nn.Conv2d(cin, cout, k, padding=pad, bias=False)      # same parameter shape / default init as DeformConv2d
        nn.init.constant_(self.offset_conv.weight, 0)
        nn.init.constant_(self.offset_conv.bias, 0)
        if self.mask_conv is not None:
            nn.init.constant_(self.mask_conv[0].weight, 0)
            nn.init.constant_(self.mask_conv[0].bias, mask_init_bias)


class DeformableConvBn(nn.Module):
    """make_conv(conv_type="deformable") (layers.py:47-54): DeformableConv2dBlock + BN + ReLU, state_dict keys 0.*, 1.*."""

    def __init__(self, cin, cout, k=3, version=2, mask_activation=None, mask_init_bias=0.0):
        super().__init__()
        if mask_activation not in (None, "Sigmoid"):
            raise ValueError(f"mask_activation={mask_activation!r}: the gfx950 deformable path applies the reference default (Sigmoid)")
        if version not in (1, 2):
            raise ValueError(f"deformable conv version {version}: expected 1 or 2")
        self.k, self.version = k, version
        self.add_module("0", _DeformableBlock(cin, cout, k, version, mask_init_bias))
        self.add_module("1", nn.BatchNorm2d(cout))

    block = property(lambda self: getattr(self, "0"))
    bn = property(lambda self: getattr(self, "1"))


class DeconvBn(nn.Module):
    """make_upsample(upsample_type="conv_transpose") (layers.py:86-96): ConvTranspose2d(C, C, k, stride=2, padding, output_padding,
    bias=False) + BN + ReLU, keys 0, 1; `deconv_init_bilinear` reproduces _init_bilinear_upsampling (layers.py:103-116) as
    written — it fills w[c, 0] only."""

    def __init__(self, channels, kernel=3, init_bilinear=True):
        super().__init__()
        if kernel not in (2, 3, 4):
            raise ValueError(f"deconv_kernel={kernel}: the gfx950 transposed-conv path covers kernels 2, 3 and 4")
        op = kernel % 2
        self.kernel = kernel
        self.add_module("0", nn.ConvTranspose2d(channels, channels, kernel, stride=2, padding=(kernel + op) // 2 - 1,
                                                output_padding=op, bias=False))
        self.add_module("1", nn.BatchNorm2d(channels))
        if init_bilinear:
            import math
            w = getattr(self, "0").weight.data
            f = math.ceil(w.size(2) / 2)
            c = (2 * f - 1 - f % 2) / (f * 2.0)
            for i in range(w.size(2)):
                for j in range(w.size(3)):
                    w[0, 0, i, j] = (1 - math.fabs(i / f - c)) * (1 - math.fabs(j / f - c))
            for ch in range(1, w.size(0)):
                w[ch, 0, :, :] = w[0, 0, :, :]

    deconv = property(lambda self: getattr(self, "0"))
    bn = property(lambda self: getattr(self, "1"))


def make_conv_params(cin, cout, conv_type="normal", **kw):
    """Parameter container of layers.py:40-79 make_conv (kernel 3)."""
    if conv_type == "normal":
        return ConvBn(cin, cout, 3, names=("0", "1"))
    if conv_type == "separable":
        return SeparableConvBn(cin, cout, 3)
    if conv_type == "deformable":
        return DeformableConvBn(cin, cout, 3, version=kw.get("version", 2), mask_activation=kw.get("mask_activation"),
                                mask_init_bias=kw.get("mask_init_bias", 0.0))
    raise ValueError(f"conv_type={conv_type!r}: expected 'normal', 'separable' or 'deformable' (layers.py:43)")


UPSAMPLE_TYPES = ("nearest", "bilinear", "conv_transpose")


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.stride = stride
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout))
        for m in (self.conv1, self.conv2) + ((self.downsample[0],) if self.downsample is not None else ()):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class ResNetBackbone(nn.Module):
    """Parameter layout of torchvision `resnet34` (keys conv1, bn1, layer{1..4}.{i}.conv{1,2}, bn{1,2}, downsample.{0,1})."""

    def __init__(self, name="resnet34", pretrained=False, frozen_stages=0, **ignored):
        super().__init__()
        if name not in RESNET_LAYERS:
            raise ValueError(f"backbone '{name}' is outside the MI355X hot-path scope (supported: {sorted(RESNET_LAYERS)})")
        self.name = name
        self.out_channels = list(BACKBONE_CHANNELS)
        self.output_stride = 32
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        nn.init.kaiming_normal_(self.conv1.weight, mode="fan_out", nonlinearity="relu")
        cin = 64
        for li, (n_blocks, cout) in enumerate(zip(RESNET_LAYERS[name], [64, 128, 256, 512])):
            blocks = []
            for b in range(n_blocks):
                blocks.append(BasicBlock(cin, cout, stride=2 if (b == 0 and li > 0) else 1))
                cin = cout
            setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))
        # `pretrained: True` (configs/base_resnet34.yaml:5) cannot be honoured offline; weights arrive through
        # load_state_dict().  Recorded so callers can see the request.
        self.pretrained_requested = bool(pretrained)
        if self.pretrained_requested:
            import warnings
            warnings.warn("backbone.pretrained=True is recorded but NOT honoured (no network access / no torchvision here): the ResNet "
                          "backbone is randomly initialised until weights are loaded — model.backbone.load_state_dict(torchvision_sd, "
                          "strict=False) or formats.load_checkpoint(model, ckpt)", stacklevel=3)


class SimpleNeck(nn.Module):
    """3 x [conv3x3+BN+ReLU -> x2 upsample] on the last backbone feature (docs/implementation.md:40-48;
    tests/test_necks.py:23-39).  conv = make_conv(conv_type), upsample = make_upsample(upsample_type, deconv_channels=c)
    (layers.py:40-101; option names of configs/test_config.yaml:8-18)."""

    def __init__(self, backbone_channels, upsample_channels=(256, 128, 64), upsample_type="nearest", conv_type="normal",
                 deconv_kernel=3, deconv_init_bilinear=True, **conv_kw):
        super().__init__()
        _check_neck_options(upsample_type, conv_type)
        self.upsample_type, self.conv_type = upsample_type, conv_type
        self.out_channels = upsample_channels[-1]
        self.upsample_stride = 2 ** len(upsample_channels)
        layers, ups = [], []
        cin = backbone_channels[-1]
        for c in upsample_channels:
            layers.append(make_conv_params(cin, c, conv_type, **conv_kw))
            ups.append(DeconvBn(c, deconv_kernel, deconv_init_bilinear) if upsample_type == "conv_transpose" else nn.Identity())
            cin = c
        self.layers = nn.Sequential(*layers)
        self.upsamples = nn.ModuleList(ups)


class FuseParams(nn.Module):
    """Parameters of layers.py:138-177 `Fuse(in_channels=[skip_c, top_c], out, resize="up", upsample, conv_type, weighted_fusion)`:
    optional 1x1 projections WITH bias where channels differ (:152), the resize layer (:154, parameters only for
    conv_transpose), the fusion weights (:148) and the output conv (:158).  Key names equal the reference module's."""

    def __init__(self, skip_c, top_c, out, upsample="nearest", conv_type="normal", weighted_fusion=False, deconv_kernel=3,
                 deconv_init_bilinear=True, **conv_kw):
        super().__init__()
        self.upsample_type, self.conv_type = upsample, conv_type
        self.weights = nn.Parameter(torch.ones(2), requires_grad=True) if weighted_fusion else None
        self.project = nn.ModuleList([nn.Conv2d(skip_c, out, 1) if skip_c != out else nn.Identity(),
                                      nn.Conv2d(top_c, out, 1) if top_c != out else nn.Identity()])
        self.resize = DeconvBn(out, deconv_kernel, deconv_init_bilinear) if upsample == "conv_transpose" else nn.Identity()
        self.output_conv = make_conv_params(out, out, conv_type, **conv_kw)


class FPNNeck(nn.Module):
    """top = 1x1(c5 -> up[0]); level i: Fuse([skip_{/16,/8,/4}, top] -> up[i]) (SURVEY.md §8c decision (i);
    docs/implementation.md:49-52; tests/test_necks.py:41-56)."""

    def __init__(self, backbone_channels, upsample_channels=(256, 128, 64), upsample_type="nearest", conv_type="normal",
                 weighted_fusion=False, deconv_kernel=3, deconv_init_bilinear=True, **conv_kw):
        super().__init__()
        _check_neck_options(upsample_type, conv_type)
        if len(upsample_channels) > len(backbone_channels) - 1:
            raise ValueError("FPN neck needs one backbone skip feature per upsample stage")
        self.upsample_type, self.conv_type, self.weighted_fusion = upsample_type, conv_type, bool(weighted_fusion)
        self.out_channels = upsample_channels[-1]
        self.upsample_stride = 2 ** len(upsample_channels)
        self.top_conv = nn.Conv2d(backbone_channels[-1], upsample_channels[0], 1)
        fuse = []
        top_c = upsample_channels[0]
        for i, c in enumerate(upsample_channels):
            skip_c = backbone_channels[-2 - i]
            fuse.append(FuseParams(skip_c, top_c, c, upsample_type, conv_type, weighted_fusion, deconv_kernel, deconv_init_bilinear,
                                   **conv_kw))
            top_c = c
        self.fuse = nn.ModuleList(fuse)


class FuseNode(nn.Module):
    """layers.py:138-177 `Fuse(in_channels, out, resize, upsample, downsample, conv_type, weighted_fusion)` for any number of inputs —
    the node IDA and BiFPN necks are made of: 1x1 projections WITH bias where channels differ (:152), the LAST input resized (:154-156;
    "down" is always MaxPool2d(2, 2): the reference passes `downsample=` where make_downsample expects `downsample_type=`, :118/:156,
    so the choice never arrives), plain or weighted sum (:163-171), output conv (:158).  Key names equal the reference module's."""

    def __init__(self, in_channels, out, resize="up", upsample="nearest", conv_type="normal", weighted_fusion=False, deconv_kernel=3,
                 deconv_init_bilinear=True, **conv_kw):
        super().__init__()
        if resize not in ("up", "down"):
            raise ValueError(f"Fuse resize={resize!r}: 'up' or 'down' (layers.py:144)")
        self.resize_kind, self.upsample_type, self.conv_type = resize, upsample, conv_type
        self.weights = nn.Parameter(torch.ones(len(in_channels)), requires_grad=True) if weighted_fusion else None
        self.project = nn.ModuleList([nn.Conv2d(c, out, 1) if c != out else nn.Identity() for c in in_channels])
        self.resize = DeconvBn(out, deconv_kernel, deconv_init_bilinear) if (resize == "up" and upsample == "conv_transpose") else nn.Identity()
        self.output_conv = make_conv_params(out, out, conv_type, **conv_kw)


class IDANeck(nn.Module):
    """"Iteratively fuse consecutive feature maps from backbone until there is only 1 feature map left" (docs/implementation.md:43; the
    class is missing from the reference tree — tests/test_necks.py:61-62 is an empty test).  Defined on the reference's Fuse node over the
    features at strides 4 .. 32: stage s maps the levels to [Fuse([c_i, c_{i+1}], out=c_i, "up")(level_i, level_{i+1})]; three stages
    leave one map with the stride-4 feature's channel count at stride 4.  State keys: neck.stages.{s}.{i}.<Fuse keys>."""

    def __init__(self, backbone_channels, upsample_type="nearest", conv_type="normal", weighted_fusion=False, deconv_kernel=3,
                 deconv_init_bilinear=True, **conv_kw):
        super().__init__()
        _check_neck_options(upsample_type, conv_type)
        self.upsample_type, self.conv_type, self.weighted_fusion = upsample_type, conv_type, bool(weighted_fusion)
        ch = list(backbone_channels[1:])                      # strides 4, 8, 16, 32
        self.out_channels = ch[0]
        self.upsample_stride = 2 ** (len(ch) - 1)
        stages = []
        while len(ch) > 1:
            stages.append(nn.ModuleList([FuseNode([ch[i], ch[i + 1]], ch[i], "up", upsample_type, conv_type, weighted_fusion, deconv_kernel,
                                                  deconv_init_bilinear, **conv_kw) for i in range(len(ch) - 1)]))
            ch = ch[:-1]
        self.stages = nn.ModuleList(stages)


class _BiFPNLayer(nn.Module):
    def __init__(self, in_ch, C, last, kw):
        super().__init__()
        n = len(in_ch)
        # td[i], i = 0 .. n-2: Fuse([in_i, td_{i+1}], C, "up"); td_{n-1} is the input itself
        self.td = nn.ModuleList([FuseNode([in_ch[i], C if i < n - 2 else in_ch[n - 1]], C, "up", **kw) for i in range(n - 1)])
        # bu[i-1], i = 1 .. n-1: Fuse([in_i, td_i, out_{i-1}], C, "down"); the coarsest level has no td node of its own
        self.bu = None if last else nn.ModuleList(
            [FuseNode([in_ch[i], C, C], C, "down", **kw) for i in range(1, n - 1)] + [FuseNode([in_ch[n - 1], C], C, "down", **kw)])


class BiFPNNeck(nn.Module):
    """EfficientDet's BiFPN (docs/implementation.md:42; class missing from the reference tree, tests/test_necks.py:58-59 is empty), on the
    reference's Fuse node over the features at strides 4 .. 32.  Per layer: top-down td_i = Fuse([in_i, td_{i+1}], C, "up"), then
    bottom-up out_i = Fuse([in_i, td_i, out_{i-1}], C, "down") (out_0 = td_0; the coarsest level fuses [in, out_{below}]).  The neck
    returns out_0 of the last layer, so the last layer builds no bottom-up nodes.  `num_channels` = C, `num_layers` >= 1.
    State keys: neck.bifpn.{l}.td.{i}.* / neck.bifpn.{l}.bu.{i}.*."""

    def __init__(self, backbone_channels, num_channels=64, num_layers=3, upsample_type="nearest", conv_type="normal",
                 weighted_fusion=False, deconv_kernel=3, deconv_init_bilinear=True, **conv_kw):
        super().__init__()
        _check_neck_options(upsample_type, conv_type)
        if int(num_layers) < 1 or int(num_channels) < 4 or int(num_channels) % 4:
            raise ValueError(f"bifpn: num_layers={num_layers} (>= 1), num_channels={num_channels} (a positive multiple of 4)")
        self.upsample_type, self.conv_type, self.weighted_fusion = upsample_type, conv_type, bool(weighted_fusion)
        self.num_channels, self.num_layers = int(num_channels), int(num_layers)
        self.out_channels = self.num_channels
        in_ch = list(backbone_channels[1:])
        self.upsample_stride = 2 ** (len(in_ch) - 1)
        kw = dict(upsample=upsample_type, conv_type=conv_type, weighted_fusion=weighted_fusion, deconv_kernel=deconv_kernel,
                  deconv_init_bilinear=deconv_init_bilinear, **conv_kw)
        layers = []
        for l in range(self.num_layers):
            last = l == self.num_layers - 1
            layers.append(_BiFPNLayer(in_ch, self.num_channels, last, kw))
            # what the next layer reads: out_0 .. out_{n-1}, all C channels
            in_ch = [self.num_channels] * len(in_ch)
        self.bifpn = nn.ModuleList(layers)


def _check_neck_options(upsample_type, conv_type):
    if upsample_type not in UPSAMPLE_TYPES:
        raise ValueError(f"upsample_type={upsample_type!r}: expected one of {UPSAMPLE_TYPES} (layers.py:84)")
    if conv_type not in ("normal", "separable", "deformable"):
        make_conv_params(64, 64, conv_type)          # raises with the explanation


class GenericHead(nn.Module):
    """models/meta.py:21-30: depth x (3x3 conv + BN + ReLU, `width` channels) then a 1x1 conv with bias, the bias
    filled with init_bias."""

    def __init__(self, in_channels, out_channels, width=256, depth=3, init_bias=None):
        super().__init__()
        self.in_channels, self.out_channels, self.width, self.depth = in_channels, out_channels, width, depth
        for i in range(depth):
            self.add_module(f"block_{i + 1}", ConvBn(in_channels if i == 0 else width, width, 3))
        self.out_conv = nn.Conv2d(width if depth > 0 else in_channels, out_channels, 1)
        if init_bias is not None:
            self.out_conv.bias.data.fill_(init_bias)

    def blocks(self):
        return [getattr(self, f"block_{i + 1}") for i in range(self.depth)]


def build_neck(cfg, backbone_channels):
    cfg = dict(cfg)
    name = cfg.pop("name")
    if name == "simple":
        return SimpleNeck(backbone_channels, **cfg)
    if name == "fpn":
        return FPNNeck(backbone_channels, **cfg)
    if name == "ida":
        return IDANeck(backbone_channels, **cfg)
    if name == "bifpn":
        return BiFPNNeck(backbone_channels, **cfg)
    raise ValueError(f"neck '{name}': expected one of simple, fpn, bifpn, ida (README.md:70 of the reference)")
