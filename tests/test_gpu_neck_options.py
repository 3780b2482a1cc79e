"""GPU: the remaining neck options through the C ABI — transposed conv (sub-pixel phases on the MFMA
kernel), x2 nearest / bilinear upsampling (+ Fuse sum), depthwise 3x3, ReLU6 — against plain PyTorch CPU fp32, and whole
models with every option against the CPU oracle (oracle/ref_cpu.py, whose primitives are pinned to the reference's own
layers.py by tests/golden/layers_neck_options.npz).  Tolerance: rtol = atol = 1e-4 (north star); the HBM-bound elementwise
kernels are compared at 1e-6."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import recipes
import ref_cpu
import centernet_lightning_amd as cl
from centernet_lightning_amd import _lib, engine, params as P

pytestmark = pytest.mark.gpu
TOL = 1e-4


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("K,cin,cout,shape,res", [(3, 64, 64, (2, 9, 13), False), (4, 32, 32, (1, 16, 16), True),
                                                  (2, 64, 96, (3, 5, 7), False), (3, 256, 256, (1, 8, 8), True)])
def test_deconv2x_matches_conv_transpose(K, cin, cout, shape, res):
    lib = _lib.load()
    g = torch.Generator().manual_seed(K * 100 + cin)
    N, H, W = shape
    op = K % 2
    mod = P.DeconvBn(cin, K, init_bilinear=False)
    if cout != cin:                                            # the kernel is general in Cout; the reference always uses C -> C
        mod._modules["0"] = torch.nn.ConvTranspose2d(cin, cout, K, stride=2, padding=(K + op) // 2 - 1, output_padding=op, bias=False)
        mod._modules["1"] = torch.nn.BatchNorm2d(cout)
    with torch.no_grad():
        mod.deconv.weight.copy_(torch.randn(mod.deconv.weight.shape, generator=g) * 0.1)
        mod.bn.weight.copy_(torch.rand(cout, generator=g) + 0.5); mod.bn.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        mod.bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.2); mod.bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    mod.eval()
    x = torch.randn(N, cin, H, W, generator=g)
    r = torch.randn(N, cout, 2 * H, 2 * W, generator=g) if res else None
    with torch.no_grad():
        ref = F.relu(mod.bn(mod.deconv(x)))
        if res:
            ref = ref + r
    layer = engine._DeconvLayer(mod, torch.device("cuda:0"))
    xd, rd = nhwc(x), (nhwc(r) if res else None)
    y = torch.full((N, 2 * H, 2 * W, cout), float("nan"), device="cuda")
    p = _lib.DeconvParams()
    p.x, p.w, p.bias, p.y = xd.data_ptr(), layer.w.data_ptr(), layer.b.data_ptr(), y.data_ptr()
    p.residual = rd.data_ptr() if res else None
    p.N, p.H_in, p.W_in, p.Cin, p.Cout, p.K = N, H, W, cin, cout, K
    p.ldx, p.ldy, p.ldr, p.flags = cin, cout, cout, _lib.CNL_RELU
    _lib.check(lib.cnl_deconv2x_nhwc_f32(ctypes.byref(p), None))
    torch.testing.assert_close(nchw(y), ref, rtol=TOL, atol=TOL)


def test_deconv_argument_errors():
    lib = _lib.load()
    x = torch.zeros(4096, device="cuda")
    p = _lib.DeconvParams()
    p.x = p.w = p.bias = p.y = x.data_ptr()
    p.N, p.H_in, p.W_in, p.Cin, p.Cout, p.K, p.ldx, p.ldy = 1, 4, 4, 32, 32, 5, 32, 32
    assert lib.cnl_deconv2x_nhwc_f32(ctypes.byref(p), None) == _lib.CNL_E_UNSUPPORTED and "deconv_kernel" in _lib.last_error()
    p.K, p.Cin, p.ldx = 3, 24, 24
    assert lib.cnl_deconv2x_nhwc_f32(ctypes.byref(p), None) == _lib.CNL_E_UNSUPPORTED
    p.Cin, p.ldx, p.flags = 32, 32, _lib.CNL_SIGMOID
    assert lib.cnl_deconv2x_nhwc_f32(ctypes.byref(p), None) == _lib.CNL_E_UNSUPPORTED
    t, pd = ctypes.c_int32(), ctypes.c_int32()
    geo = {}
    for K in (2, 3, 4):
        for d in (0, 1):
            assert lib.cnl_deconv_phase_geometry(K, d, ctypes.byref(t), ctypes.byref(pd)) == 0
            geo[(K, d)] = (t.value, pd.value)
    assert geo == {(2, 0): (1, 0), (2, 1): (1, 0), (3, 0): (1, 0), (3, 1): (2, 0), (4, 0): (2, 1), (4, 1): (2, 0)}
    assert lib.cnl_deconv_weight_floats(64, 32, 3) == 64 * 32 * 9


@pytest.mark.parametrize("mode", ["nearest", "bilinear"])
@pytest.mark.parametrize("shape,res", [((2, 8, 5, 7), False), ((1, 64, 16, 12), True), ((3, 4, 1, 1), True)])
def test_upsample2x(mode, shape, res):
    lib = _lib.load()
    g = torch.Generator().manual_seed(7)
    N, C, H, W = shape
    x = torch.randn(*shape, generator=g)
    r = torch.randn(N, C, 2 * H, 2 * W, generator=g) if res else None
    ref = F.interpolate(x, scale_factor=2, mode=mode)
    if res:
        ref = r + ref
    xd, rd = nhwc(x), (nhwc(r) if res else None)
    y = torch.full((N, 2 * H, 2 * W, C), float("nan"), device="cuda")
    _lib.check(lib.cnl_upsample2x_nhwc_f32(xd.data_ptr(), rd.data_ptr() if res else None, y.data_ptr(), N, H, W, C, C, C, C,
                                           0 if mode == "nearest" else 1, None))
    if mode == "nearest":
        assert torch.equal(nchw(y), ref)
    else:
        torch.testing.assert_close(nchw(y), ref, rtol=0, atol=1e-6)


@pytest.mark.parametrize("shape,flags", [((2, 32, 9, 11), _lib.CNL_RELU6), ((1, 256, 16, 16), _lib.CNL_RELU6), ((1, 8, 1, 5), 0)])
def test_depthwise3x3(shape, flags):
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    N, C, H, W = shape
    x = torch.randn(*shape, generator=g) * 3
    w = torch.randn(C, 1, 3, 3, generator=g)
    b = torch.randn(C, generator=g)
    ref = F.conv2d(x, w, b, padding=1, groups=C)
    if flags:
        ref = F.relu6(ref)
    wd = w[:, 0].permute(1, 2, 0).contiguous().cuda()
    y = torch.full((N, H, W, C), float("nan"), device="cuda")
    xd, bd = nhwc(x), b.cuda()
    _lib.check(lib.cnl_depthwise3x3_nhwc_f32(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), N, H, W, C, C, C, flags, None))
    torch.testing.assert_close(nchw(y), ref, rtol=1e-5, atol=1e-5)
    if flags:
        assert float(y.max()) == 6.0 and float(y.min()) == 0.0


def test_conv_relu6_flag():
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 12, 10, generator=g) * 2
    w = torch.randn(96, 64, 1, 1, generator=g)
    b = torch.randn(96, generator=g)
    ref = F.relu6(F.conv2d(x, w, b))
    y = torch.empty(2, 12, 10, 96, device="cuda")
    p = _lib.ConvParams()
    xd, wd, bd = nhwc(x), w.permute(0, 2, 3, 1).contiguous().cuda(), b.cuda()
    p.x, p.w, p.bias, p.y = xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr()
    p.N, p.H_in, p.W_in, p.Cin, p.Cout, p.KH, p.KW, p.stride, p.pad = 2, 12, 10, 64, 96, 1, 1, 1, 0
    p.ldx, p.ldy, p.flags = 64, 96, _lib.CNL_RELU6
    _lib.check(lib.cnl_conv2d_nhwc_f32(ctypes.byref(p), None))
    torch.testing.assert_close(nchw(y), ref, rtol=TOL, atol=TOL)
    assert float(y.max()) == 6.0 and float(y.min()) == 0.0
    p.KH = p.KW = 3; p.pad = 1                                  # Winograd has no ReLU6 epilogue: it must refuse, not ignore
    assert lib.cnl_conv3x3_winograd_f32(ctypes.byref(p), None) == _lib.CNL_E_UNSUPPORTED


@pytest.mark.parametrize("shape,K,has_mask", [((2, 32, 9, 11), 3, True), ((1, 64, 16, 16), 3, False), ((1, 8, 5, 7), 1, True)])
def test_deform_sample_matches_torchvision_rule(shape, K, has_mask):
    """cnl_deform_sample_nhwc_f32 against the oracle's restatement of torchvision's deformable sampling (offsets up to +-2.5 px:
    taps leave the image on every side)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(K + shape[1])
    N, C, H, W = shape
    KK = K * K
    x = torch.randn(*shape, generator=g)
    off = (torch.rand(N, 2 * KK, H, W, generator=g) - 0.5) * 5
    off[:, :, 0, 0] = 0.0                                        # exact-integer positions too
    off[:, 0, 1, 1] = -1.0
    mlog = torch.randn(N, KK, H, W, generator=g)
    pad = (K - 1) // 2
    ys = torch.arange(H, dtype=torch.float32).view(1, H, 1)
    xs = torch.arange(W, dtype=torch.float32).view(1, 1, W)
    cols = []
    for k in range(KK):
        v = ref_cpu._bilinear_zero(x, ys - pad + k // K + off[:, 2 * k], xs - pad + k % K + off[:, 2 * k + 1])
        cols.append(v * torch.sigmoid(mlog[:, k]).unsqueeze(1) if has_mask else v)
    ref = torch.stack(cols, dim=1)                               # [N, KK, C, H, W]
    om = torch.cat([off, mlog], dim=1) if has_mask else off
    omd, xd = nhwc(om), nhwc(x)
    col = torch.full((N, H, W, KK * C), float("nan"), device="cuda")
    _lib.check(lib.cnl_deform_sample_nhwc_f32(xd.data_ptr(), omd.data_ptr(), col.data_ptr(), N, H, W, C, C, om.shape[1], K, int(has_mask), None))
    got = col.cpu().view(N, H, W, KK, C).permute(0, 3, 4, 1, 2)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
    assert lib.cnl_deform_sample_nhwc_f32(xd.data_ptr(), omd.data_ptr(), col.data_ptr(), N, H, W, C, C, om.shape[1], 4, 1, None) == _lib.CNL_E_UNSUPPORTED


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("three", [False, True])
def test_fuse_sum_kernel(mode, three):
    """cnl_fuse_sum_nhwc_f32 against the reference Fuse.forward's arithmetic (layers.py:160-175) written with torch ops: plain sum
    bit-exact, weighted sum within 2 ulp-ish (same operation order, division included)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(10 * mode + three)
    N, C, H, W = 2, 24, 12, 20
    a, b = torch.randn(N, C, H, W, generator=g), torch.randn(N, C, H, W, generator=g)
    lshape = {0: (H // 2, W // 2), 1: (H // 2, W // 2), 2: (2 * H, 2 * W), 3: (H, W)}[mode]
    last = torch.randn(N, C, *lshape, generator=g)
    res = {0: lambda t: F.interpolate(t, scale_factor=2, mode="nearest"), 1: lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False),
           2: lambda t: F.max_pool2d(t, 2, 2), 3: lambda t: t}[mode](last)
    ins = [a, b, res] if three else [a, res]
    y = torch.full((N, H, W, C), float("nan"), device="cuda")
    ad, bd, ld = nhwc(a), nhwc(b), nhwc(last)
    call = lambda g0, g1, gl, den: _lib.check(lib.cnl_fuse_sum_nhwc_f32(ad.data_ptr(), bd.data_ptr() if three else None, ld.data_ptr(), y.data_ptr(),
                                                                        N, H, W, C, C, C, C, C, g0, g1, gl, den, mode, None))
    call(1.0, 1.0, 1.0, 1.0)
    want = torch.stack(ins, dim=-1).sum(dim=-1)
    if mode == 1:                                               # ATen's CPU bilinear kernel orders the four-tap blend differently
        torch.testing.assert_close(nchw(y), want, rtol=1e-6, atol=1e-6)
    else:
        assert torch.equal(nchw(y), want)
    wts = torch.relu(torch.tensor([0.7, -0.2, 1.9] if three else [0.7, 1.9]))
    call(float(wts[0]), float(wts[1]) if three else 0.0, float(wts[-1]), float(wts.sum() + 1e-6))
    want = torch.sum(torch.stack([t * wts[j] for j, t in enumerate(ins)], dim=-1), dim=-1) / (torch.sum(wts) + 1e-6)
    torch.testing.assert_close(nchw(y), want, rtol=1e-6, atol=1e-6)
    if mode < 2:
        assert lib.cnl_fuse_sum_nhwc_f32(ad.data_ptr(), None, ld.data_ptr(), y.data_ptr(), N, 11, W, C, C, C, C, C, 1.0, 0.0, 1.0, 1.0, mode, None) == _lib.CNL_E_BAD_ARG
    assert lib.cnl_fuse_sum_nhwc_f32(ad.data_ptr(), None, ld.data_ptr(), y.data_ptr(), N, H, W, C, C, C, C, C, 1.0, 0.0, 1.0, 1.0, 4, None) == _lib.CNL_E_UNSUPPORTED
    assert lib.cnl_fuse_sum_nhwc_f32(ad.data_ptr(), None, ld.data_ptr(), y.data_ptr(), N, H, W, 22, C, C, C, C, 1.0, 0.0, 1.0, 1.0, mode, None) == _lib.CNL_E_UNSUPPORTED


NECKS = {
    "simple_deconv": {"name": "simple", "upsample_type": "conv_transpose", "deconv_kernel": 3},        # configs/test_config.yaml
    "simple_deconv4_separable": {"name": "simple", "upsample_type": "conv_transpose", "deconv_kernel": 4, "conv_type": "separable"},
    "simple_bilinear": {"name": "simple", "upsample_type": "bilinear"},
    "simple_nearest_separable": {"name": "simple", "upsample_type": "nearest", "conv_type": "separable"},
    "fpn_nearest_weighted": {"name": "fpn", "upsample_type": "nearest", "weighted_fusion": True},
    "fpn_bilinear_separable_weighted": {"name": "fpn", "upsample_type": "bilinear", "conv_type": "separable", "weighted_fusion": True},
    "fpn_deconv_weighted": {"name": "fpn", "upsample_type": "conv_transpose", "weighted_fusion": True},
    "fpn_deconv2_separable": {"name": "fpn", "upsample_type": "conv_transpose", "deconv_kernel": 2, "conv_type": "separable"},
    "fpn_bilinear": {"name": "fpn", "upsample_type": "bilinear"},
    "fpn_deformable": {"name": "fpn", "upsample_type": "nearest", "conv_type": "deformable"},                       # DCNv2 (mask)
    "simple_deformable_v1_bilinear": {"name": "simple", "upsample_type": "bilinear", "conv_type": "deformable", "version": 1},
    "simple_deformable_nearest": {"name": "simple", "upsample_type": "nearest", "conv_type": "deformable"},
    # IDA / BiFPN (docs/implementation.md:42-43): defined on the reference's Fuse node, params.IDANeck / params.BiFPNNeck
    "ida_nearest": {"name": "ida"},
    "ida_bilinear_weighted": {"name": "ida", "upsample_type": "bilinear", "weighted_fusion": True},
    "ida_deconv_separable": {"name": "ida", "upsample_type": "conv_transpose", "conv_type": "separable"},
    "bifpn_nearest": {"name": "bifpn", "num_channels": 64, "num_layers": 2},
    "bifpn_weighted_3layers": {"name": "bifpn", "num_channels": 32, "num_layers": 3, "weighted_fusion": True},
    "bifpn_bilinear_separable_weighted": {"name": "bifpn", "num_channels": 64, "num_layers": 2, "upsample_type": "bilinear", "conv_type": "separable",
                                          "weighted_fusion": True},
    "bifpn_deconv_1layer": {"name": "bifpn", "num_channels": 96, "num_layers": 1, "upsample_type": "conv_transpose"},
}


@pytest.mark.parametrize("name", sorted(NECKS))
def test_model_with_neck_option_matches_cpu_oracle(name):
    neck = dict(NECKS[name])
    if neck["name"] in ("simple", "fpn"):
        neck["upsample_channels"] = [256, 128, 64]
    ups = neck.get("upsample_type", "nearest")
    new = neck["name"] in ("ida", "bifpn")
    torch.manual_seed(0)
    model = cl.build_centernet({"task": "detection", "backbone": {"name": "resnet34", "pretrained": False}, "neck": neck,
                                "output_heads": {"heatmap": {"num_classes": 5, "init_bias": -2.19}, "box_2d": {"init_bias": 10}}})
    sd = ref_cpu.synth_state_dict(model.state_dict(), seed=1, calib_shape=(2, 3, 128, 128), upsample_type=ups)
    if "weighted" in name:
        assert any(k.endswith(".weights") for k in sd)
        first = sorted(k for k in sd if k.endswith(".weights"))[1]
        sd[first][0] = -0.3                                     # relu(weights): this node ignores its first input
        for i, k in enumerate(sorted(k for k in sd if k.endswith(".weights"))[2:]):
            sd[k] = torch.rand(sd[k].shape, generator=torch.Generator().manual_seed(i)) + 0.25
    model.load_state_dict(sd)
    model = model.cuda()
    x = recipes.images(4321, (2, 3, 96, 128))
    ref, feats, neck_ref = ref_cpu.forward(sd, x, sigmoid=False, return_intermediates=True, upsample_type=ups)
    enc = model.get_encoded_outputs(x.cuda())
    assert float(neck_ref.abs().mean()) > 1e-2
    for k, r in ref.items():
        assert tuple(enc[k].shape) == tuple(r.shape) == (2, r.shape[1], 24, 32)
        torch.testing.assert_close(enc[k].cpu(), r, rtol=TOL, atol=TOL)
    what = [L.what for L in next(iter(model._engine.plans.values())).launches]
    if new:
        nodes = {"ida": 6, "bifpn": 3 * neck.get("num_layers", 3) + 3 * (neck.get("num_layers", 3) - 1)}[neck["name"]]
        assert sum(w.startswith("neck.") and ".output_conv" in w and not w.endswith((".dw", ".absmax")) for w in what) == nodes
        if name == "ida_nearest":                               # project -> upsample -> sum inside one 1x1 conv per node
            assert sum("project+up+sum" in w for w in what) == 6 and not any(".sum (" in w for w in what)
        if name == "bifpn_nearest":
            assert sum("max-pool down" in w for w in what) == 3
        if ups == "conv_transpose":
            assert sum(".resize (conv_transpose)" in w for w in what) == (6 if neck["name"] == "ida" else 3)
        return
    if ups == "conv_transpose":
        assert sum("conv_transpose" in w for w in what) == 3
    if ups == "bilinear":
        assert sum("bilinear" in w for w in what) == 3
    if neck.get("conv_type") == "separable":
        assert sum(w.endswith(".dw") for w in what) == 3
    if neck.get("conv_type") == "deformable":
        assert sum(w.endswith(".sample") for w in what) == 3 and sum("deform_conv (GEMM)" in w for w in what) == 3
