"""CPU: oracle/ref_cpu.py's neck primitives (make_conv / make_upsample / Fuse restatements) against golden vectors produced by
the reference's own models/layers.py (oracle/make_golden_layers.py), and the product's parameter containers / config surface
for the same options."""
import os

import numpy as np
import pytest
import torch

import ref_cpu
import centernet_lightning_amd as cl
from centernet_lightning_amd import params as P

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "layers_neck_options.npz")


@pytest.fixture(scope="module")
def g():
    return dict(np.load(GOLDEN))


def _sd(g, prefix, new_prefix):
    return {new_prefix + k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def test_fuse_restatement_matches_reference(g):
    for name in g["fuse_cases"]:
        sd = _sd(g, f"fuse.{name}.sd.", "f.")
        out = ref_cpu.fuse_forward(sd, "f.", torch.from_numpy(g[f"fuse.{name}.skip"]), torch.from_numpy(g[f"fuse.{name}.top"]),
                                   upsample_type=str(g[f"fuse.{name}.upsample"]))
        torch.testing.assert_close(out, torch.from_numpy(g[f"fuse.{name}.out"]), rtol=0, atol=1e-6)


def test_deconv_and_separable_restatements_match_reference(g):
    for k in (2, 3, 4):
        sd = _sd(g, f"deconv.k{k}.sd.", "u.")
        out = ref_cpu.make_upsample_forward(torch.from_numpy(g[f"deconv.k{k}.x"]), sd, "u", "conv_transpose")
        torch.testing.assert_close(out, torch.from_numpy(g[f"deconv.k{k}.out"]), rtol=0, atol=1e-6)
    out = ref_cpu.make_conv_forward(torch.from_numpy(g["sepconv.x"]), _sd(g, "sepconv.sd.", "c."), "c")
    torch.testing.assert_close(out, torch.from_numpy(g["sepconv.out"]), rtol=0, atol=1e-6)
    assert float(out.max()) == 6.0 and float(out.min()) == 0.0            # ReLU6 clips on both sides


def test_param_containers_mirror_reference_modules(g):
    """The product's FuseParams / DeconvBn / SeparableConvBn accept the reference modules' state_dicts key for key."""
    cases = {"nearest_normal": ([16, 32], 16, "nearest", "normal", False),
             "nearest_noproj": ([16, 16], 16, "nearest", "normal", False),
             "bilinear_separable_weighted": ([24, 16], 16, "bilinear", "separable", True),
             "deconv_normal_weighted": ([16, 32], 16, "conv_transpose", "normal", True),
             "deconv_separable": ([8, 8], 8, "conv_transpose", "separable", False)}
    for name, (inc, out, ups, ct, wf) in cases.items():
        m = P.FuseParams(inc[0], inc[1], out, ups, ct, wf)
        sd = _sd(g, f"fuse.{name}.sd.", "")
        assert set(m.state_dict().keys()) == set(sd.keys()), name
        m.load_state_dict(sd, strict=True)
    for k in (2, 3, 4):                                                   # _init_bilinear_upsampling (layers.py:103-116) as written
        torch.manual_seed(100 + k)
        d = P.DeconvBn(8, k, init_bilinear=True)
        w_ref = g[f"deconv.k{k}.init_w"]
        assert d.deconv.weight.shape == w_ref.shape and d.deconv.padding == ((k + k % 2) // 2 - 1,) * 2
        np.testing.assert_array_equal(d.deconv.weight.detach().numpy()[:, 0], w_ref[:, 0])
        np.testing.assert_array_equal(d.deconv.weight.detach().numpy(), w_ref)      # same RNG stream for the untouched entries
    s = P.SeparableConvBn(12, 20)
    assert set(s.state_dict().keys()) == set(_sd(g, "sepconv.sd.", "").keys())


def _cfg(neck):
    return {"task": "detection", "backbone": {"name": "resnet34", "pretrained": False}, "neck": neck,
            "output_heads": {"heatmap": {"num_classes": 3, "init_bias": -2.19}, "box_2d": {"init_bias": 10}}}


def test_config_surface_neck_options():
    # configs/test_config.yaml:8-18 of the reference: simple neck with transposed-conv upsampling
    m = cl.build_centernet(_cfg({"name": "simple", "upsample_channels": [256, 128, 64], "upsample_type": "conv_transpose",
                                 "conv_type": "normal", "deconv_kernel": 3, "deconv_init_bilinear": True}))
    assert "neck.upsamples.0.0.weight" in m.state_dict() and m.output_stride == 4
    m = cl.build_centernet(_cfg({"name": "fpn", "upsample_channels": [256, 128, 64], "upsample_type": "bilinear",
                                 "conv_type": "separable", "weighted_fusion": True}))
    sd = m.state_dict()
    assert "neck.fuse.0.weights" in sd and "neck.fuse.2.output_conv.3.weight" in sd and sd["neck.fuse.1.output_conv.0.weight"].shape == (128, 1, 3, 3)
    for bad in ({"name": "fpn", "conv_type": "dilated"}, {"name": "simple", "upsample_type": "cubic"},
                {"name": "fpn", "conv_type": "deformable", "mask_activation": "Hardsigmoid"}, {"name": "fpn", "conv_type": "deformable", "version": 3},
                {"name": "simple", "upsample_type": "conv_transpose", "deconv_kernel": 5}, {"name": "panet"},
                {"name": "bifpn", "num_layers": 0}, {"name": "bifpn", "num_channels": 30}, {"name": "ida", "upsample_type": "cubic"}):
        with pytest.raises(ValueError):
            cl.build_centernet(_cfg(bad))
    # the CPU oracle runs every option (shape contract of tests/test_necks.py:27-28,44-45: stride 32 -> 4, C = upsample_channels[-1])
    for neck, ups in (({"name": "simple", "upsample_type": "conv_transpose", "conv_type": "separable"}, "conv_transpose"),
                      ({"name": "fpn", "upsample_type": "bilinear", "weighted_fusion": True}, "bilinear")):
        m = cl.build_centernet(_cfg(neck))
        x = torch.rand(1, 3, 64, 96)
        out, feats, nk = ref_cpu.forward(m.state_dict(), x, return_intermediates=True, upsample_type=ups)
        assert tuple(nk.shape) == (1, 64, 16, 24) and tuple(out["heatmap"].shape) == (1, 3, 16, 24)


def test_fuse_down_and_three_inputs_match_reference_fuse():
    """The node types BiFPN's bottom-up path is made of — resize="down" (always MaxPool2d(2, 2): the `downsample=` argument never reaches
    make_downsample, layers.py:118/:156) and three inputs — against outputs of the reference's own Fuse (oracle/make_golden_fuse_down.py);
    and the product's FuseNode takes the reference module's state_dict verbatim."""
    g = dict(np.load(os.path.join(os.path.dirname(GOLDEN), "layers_fuse_down.npz")))
    assert len(g["cases"]) == 5
    for name in [str(c) for c in g["cases"]]:
        sd = {"f." + k: v for k, v in _sd(g, f"{name}.sd.", "").items()}
        xs = [torch.from_numpy(g[f"{name}.in{j}"]) for j in range(int(g[f"{name}.n_in"]))]
        y = ref_cpu.fuse_forward_n(sd, "f.", xs, str(g[f"{name}.resize"]))
        np.testing.assert_array_equal(y.numpy(), g[f"{name}.out"])
        in_ch = [int(x.shape[1]) for x in xs]
        node = P.FuseNode(in_ch, int(g[f"{name}.out"].shape[1]), str(g[f"{name}.resize"]), weighted_fusion="f.weights" in sd,
                          conv_type="separable" if "f.output_conv.3.weight" in sd else "normal")
        node.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)


def test_ida_and_bifpn_necks_surface():
    """neck.name = ida | bifpn (README.md:70, docs/implementation.md:42-43 of the reference; the classes themselves are missing there):
    shape contract of the reference's neck tests (stride 32 -> 4), key layout, what the CPU oracle computes."""
    for neck, n_out, keys in (({"name": "ida"}, 64, ("neck.stages.0.2.project.1.weight", "neck.stages.2.0.output_conv.1.running_var")),
                              ({"name": "bifpn", "num_channels": 32, "num_layers": 2, "weighted_fusion": True}, 32,
                               ("neck.bifpn.0.td.2.project.1.weight", "neck.bifpn.0.bu.0.weights", "neck.bifpn.0.bu.2.output_conv.0.weight",
                                "neck.bifpn.1.td.0.weights")),
                              ({"name": "ida", "upsample_type": "conv_transpose", "conv_type": "separable"}, 64, ("neck.stages.1.1.resize.0.weight",))):
        m = cl.build_centernet(_cfg(neck))
        sd = m.state_dict()
        assert all(k in sd for k in keys), [k for k in keys if k not in sd]
        assert m.output_stride == 4 and m.neck.out_channels == n_out
        out, feats, nk = ref_cpu.forward(sd, torch.rand(1, 3, 64, 96), return_intermediates=True, upsample_type=neck.get("upsample_type", "nearest"))
        assert tuple(nk.shape) == (1, n_out, 16, 24) and tuple(out["box_2d"].shape) == (1, 4, 16, 24)
    sd = cl.build_centernet(_cfg({"name": "bifpn", "num_layers": 2})).state_dict()
    assert "neck.bifpn.1.bu.0.output_conv.0.weight" not in sd          # the last layer's bottom-up nodes feed nothing
    assert sd["neck.bifpn.0.bu.2.project.0.weight"].shape == (64, 512, 1, 1) and "neck.bifpn.1.td.2.project.0.weight" not in sd
    # IDA == nested Fuse calls written out by hand on one stage
    m = cl.build_centernet(_cfg({"name": "ida"}))
    sd = m.state_dict()
    feats = ref_cpu.backbone_features(sd, torch.rand(1, 3, 64, 64))
    lv = list(feats[1:])
    s0 = [ref_cpu.fuse_forward_n(sd, f"neck.stages.0.{i}.", [lv[i], lv[i + 1]]) for i in range(3)]
    s1 = [ref_cpu.fuse_forward_n(sd, f"neck.stages.1.{i}.", [s0[i], s0[i + 1]]) for i in range(2)]
    want = ref_cpu.fuse_forward_n(sd, "neck.stages.2.0.", [s1[0], s1[1]])
    torch.testing.assert_close(ref_cpu.neck_forward(sd, feats), want, rtol=0, atol=0)


def test_deformable_conv_params_and_oracle():
    """conv_type="deformable" (layers.py:9-38, 47-54): key names / initialisation of the reference module, and the oracle's
    restatement of torchvision's deform_conv2d (absent from the image: "parity unpinned") on cases with a known answer."""
    m = P.DeformableConvBn(16, 24)
    keys = set(m.state_dict().keys())
    assert {"0.offset_conv.weight", "0.offset_conv.bias", "0.mask_conv.0.weight", "0.mask_conv.0.bias", "0.deform_conv.weight",
            "1.weight", "1.bias", "1.running_mean", "1.running_var"} <= keys
    assert m.block.offset_conv.weight.shape == (18, 16, 3, 3) and m.block.mask_conv[0].weight.shape == (9, 16, 3, 3)
    assert m.block.deform_conv.weight.shape == (24, 16, 3, 3) and "0.deform_conv.bias" not in keys
    assert float(m.block.offset_conv.weight.abs().sum()) == 0 and float(m.block.mask_conv[0].bias.abs().sum()) == 0     # layers.py:27-31
    assert "0.mask_conv.0.weight" not in P.DeformableConvBn(8, 8, version=1).state_dict()
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(2, 5, 7, 9, generator=g), torch.randn(4, 5, 3, 3, generator=g)
    F = torch.nn.functional
    # zero offsets, unit mask == the plain convolution; at initialisation (sigmoid(0) = 0.5 mask) it is half of it
    torch.testing.assert_close(ref_cpu.deform_conv2d(x, torch.zeros(2, 18, 7, 9), w, torch.ones(2, 9, 7, 9)), F.conv2d(x, w, padding=1), rtol=0, atol=2e-5)
    # dy = +1 on every tap == correlation with rows y .. y+2 (zeros beyond the image)
    off = torch.zeros(2, 18, 7, 9); off[:, 0::2] = 1.0
    torch.testing.assert_close(ref_cpu.deform_conv2d(x, off, w), F.conv2d(F.pad(x, (1, 1, 0, 2)), w), rtol=0, atol=2e-5)
    # dx = +0.5 == the mean of horizontal neighbours (zero padding)
    off = torch.zeros(2, 18, 7, 9); off[:, 1::2] = 0.5
    xp = F.pad(x, (1, 2, 1, 1))
    torch.testing.assert_close(ref_cpu.deform_conv2d(x, off, w), F.conv2d(0.5 * (xp[..., :-1] + xp[..., 1:]), w), rtol=0, atol=2e-5)
    # a tap thrown far outside the image contributes nothing
    off = torch.zeros(2, 18, 7, 9); off[:, 8] = 100.0            # tap k = 4 (the centre): dy = +100
    w_c = w.clone(); w_c[:, :, 1, 1] = 0
    torch.testing.assert_close(ref_cpu.deform_conv2d(x, off, w), F.conv2d(x, w_c, padding=1), rtol=0, atol=2e-5)
