"""ORACLE — TEST INFRASTRUCTURE ONLY.  Generates tests/golden/layers_*.npz by RUNNING THE REFERENCE'S OWN neck primitives
(centernet_lightning/models/layers.py: make_conv :40-79, make_upsample :81-101, _init_bilinear_upsampling :103-116, Fuse
:138-177) in the build container, to pin oracle/ref_cpu.py's make_conv_forward / make_upsample_forward / fuse_forward.

Run:  python oracle/make_golden_layers.py     (only where /root/reference exists)
Fixtures are data only: module state_dict + input tensors + the reference's output.
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_cpu                                                   # noqa: E402
from _ref_import import import_reference_centernet               # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def randomize(mod, g):
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
        elif isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            m.weight.data.copy_(torch.randn(m.weight.shape, generator=g) * 0.2)
            if m.bias is not None:
                m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    return mod.eval()


def main():
    os.makedirs(OUT, exist_ok=True)
    import_reference_centernet()
    layers = importlib.import_module("centernet_lightning.models.layers")
    g = torch.Generator().manual_seed(21)
    payload = {}

    # ---- Fuse, every option of the config surface ----
    fuse_cases = [
        # name, in_channels [skip, top], out, upsample, conv_type, weighted, fusion weights
        ("nearest_normal", [16, 32], 16, "nearest", "normal", False, None),
        ("nearest_noproj", [16, 16], 16, "nearest", "normal", False, None),
        ("bilinear_separable_weighted", [24, 16], 16, "bilinear", "separable", True, [0.7, 1.6]),
        ("deconv_normal_weighted", [16, 32], 16, "conv_transpose", "normal", True, [1.3, 0.4]),
        ("deconv_separable", [8, 8], 8, "conv_transpose", "separable", False, None),
        ("bilinear_normal_negweight", [16, 16], 16, "bilinear", "normal", True, [-0.5, 0.9]),     # relu(weights) zeroes input 1
    ]
    names = []
    with torch.no_grad():
        for name, inc, out, ups, ct, wf, wts in fuse_cases:
            m = layers.Fuse(inc, out, "up", upsample=ups, conv_type=ct, weighted_fusion=wf)
            randomize(m, g)
            if wts is not None:
                m.weights.data.copy_(torch.tensor(wts))
            skip = torch.randn(2, inc[0], 10, 12, generator=g)
            top = torch.randn(2, inc[1], 5, 6, generator=g)
            y = m(skip, top)
            sd = {k: v.clone() for k, v in m.state_dict().items()}
            o = ref_cpu.fuse_forward({"f." + k: v for k, v in sd.items()}, "f.", skip, top, upsample_type=ups)
            assert torch.equal(o, y) or torch.allclose(o, y, rtol=0, atol=1e-6), (name, float((o - y).abs().max()))
            print(f"fuse_{name}: out {tuple(y.shape)}  |oracle - ref|max = {float((o - y).abs().max()):.2e}")
            payload[f"fuse.{name}.skip"] = skip.numpy()
            payload[f"fuse.{name}.top"] = top.numpy()
            payload[f"fuse.{name}.out"] = y.numpy()
            payload[f"fuse.{name}.upsample"] = np.array(ups)
            for k, v in sd.items():
                payload[f"fuse.{name}.sd.{k}"] = v.numpy()
            names.append(name)
        payload["fuse_cases"] = np.array(names)

        # ---- make_upsample(conv_transpose) for every supported kernel, with the bilinear init as written ----
        for k in (2, 3, 4):
            torch.manual_seed(100 + k)
            up = layers.make_upsample("conv_transpose", deconv_channels=8, deconv_kernel=k, deconv_init_bilinear=True).eval()
            payload[f"deconv.k{k}.init_w"] = up[0].weight.detach().numpy().copy()      # what _init_bilinear_upsampling left
            randomize(up, g)
            x = torch.randn(1, 8, 5, 7, generator=g)
            y = up(x)
            sd = {"u." + kk: v for kk, v in up.state_dict().items()}
            o = ref_cpu.make_upsample_forward(x, sd, "u", "conv_transpose")
            assert torch.allclose(o, y, rtol=0, atol=1e-6), k
            payload[f"deconv.k{k}.x"] = x.numpy()
            payload[f"deconv.k{k}.out"] = y.numpy()
            for kk, v in up.state_dict().items():
                payload[f"deconv.k{k}.sd.{kk}"] = v.numpy()
            print(f"deconv k={k}: {tuple(x.shape)} -> {tuple(y.shape)}")

        # ---- make_conv(separable) ----
        mc = randomize(layers.make_conv(12, 20, conv_type="separable"), g)
        x = torch.randn(2, 12, 9, 11, generator=g) * 40         # large enough that ReLU6 clips
        y = mc(x)
        o = ref_cpu.make_conv_forward(x, {"c." + kk: v for kk, v in mc.state_dict().items()}, "c")
        assert torch.allclose(o, y, rtol=0, atol=1e-6)
        assert float(y.max()) == 6.0
        payload["sepconv.x"] = x.numpy()
        payload["sepconv.out"] = y.numpy()
        for kk, v in mc.state_dict().items():
            payload[f"sepconv.sd.{kk}"] = v.numpy()
    payload["torch"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(OUT, "layers_neck_options.npz"), **payload)
    print("saved", os.path.getsize(os.path.join(OUT, "layers_neck_options.npz")), "bytes")


if __name__ == "__main__":
    main()
