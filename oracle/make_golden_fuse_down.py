"""ORACLE — TEST INFRASTRUCTURE ONLY.  Generates tests/golden/layers_fuse_down.npz by RUNNING THE REFERENCE'S OWN Fuse node
(centernet_lightning/models/layers.py:138-177) with resize="down" and with three inputs — the node types a BiFPN bottom-up path is
made of — to pin oracle/ref_cpu.fuse_forward_n.  Run:  python oracle/make_golden_fuse_down.py   (only where /root/reference exists).
Fixtures are data only: module state_dict + input tensors + the reference's output."""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_cpu                                                   # noqa: E402
from _ref_import import import_reference_centernet               # noqa: E402
from make_golden_layers import randomize                         # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    import_reference_centernet()
    layers = importlib.import_module("centernet_lightning.models.layers")
    g = torch.Generator().manual_seed(33)
    payload, names = {}, []
    cases = [
        # name, in_channels, out, resize, downsample arg, conv_type, weighted, weights
        ("down2_plain", [16, 16], 16, "down", "max", "normal", False, None),
        ("down3_plain", [16, 16, 16], 16, "down", "max", "normal", False, None),
        ("down3_project_weighted", [24, 16, 8], 16, "down", "max", "normal", True, [0.8, 1.7, 0.3]),
        ("down3_average_arg_is_ignored", [16, 16, 16], 16, "down", "average", "separable", True, [1.0, -0.4, 2.0]),
        ("up3_weighted", [16, 32, 16], 16, "up", "max", "normal", True, [0.5, 0.9, 1.4]),
    ]
    with torch.no_grad():
        for name, inc, out, resize, down, ct, wf, wts in cases:
            m = layers.Fuse(inc, out, resize, downsample=down, conv_type=ct, weighted_fusion=wf)
            randomize(m, g)
            if wts is not None:
                m.weights.data.copy_(torch.tensor(wts))
            h, w = 8, 10
            xs = [torch.randn(2, c, h, w, generator=g) for c in inc[:-1]]
            xs.append(torch.randn(2, inc[-1], h * 2, w * 2, generator=g) if resize == "down" else torch.randn(2, inc[-1], h // 2, w // 2, generator=g))
            y = m(*xs)
            sd = {k: v.clone() for k, v in m.state_dict().items()}
            o = ref_cpu.fuse_forward_n({"f." + k: v for k, v in sd.items()}, "f.", xs, resize)
            assert torch.allclose(o, y, rtol=0, atol=1e-6), (name, float((o - y).abs().max()))
            print(f"{name}: out {tuple(y.shape)}  |oracle - ref|max = {float((o - y).abs().max()):.2e}")
            for j, x in enumerate(xs):
                payload[f"{name}.in{j}"] = x.numpy()
            payload[f"{name}.out"] = y.numpy()
            payload[f"{name}.resize"] = np.array(resize)
            payload[f"{name}.n_in"] = np.array(len(xs))
            for k, v in sd.items():
                payload[f"{name}.sd.{k}"] = v.numpy()
            names.append(name)
    payload["cases"] = np.array(names)
    payload["torch"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(OUT, "layers_fuse_down.npz"), **payload)
    print("saved", os.path.getsize(os.path.join(OUT, "layers_fuse_down.npz")), "bytes")


if __name__ == "__main__":
    main()
