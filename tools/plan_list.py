#!/usr/bin/env python
"""Every launch of a configuration's plan, in order (name, C-ABI entry point, Winograd variant where it applies):
python tools/plan_list.py [--config simple|fpn|tracking] [--batch N] [--size H W]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="simple")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, nargs=2, default=[512, 512])
    a = ap.parse_args()
    model = bench.build_model(a.config)
    x = torch.rand(a.batch, 3, *a.size, device="cuda")
    with torch.no_grad():
        model(x)
    plan = model._engine.plan_for(x)
    lib = plan.lib
    for i, L in enumerate(plan.launches):
        name = L.fn if isinstance(L.fn, str) else getattr(L.fn, "__name__", str(L.fn))
        v = ""
        if L.fn is lib.cnl_conv3x3_winograd_f32:
            v = f"variant {lib.cnl_conv3x3_winograd_variant(ctypes.byref(L.args))}"
        print(f"{i:3d} {L.what:60s} {name:32s} {v}")


if __name__ == "__main__":
    main()
