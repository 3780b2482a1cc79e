#!/usr/bin/env python
"""Per-launch table of one forward's plan (HIP events around every launch, whole plan replayed in order):
python tools/plan_profile.py [--config simple|fpn|tracking] [--batch N] [--size H W] [--algo auto|f32] [--latency] [--opt field=0|1 ...]"""
import argparse
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
    ap.add_argument("--algo", default="auto")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--split-small", action="store_true")
    ap.add_argument("--latency", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="KernelOptions field=0|1 (e.g. --opt reverse_out_conv=0)")
    a = ap.parse_args()
    extra = {kv.split("=")[0]: bool(int(kv.split("=")[1])) for kv in a.opt}
    model = bench.build_model(a.config, algo=a.algo, split_small=a.split_small, latency=a.latency, **extra)
    x = torch.rand(a.batch, 3, *a.size, device="cuda")
    rows, plan = bench.conv_kernel_profile(model, x, reps=a.reps)
    tot = sum(r[2] for r in rows)
    print(f"{'launch':58s} {'kind':15s} {'us':>9s} {'TFLOP/s':>8s} {'GB/s alg':>9s} {'%':>5s}")
    for what, fl, ms, kd, nb, _fx in rows:
        print(f"{what[:58]:58s} {kd:15s} {ms * 1e3:9.1f} {fl / ms / 1e9 if ms else 0:8.1f} {nb / ms / 1e6 if ms else 0:9.0f} {100 * ms / tot:5.1f}")
    print(f"conv launches {len(rows)}: {tot:.3f} ms; all launches in the plan: {len(plan.launches)}")


if __name__ == "__main__":
    main()
