#!/usr/bin/env python
"""Quick GPU check of one Winograd variant (default 9: row-Winograd) against torch float64 conv2d on a set of shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from test_gpu_conv import run_winograd, ref_conv, mk  # noqa: E402
from centernet_lightning_amd._lib import CNL_ALGO_FORCE, CNL_RELU, CNL_UPSAMPLE_IN  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 9
CASES = [  # N, Cin, H, W, Cout, flags, residual
    (1, 32, 8, 64, 64, 0, False),
    (1, 32, 8, 64, 64, CNL_RELU, True),
    (2, 64, 16, 128, 64, CNL_RELU, False),
    (1, 256, 32, 32, 256, CNL_RELU, True),
    (2, 64, 19, 34, 96, CNL_RELU, True),
    (1, 128, 40, 24, 128, 0, True),
    (3, 32, 5, 7, 4, 0, False),
    (1, 64, 9, 130, 72, CNL_RELU, False),
    (2, 32, 6, 10, 64, CNL_RELU | CNL_UPSAMPLE_IN, False),
    (1, 512, 16, 16, 512, CNL_RELU, True),
    (5, 64, 32, 32, 64, CNL_RELU, True),
    (3, 32, 9, 32, 96, 0, False),
    (7, 64, 16, 16, 128, CNL_RELU, True),
    (2, 32, 5, 16, 64, 0, False),
]
bad = 0
for (N, Cin, H, W, Cout, flags, use_res) in CASES:
    x, w, b = mk(N, Cin, H, W, Cout, 3, seed=Cin + Cout + H + W)
    up = 2 if flags & CNL_UPSAMPLE_IN else 1
    res = torch.randn(N, Cout, H * up, W * up, generator=torch.Generator().manual_seed(6)) if use_res else None
    ref = ref_conv(x.double(), w.double(), b.double(), 1, flags, res.double() if use_res else None)
    out, ym = run_winograd(x, w, b, flags, res, algo=CNL_ALGO_FORCE + V, ymax=True)
    o2 = run_winograd(x, w, b, flags, res, algo=CNL_ALGO_FORCE + 2)
    e9 = (out.double() - ref).abs().max().item()
    e2 = (o2.double() - ref).abs().max().item()
    sc = ref.abs().max().item()
    ok_m = torch.equal(ym, out.abs().amax(dim=(1, 2, 3)))
    nan = bool(torch.isnan(out).any())
    flag = "" if (e9 <= 1.25 * e2 + 1e-7 * sc and ok_m and not nan) else "   <<<<<< BAD"
    bad += bool(flag)
    print(f"N{N} c{Cin} {H}x{W} o{Cout} f{flags} r{int(use_res)}: err v{V} {e9:.3e}  fp32mfma {e2:.3e}  scale {sc:.2f} ymax_ok {ok_m} nan {nan}{flag}", flush=True)
g = torch.Generator().manual_seed(3)
x = torch.randint(-3, 4, (2, 64, 20, 70), generator=g).float()
w = torch.randint(-2, 3, (96, 64, 3, 3), generator=g).float() * 4
b = torch.randint(-5, 6, (96,), generator=g).float()
eq = torch.equal(run_winograd(x, w, b, 0, algo=CNL_ALGO_FORCE + V), ref_conv(x, w, b, 1, 0))
print("exact on integers:", eq)
# batch invariance, images side by side
g = torch.Generator().manual_seed(18)
for (n_, w_) in ((5, 32), (6, 16)):
    x = torch.randn(n_, 64, 16, w_, generator=g).clamp_min(0) * torch.pow(10.0, torch.randint(-3, 3, (n_, 1, 1, 1), generator=g).float())
    w = torch.randn(64, 64, 3, 3, generator=g) * (2.0 / (64 * 9)) ** 0.5
    b = torch.randn(64, generator=g)
    full = run_winograd(x, w, b, CNL_RELU, algo=CNL_ALGO_FORCE + V)
    ok_ = all(torch.equal(full[i:i + 1], run_winograd(x[i:i + 1], w, b, CNL_RELU, algo=CNL_ALGO_FORCE + V)) for i in range(n_))
    print("side by side W", w_, "batch invariant:", ok_)
    bad += not ok_
# batch invariance
g = torch.Generator().manual_seed(17)
x = torch.randn(3, 64, 24, 80, generator=g).clamp_min(0) * torch.tensor([1.0, 1e-3, 300.0]).view(3, 1, 1, 1)
w = torch.randn(128, 64, 3, 3, generator=g) * (2.0 / (64 * 9)) ** 0.5
b = torch.randn(128, generator=g)
full = run_winograd(x, w, b, CNL_RELU, algo=CNL_ALGO_FORCE + V)
inv = all(torch.equal(full[i:i + 1], run_winograd(x[i:i + 1], w, b, CNL_RELU, algo=CNL_ALGO_FORCE + V)) for i in range(3))
print("batch invariant:", inv)
sys.exit(1 if (bad or not eq or not inv) else 0)
