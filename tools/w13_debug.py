#!/usr/bin/env python
"""Debug aid for winograd13.hip: where (row, pixel phase, tile, cout piece) do the outputs of one small launch differ from float64?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from test_gpu_conv import run_winograd, ref_conv, mk  # noqa: E402
from centernet_lightning_amd._lib import CNL_ALGO_FORCE  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 13
for (N, Cin, H, W, Cout) in [(1, 32, 4, 128, 64), (1, 32, 4, 16, 64), (1, 32, 8, 64, 64), (1, 64, 4, 128, 64), (1, 128, 4, 128, 64), (2, 32, 5, 7, 4)]:
    x, w, b = mk(N, Cin, H, W, Cout, 3, seed=1)
    ref = ref_conv(x.double(), w.double(), b.double(), 1, 0, None)
    out = run_winograd(x, w, b, 0, None, algo=CNL_ALGO_FORCE + V)
    err = (out.double() - ref).abs()          # [N, Cout, H, W]
    sc = ref.abs().max().item()
    print(f"--- N{N} c{Cin} {H}x{W} o{Cout}: max err {err.max().item():.3e} scale {sc:.2f} nan {bool(torch.isnan(out).any())}")
    bad = err > 1e-4 * sc
    print("  bad fraction", bad.float().mean().item())
    print("  by row      ", [round(bad[:, :, r].float().mean().item(), 3) for r in range(H)])
    print("  by x % 4    ", [round(bad[:, :, :, p::4].float().mean().item(), 3) for p in range(4)])
    nt = (W + 3) // 4
    print("  by tile     ", [round(bad[:, :, :, 4 * t:4 * t + 4].float().mean().item(), 2) for t in range(nt)])
    print("  by cout / 4 ", [round(bad[:, 4 * c:4 * c + 4].float().mean().item(), 2) for c in range((Cout + 3) // 4)])
    print("  max by cout/4", ["%.1e" % err[:, 4 * c:4 * c + 4].max().item() for c in range((Cout + 3) // 4)])
x, w, b = mk(1, 32, 4, 128, 64, 3, seed=1)
ref = ref_conv(x.double(), w.double(), b.double(), 1, 0, None)
outs = [run_winograd(x, w, b, 0, None, algo=CNL_ALGO_FORCE + V) for _ in range(3)]
print("deterministic:", torch.equal(outs[0], outs[1]), torch.equal(outs[1], outs[2]))
o = outs[0]
torch.set_printoptions(precision=4, linewidth=200, sci_mode=True)
print("out [cout 16..19, row 1, x 4..8]\n", o[0, 16:20, 1, 4:9])
print("ref\n", ref[0, 16:20, 1, 4:9])
bad = ((o.double() - ref).abs() > 1e-3)
idx = bad.nonzero()
print("first bad idx", idx[:20].tolist())
print("bad values", [float(o[tuple(i)]) for i in idx[:20]])
print("bad values as int bits", [hex(o[tuple(i)].view(torch.int32).item() & 0xffffffff) for i in idx[:20]])
