#!/usr/bin/env python
"""Phase timing of stem_f16x2_kernel per workgroup (s_memrealtime stamps, 10 ns) from the S5_TRACE build:
`make -C centernet-lightning_amd/csrc variant TAG=s5trace EXTRA=-DS5_TRACE`, then `python tools/stem_trace.py` (S5LIB = library under tools/ablibs).
Prints the phase durations (median over workgroups), how the workgroups of one CU overlap, and the timeline of one CU."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CENTERNET_GFX950_LIB"] = os.path.join(ROOT, "tools/ablibs/" + os.environ.get("S5LIB", "libcnl_s5trace.so"))
sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
import numpy as np
import torch
from centernet_lightning_amd import _lib
lib = _lib.load()
N = 32
x = torch.rand(N, 3, 512, 512, device="cuda")
w = torch.randn(64, 7, 7, 3, device="cuda") * 0.1
wp = torch.empty(lib.cnl_stem_packed_weight_floats(), device="cuda")
b = torch.zeros(64, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.cnl_stem_pack_weights_f32(w.data_ptr(), wp.data_ptr(), st)
sn, sc, sh, sw = x.stride()
yp = torch.empty(N, 128, 128, 64, device="cuda")
ym = _lib.absmax_buffer(N)
g = lambda: lib.cnl_stem_conv7x7_maxpool_f32(x.data_ptr(), sn, sc, sh, sw, wp.data_ptr(), b.data_ptr(), yp.data_ptr(), ym.data_ptr(), N, 512, 512, st)
nwg = N * 16 * 8
tr = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
lib.cnl_stem5_set_trace.argtypes = [ctypes.c_void_p]
for _ in range(3):
    g()
torch.cuda.synchronize()
assert lib.cnl_stem5_set_trace(ctypes.c_void_p(tr.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g(); e1.record(); torch.cuda.synchronize()
print(f"traced launch (with the seam zeroing): {e0.elapsed_time(e1) * 1e3:.1f} us")
t = tr.cpu().numpy().reshape(nwg, 8)
t0 = t[:, 0].min()
T = (t[:, :5] - t0) / 100.0            # us
hw, xcc = t[:, 5], t[:, 6] & 0xF
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
slot = hw & 0xF
names = ["patch + weight DMA", "scan + split", "K loop", "epilogue + stores"]
d = np.diff(T, axis=1)
print("kernel span %.1f us;  workgroups %d;  CUs seen %d;  wave slots seen %s" % (T[:, 4].max(), nwg, len(np.unique(cu)), np.unique(slot)))
for i, nm in enumerate(names):
    print(f"  {nm:22s} median {np.median(d[:, i]):6.2f} us   p10 {np.percentile(d[:, i], 10):6.2f}   p90 {np.percentile(d[:, i], 90):6.2f}")
print(f"  {'workgroup lifetime':22s} median {np.median(T[:, 4] - T[:, 0]):6.2f} us")
ucu = np.unique(cu)
samples = np.arange(5.0, T[:, 4].max() - 5.0, 0.25)
tot = {k: 0.0 for k in range(4)}
res = 0.0
for c in ucu[:64]:
    idx = np.where(cu == c)[0]
    live = np.array([((T[idx, 0] <= s) & (T[idx, 4] > s)).sum() for s in samples])
    ink = np.array([((T[idx, 2] <= s) & (T[idx, 3] > s)).sum() for s in samples])
    res += live.mean()
    for k in range(4):
        tot[k] += (ink == k).mean()
print(f"per CU (64 CUs, {len(samples)} time points): resident workgroups {res / 64:.2f}")
for k in range(3):
    print(f"  fraction of the time with {k} workgroup(s) of the CU in the K loop: {tot[k] / 64:.3f}")
c = ucu[len(ucu) // 2]
idx = np.where(cu == c)[0]
idx = idx[np.argsort(T[idx, 0])]
print(f"timeline of CU {int(c):#x} ({len(idx)} workgroups): block  slot  start  staged  planes  k-done  end")
for q in idx[:20]:
    print(f"   {q:5d}  {int(slot[q]):2d}  " + "  ".join(f"{v:7.2f}" for v in T[q]))
