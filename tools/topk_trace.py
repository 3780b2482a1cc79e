"""Phase stamps of cnl_decode::topk_kernel (block 0..3, thread 0; s_memtime) from a -DTK_TIMING build:
make -C centernet-lightning_amd/csrc variant TAG=tktime EXTRA=-DTK_TIMING;  CENTERNET_GFX950_LIB=tools/ablibs/libcnl_tktime.so python tools/topk_trace.py [c1|c4|net]
(net: the heat map / boxes of bench.py's C1 model on its own seeded input instead of synthetic maps)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "centernet-lightning_amd"))
from centernet_lightning_amd import decode as D, _lib
which = sys.argv[1] if len(sys.argv) > 1 else "c1"
N, C, H, W, k, E = (32, 2, 152, 272, 300, 64) if which == "c4" else (32, 80, 128, 128, 100, 0)
g = torch.Generator(device="cuda").manual_seed(0)
if which == "net":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    model = bench.build_model("simple")
    x = torch.rand(N, 3, 512, 512, generator=torch.Generator().manual_seed(0)).cuda()
    with torch.no_grad():
        out = model(x)
    heat, box, emb = out[0], out[1], None
    print("net: heat", tuple(heat.shape), "min/max %.4g %.4g" % (heat.min().item(), heat.max().item()))
else:
    heat = torch.randn(N, H, W, C, device="cuda", generator=g).sub_(2.19).sigmoid_().permute(0, 3, 1, 2)
    box = (torch.rand(N, H, W, 4, device="cuda", generator=g) * 16).permute(0, 3, 1, 2)
    emb = torch.randn(N, H, W, E, device="cuda", generator=g).permute(0, 3, 1, 2) if E else None
for _ in range(3): D.decode(heat, box, emb, k, 3)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.lib_path())
buf = (ctypes.c_ulonglong * (64 * 16))()
assert lib.cnl_debug_topk_stamps(buf) == 0
names = {0: "start", 11: "zeroed", 9: "keys loaded (+ LDS copy, thread maxima)", 10: "wave sort of the maxima", 1: "barrier", 2: "compaction", 3: "rank", 5: "(rank end)", 6: "gathers"}
order = [0, 11, 9, 10, 1, 2, 3, 5, 6]
for b in range(3):
    r = [buf[b * 16 + i] for i in range(16)]
    line = f"block {b}: candidates {r[8]}: "
    for a_, b_ in zip(order[:-1], order[1:]):
        line += f"{names[b_]} {r[b_] - r[a_]} | "
    print(line + f"total {r[6] - r[0]} cycles; inside the compaction (thread 0): bound + prefilter scan {r[12] - r[1]} | exact re-reads + emits {r[13] - r[12]} | in-register rounds {r[14] - r[13]} | barrier {r[2] - r[14]}")
