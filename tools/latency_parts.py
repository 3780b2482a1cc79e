"""N = 1 latency in parts (30 calls back to back between HIP events, best of 5): forward alone, decode alone, both — run in two checkouts to A/B them."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
m = bench.build_model("simple")
x = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(1)).cuda()
def t(fn, calls=30, rounds=5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(calls): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / calls)
    return best
with torch.no_grad():
    out = m(x)
    print("forward %.4f ms | decode %.4f ms | both %.4f ms" % (t(lambda: m(x)), t(lambda: m.gather_detection2d(out, num_detections=100)),
                                                         t(lambda: m.gather_detection2d(m(x), num_detections=100))))
