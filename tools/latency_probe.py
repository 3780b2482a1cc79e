#!/usr/bin/env python
"""Small-batch latency of the forward + decode: wall per call (launches issued from Python through ctypes) against the sum of the
kernels' own durations, and the same plan replayed from a captured HIP graph.  Usage: python tools/latency_probe.py [N ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
import torch  # noqa: E402
import centernet_lightning_amd as cl  # noqa: E402


def main():
    ns = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    m = cl.CenterNet({"name": "resnet34"}, {"name": "simple"}, {"heatmap": {"num_classes": 80}, "box_2d": {}}).cuda().eval()
    for n in ns:
        x = torch.rand(n, 3, 512, 512, device="cuda")
        for _ in range(5):
            out = m.get_encoded_outputs(x)
            m.gather_detection2d(out["heatmap"], out["box_2d"])
        torch.cuda.synchronize()
        t = time.perf_counter()
        R = 50
        for _ in range(R):
            out = m.get_encoded_outputs(x)
            det = m.gather_detection2d(out["heatmap"], out["box_2d"])
            torch.cuda.synchronize()
        lat = (time.perf_counter() - t) / R
        t = time.perf_counter()
        for _ in range(R):
            out = m.get_encoded_outputs(x)
            det = m.gather_detection2d(out["heatmap"], out["box_2d"])
        t_issue = (time.perf_counter() - t) / R
        torch.cuda.synchronize()
        thr = (time.perf_counter() - t) / R
        # the same work from a captured graph
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                out = m.get_encoded_outputs(x)
                det = m.gather_detection2d(out["heatmap"], out["box_2d"])
            s.synchronize()
            try:
                with torch.cuda.graph(g, stream=s):
                    out = m.get_encoded_outputs(x)
                    det = m.gather_detection2d(out["heatmap"], out["box_2d"])
                ok = True
            except Exception as e:                      # noqa: BLE001
                ok = False
                print("capture failed:", repr(e)[:300])
        line = f"N={n}: latency {lat * 1e3:.3f} ms/call (sync each), back-to-back {thr * 1e3:.3f} ms/call, host issue {t_issue * 1e3:.3f} ms/call"
        if ok:
            torch.cuda.synchronize()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(R):
                g.replay()
                torch.cuda.synchronize()
            glat = (time.perf_counter() - t) / R
            t = time.perf_counter()
            for _ in range(R):
                g.replay()
            torch.cuda.synchronize()
            gthr = (time.perf_counter() - t) / R
            line += f" | graph replay: latency {glat * 1e3:.3f}, back-to-back {gthr * 1e3:.3f} ms/call"
        print(line, flush=True)


if __name__ == "__main__":
    main()
