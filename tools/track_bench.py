"""Association-step measurement: per-frame cost of the tracker update on MI355X beside the CPU path.

    python tools/track_bench.py [--k 300] [--objects 60] [--frames 200] [--out profiles/r01_track_bench.json]

Reports, for a synthetic MOT-like stream (oracle/tracker_ref.synth_sequence; detections already resident in HBM, as they are
after gather_tracking2d):
  * costs_kernel_us   — HIP-event time of cnl_track_costs_f32 alone (k x T pairs: 3 float64 dot products of E=64 + box cost);
                        algorithmic bytes = k*(E+5)*4 + T*(E+4)*4 read, n*T*12 written -> HBM fraction (latency-bound: tiny)
  * update_us         — wall time of one Tracker.update (kernel + the frame's single D2H + scipy Hungarian + apply kernel)
  * d2h_bytes         — bytes that cross PCIe per frame (cost matrices) vs the reference's k*(6+E)*4 detections
  * cpu_update_us     — the CPU restatement of the reference's Tracker.update on the same stream (numpy + scipy, 1 thread)
"""
import argparse
import ctypes
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import centernet_lightning_amd as cl          # noqa: E402
from centernet_lightning_amd import _lib      # noqa: E402
import tracker_ref                            # noqa: E402  (CPU baseline + input recipe only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=300)
    ap.add_argument("--objects", type=int, default=60)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--emb", type=int, default=64)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    warnings.simplefilter("ignore")
    seq = tracker_ref.synth_sequence(0, frames=a.frames, objects=a.objects, k=a.k, emb_dim=a.emb)
    dev = torch.device("cuda:0")
    dseq = [tuple(torch.from_numpy(x).to(dev) for x in fr) for fr in seq]
    hseq_boxes = [fr[0] for fr in seq]

    trk = cl.Tracker(model=None, device=dev)
    for fr in dseq[:10]:
        trk.update(*fr)
    trk.reset()
    torch.cuda.synchronize()
    t_upd, n_tracks, d2h = [], [], []
    for fr in dseq:
        T = len(trk.tracks)
        t0 = time.perf_counter()
        trk.update(*fr)
        torch.cuda.synchronize()
        t_upd.append((time.perf_counter() - t0) * 1e6)
        n_tracks.append(T)
        d2h.append(trk.d2h_bytes)

    # the costs kernel alone, at the stream's typical table size
    T = int(np.median(n_tracks))
    lib = _lib.load()
    emb_t = torch.randn(T, a.emb, device=dev)
    box_t = torch.rand(T, 4, device=dev)
    bx, lb, sc, em = dseq[len(dseq) // 2]
    n_det = torch.zeros(1, dtype=torch.int32, device=dev)
    idx = torch.zeros(a.k, dtype=torch.int32, device=dev)
    reid = torch.zeros(a.k * T, dtype=torch.float64, device=dev)
    bc = torch.zeros(a.k * T, dtype=torch.float32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    call = lambda: _lib.check(lib.cnl_track_costs_f32(em.data_ptr(), bx.data_ptr(), sc.data_ptr(), a.k, a.emb, 0.3, emb_t.data_ptr(),
                                                      box_t.data_ptr(), T, 1, n_det.data_ptr(), idx.data_ptr(), reid.data_ptr(),
                                                      bc.data_ptr(), stream))
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    ks = []
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        ks.append(e0.elapsed_time(e1) * 1e3)
    n = int(n_det.item())
    alg_bytes = a.k * (a.emb + 5) * 4 + T * (a.emb + 4) * 4 + n * T * 12
    k_us = float(np.median(ks))

    # CPU path (restated reference) on the same stream
    cpu = tracker_ref.Tracker()
    t_cpu = []
    for fr in seq:
        t0 = time.perf_counter()
        cpu.update(*fr)
        t_cpu.append((time.perf_counter() - t0) * 1e6)
    same = [t.track_id for t in cpu.tracks] == [t.track_id for t in trk.tracks]

    res = {
        "workload": f"tracker association, k={a.k} detections/frame, E={a.emb}, {a.objects} objects, {a.frames} frames, box_cost=iou",
        "tracks_median": T, "n_det_sample": n,
        "costs_kernel_us_p50": round(k_us, 2),
        "costs_kernel_algorithmic_bytes": alg_bytes,
        "costs_kernel_GBps": round(alg_bytes / k_us / 1e3, 2), "hbm_peak_GBps": 8000,
        "update_us_p50": round(float(np.median(t_upd[10:])), 1), "update_us_p90": round(float(np.percentile(t_upd[10:], 90)), 1),
        "d2h_bytes_per_frame_median": int(np.median(d2h)), "reference_d2h_bytes_per_frame": a.k * (6 + a.emb) * 4, "d2h_includes_boxes_scores_labels_bytes": a.k * 24,
        "cpu_update_us_p50": round(float(np.median(t_cpu[10:])), 1), "cpu_cores_used": 1, "cpu_kind": "port (oracle/tracker_ref.py)",
        "same_track_ids_as_cpu": bool(same),
    }
    print(json.dumps(res))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
