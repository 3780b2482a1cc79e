"""Every GPU kernel / copy of N steps of forward + gather_detection2d and nothing else (run under rocprofv3 --kernel-trace --stats, read with tools/rocpd_summary.py stats):
what a step launches besides the plan's convs and the decode's two kernels.   python tools/step_kernels.py [simple|fpn|tracking] [steps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "simple"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tracking = cfg == "tracking"
B, H, W = (32, 608, 1088) if tracking else ((64, 512, 512) if cfg == "fpn" else (32, 512, 512))
m = bench.build_model(cfg)
x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(0)).cuda()
with torch.no_grad():
    for _ in range(3):
        o = m(x); d = (m.gather_tracking2d if tracking else m.gather_detection2d)(o, num_detections=100)
    torch.cuda.synchronize()
    print("STEPS BEGIN", flush=True)
    for _ in range(steps):
        o = m(x); d = (m.gather_tracking2d if tracking else m.gather_detection2d)(o, num_detections=100)
    torch.cuda.synchronize()
