"""PCIe-inclusive rate of the path when frames arrive as HOST uint8 buffers (the boundary itself takes device tensors):
pinned uint8 HWC frames -> H2D -> CenterNet.forward_uint8 (A.Normalize inside the stem kernel: no fp32 image in HBM) -> gather_detection2d,
C1 shape (32 x 512 x 512); `--hd`: 1080 x 1920 frames, resized on the device to 512 x 512 (cnl_resize_bilinear_u8, cv2 INTER_LINEAR rule) first.
Reported: images/s with the frames resident, with the copy serialised on the compute stream, and with the next batch's copy on a second stream."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
import bench  # noqa: E402
import centernet_lightning_amd as cl  # noqa: E402

HD = "--hd" in sys.argv
B, H, W, STEPS = 32, (1080 if HD else 512), (1920 if HD else 512), 20
torch.manual_seed(0)
model = bench.synthetic_weights_(cl.build_centernet(os.path.join(ROOT, "centernet-lightning_amd", "configs", "resnet34_simple.yaml"))).cuda()
host = [torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]


def step_dev(u8):
    return model.gather_detection2d(model.forward_uint8(u8, resize=(512, 512)) if HD else model.forward_uint8(u8))


with torch.no_grad():
    dev = host[0].cuda()
    for _ in range(3):
        step_dev(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        step_dev(dev)
    torch.cuda.synchronize()
    t_resident = (time.perf_counter() - t0) / STEPS
    t0 = time.perf_counter()
    for i in range(STEPS):
        step_dev(host[i & 1].cuda(non_blocking=True))
    torch.cuda.synchronize()
    t_serial = (time.perf_counter() - t0) / STEPS
    copy = torch.cuda.Stream()
    nxt = None
    with torch.cuda.stream(copy):
        nxt = host[0].cuda(non_blocking=True)
    t0 = time.perf_counter()
    for i in range(STEPS):
        torch.cuda.current_stream().wait_stream(copy)
        cur = nxt
        cur.record_stream(torch.cuda.current_stream())
        with torch.cuda.stream(copy):
            nxt = host[(i + 1) & 1].cuda(non_blocking=True)
        step_dev(cur)
    torch.cuda.synchronize()
    t_overlap = (time.perf_counter() - t0) / STEPS
mb = B * H * W * 3 / 1e6
print(f"uint8 frames resident in HBM : {B / t_resident:8.1f} images/s  ({t_resident * 1e3:.2f} ms/step)")
print(f"host -> device on the compute stream ({mb:.1f} MB/step): {B / t_serial:8.1f} images/s  ({t_serial * 1e3:.2f} ms/step)")
print(f"host -> device on a second stream (overlapped)   : {B / t_overlap:8.1f} images/s  ({t_overlap * 1e3:.2f} ms/step)")
