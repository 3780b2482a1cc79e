"""centernet_lightning_amd — MI355X-native (gfx950) CenterNet inference hot path.

Drop-in for the detection hot path of gau-nernst/centernet-lightning: `build_centernet()`,
`CenterNet.forward()`, `gather_detection2d()` / `gather_tracking2d()`.  Python orchestrates; all compute
is in libcenternet_gfx950.so (hand-written HIP kernels, C ABI in include/centernet_gfx950.h).
"""
from .config import load_config
from .models import CenterNet, DetectionOutput, TrackingOutput, build_centernet
from .collate import (Collator, all_gather_records, collate_detections, pack_detections, shard_range, unpack_detections)
from .tracker import Tracker, Track, TrackState, build_tracker, match_with_threshold
from . import decode, formats
from .export import TraceableCenterNet, export_onnx, export_torchscript

__all__ = ["CenterNet", "build_centernet", "load_config", "DetectionOutput", "TrackingOutput", "decode",
           "collate_detections", "Collator", "all_gather_records", "pack_detections", "unpack_detections", "shard_range",
           "Tracker", "Track", "TrackState", "build_tracker", "match_with_threshold", "formats", "TraceableCenterNet", "export_torchscript", "export_onnx"]
