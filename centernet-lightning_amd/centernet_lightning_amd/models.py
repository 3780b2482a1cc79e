"""`CenterNet` / `build_centernet()` — the drop-in boundary of the MI355X hot path.

Keeps the reference's Gen-A Python surface (README.md:29-37,94-101; docs/implementation.md:70-81;
tests/test_models.py:61-99; models/fairmot.py:138-151) with the Gen-B arithmetic that is actually in the
reference tree (models/meta.py:21-47, models/centernet.py:229-304):

    model = build_centernet("configs/base_resnet34_fpn.yaml").cuda().eval()
    heatmap, box_2d = model(images)                     # namedtuple; heatmap is post-sigmoid
    dets = model.gather_detection2d(heatmap, box_2d)    # {"bboxes","labels","scores"}

Everything between the input tensor and those outputs runs in libcenternet_gfx950.so (HIP, gfx950).
There is no CPU path: CPU tensors raise.  Training, datasets and evaluation stay with the reference.
"""
import ctypes
import math
from collections import OrderedDict, namedtuple
from typing import Any, Dict, Union

import torch
from torch import nn

from . import decode as _decode
from .collate import collate_detections
from .config import model_section
from .engine import Engine
from .params import GenericHead, ResNetBackbone, build_neck

DetectionOutput = namedtuple("DetectionOutput", ["heatmap", "box_2d"])
TrackingOutput = namedtuple("TrackingOutput", ["heatmap", "box_2d", "reid"])

_HEAD_CHANNELS = {"box_2d": lambda cfg: 4}


class _Head(GenericHead):
    """GenericHead parameters + the Gen-A per-head decode calls used by FairMOT.gather_tracking2d
    (models/fairmot.py:141-143)."""

    def __init__(self, name, in_channels, out_channels, model, **kw):
        super().__init__(in_channels, out_channels, **kw)
        self.head_name = name
        self._model_ref = [model]          # list: keep the parent out of the module tree

    # heads["heatmap"].gather_topk(heatmap, nms_kernel=3, num_detections=100) -> scores, indices, labels
    def gather_topk(self, heatmap, nms_kernel=3, num_detections=100):
        dummy_box = heatmap[:, :1].expand(-1, 4, -1, -1)          # strides only; the boxes are discarded
        out = _decode.decode(heatmap, dummy_box, None, num_detections, nms_kernel)
        return out["scores"], out["indices"], out["labels"]

    # heads["box_2d"].gather_at_indices(box_2d, indices, normalize_bbox=False, stride=4) / reid.gather_at_indices(reid, idx)
    def gather_at_indices(self, x, indices, normalize_bbox=False, stride=None):
        if self.head_name == "box_2d":
            m = self._model_ref[0]
            return _decode.gather_boxes(x, indices, normalize_bbox, m.box_log, m.box_multiplier,
                                        stride if stride is not None else m.output_stride)
        return _decode.gather_embeddings(x, indices)


class CenterNet(nn.Module):
    """CenterNet(backbone: dict, neck: dict, output_heads: dict, task: str, **ignored) — tests/test_models.py:62."""

    def __init__(self, backbone: Dict[str, Any], neck: Dict[str, Any], output_heads: Dict[str, Any], task: str = "detection",
                 num_detections: int = 100, nms_kernel: int = 3, box_log: bool = False, box_multiplier: float = 1.0,
                 **ignored):
        super().__init__()
        if task not in ("detection", "tracking"):
            raise ValueError(f"unknown task {task!r}")
        if "heatmap" not in output_heads:
            raise ValueError("output_heads must contain 'heatmap' (docs/implementation.md:60)")
        if "box_2d" not in output_heads:
            raise ValueError("output_heads must contain 'box_2d' for the detection/tracking decode")
        if task == "tracking" and "reid" not in output_heads:
            raise ValueError("task 'tracking' needs a 'reid' head (configs/base_tracking_resnet34_fpn.yaml:26-30)")
        self.task = task
        # the `model:` section this instance was built from (export.py rebuilds an identical model from it)
        self.config_section = {"backbone": dict(backbone), "neck": dict(neck), "output_heads": {k: dict(v or {}) for k, v in output_heads.items()},
                               "task": task, "num_detections": int(num_detections), "nms_kernel": int(nms_kernel), "box_log": bool(box_log),
                               "box_multiplier": float(box_multiplier)}
        self.backbone = ResNetBackbone(**backbone)
        self.neck = build_neck(neck, self.backbone.out_channels)
        self.output_stride = self.backbone.output_stride // self.neck.upsample_stride      # meta.py:96
        self.stride = self.output_stride
        self.num_classes = int(output_heads["heatmap"]["num_classes"])
        # Gen-B hyper-parameters of the decode (centernet.py:82-83,93-94)
        self.num_detections, self.nms_kernel = int(num_detections), int(nms_kernel)
        self.box_log, self.box_multiplier = bool(box_log), float(box_multiplier)

        heads = OrderedDict()
        in_c = self.neck.out_channels
        for name, cfg in output_heads.items():
            cfg = dict(cfg or {})
            if name == "heatmap":
                out_c, d_width, d_depth = self.num_classes, 256, 3                       # meta.py:22
            elif name == "box_2d":
                out_c, d_width, d_depth = 4, 256, 3
            elif name == "reid":
                out_c, d_width, d_depth = int(cfg.get("emb_dim", 64)), 256, 1            # fairmot.py:20
            else:
                raise ValueError(f"output head '{name}' is outside the MI355X hot-path scope (heatmap, box_2d, reid)")
            heads[name] = _Head(name, in_c, out_c, self, width=int(cfg.get("width", d_width)),
                                depth=int(cfg.get("depth", d_depth)), init_bias=cfg.get("init_bias"))
        self.heads = nn.ModuleDict(heads)
        self._engine = Engine(self)
        self.eval()

    # ------------------------------------------------------------------ weights plumbing
    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._engine.invalidate()
        return out

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        if hasattr(self, "_engine"):
            self._engine.invalidate()
        return out

    def refresh_weights(self):
        """Force re-folding of BatchNorm and re-packing of the OHWI weights.  Not normally needed: the engine notices in-place
        writes to any parameter / buffer (version counters; e.g. model.backbone.load_state_dict(...)) before the next forward."""
        self._engine.invalidate()

    def set_kernel_options(self, **kw):
        """Kernel choice of the launch plan, explicit and per model (engine.KernelOptions): algo="auto" | "f2" | "f32", winograd,
        up2, absmax_handover, stem_fused_pool, reuse_buffers.  The HIP library reads no environment variables."""
        return self._engine.set_options(**kw)

    def train(self, mode: bool = True):
        if mode:
            raise RuntimeError("this CenterNet is the inference hot path only (eval-mode BatchNorm is folded into the HIP "
                               "kernels); training stays with the reference's Lightning module")
        return super().train(False)

    # ------------------------------------------------------------------ forward
    def get_encoded_outputs(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Forward pass returning a dict of logits (heatmap BEFORE sigmoid) — docs/implementation.md:77."""
        return dict(self._engine.forward(x, sigmoid=False))

    get_output_dict = get_encoded_outputs          # alias used at utils/image_annotate.py:220, fairmot.py:88

    def forward(self, x: torch.Tensor):
        """namedtuple(heatmap after sigmoid, box_2d[, reid]) — docs/implementation.md:78; tests/test_models.py:88-99."""
        out = self._engine.forward(x, sigmoid=True)
        if "reid" in out:
            return TrackingOutput(out["heatmap"], out["box_2d"], out["reid"])
        return DetectionOutput(out["heatmap"], out["box_2d"])

    # ------------------------------------------------------------------ step before the path
    IMAGENET_MEAN = (0.485, 0.456, 0.406)      # datasets/utils.py:9-10 of the reference
    IMAGENET_STD = (0.229, 0.224, 0.225)

    def preprocess_uint8(self, images: torch.Tensor, mean=IMAGENET_MEAN, std=IMAGENET_STD) -> torch.Tensor:
        """uint8 RGB frames [N,H,W,3] on the GPU -> normalised fp32 batch, logical [N,3,H,W] (channels_last storage, read
        zero-copy by forward()).  Same arithmetic as albumentations A.Normalize + ToTensorV2 (README.md:79-87)."""
        import ctypes
        import numpy as np
        from . import _lib
        if not (isinstance(images, torch.Tensor) and images.is_cuda):
            raise RuntimeError("preprocess_uint8 runs on HIP devices only (no CPU fallback)")
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3:
            raise ValueError(f"expected uint8 [N,H,W,3], got {images.dtype} {tuple(images.shape)}")
        images = images.contiguous()
        N, H, W, _ = images.shape
        m = np.array(mean, dtype=np.float32) * np.float32(255.0)
        r = np.reciprocal(np.array(std, dtype=np.float32) * np.float32(255.0), dtype=np.float32)
        lib = _lib.load()
        with torch.cuda.device(images.device):
            out = torch.empty((N, H, W, 3), device=images.device, dtype=torch.float32)
            _lib.check(lib.cnl_normalize_u8_nhwc_f32(images.data_ptr(), out.data_ptr(), N, H, W,
                                                     m.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                                     r.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                                     ctypes.c_void_p(torch.cuda.current_stream(images.device).cuda_stream)),
                       "cnl_normalize_u8_nhwc_f32")
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def _norm_constants(mean, std):
        import numpy as np
        m = np.array(mean, dtype=np.float32) * np.float32(255.0)
        r = np.reciprocal(np.array(std, dtype=np.float32) * np.float32(255.0), dtype=np.float32)
        return (ctypes.c_float * 3)(*m.tolist()), (ctypes.c_float * 3)(*r.tolist())

    def resize_uint8(self, images: torch.Tensor, height: int, width: int) -> torch.Tensor:
        """albumentations A.Resize(height, width) (README.md:84) = cv2.resize(..., INTER_LINEAR) on uint8 frames [N,H,W,C] -> [N,height,width,C]
        (cnl_resize_bilinear_u8: OpenCV's 8-bit fixed-point rule, bit-exact against oracle/decode_ref.resize_bilinear_u8)."""
        from . import _lib
        if not (isinstance(images, torch.Tensor) and images.is_cuda):
            raise RuntimeError("resize_uint8 runs on HIP devices only (no CPU fallback)")
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] > 4:
            raise ValueError(f"expected uint8 [N,H,W,C<=4], got {images.dtype} {tuple(images.shape)}")
        images = images.contiguous()
        N, H, W, C = images.shape
        lib = _lib.load()
        with torch.cuda.device(images.device):
            out = torch.empty((N, int(height), int(width), C), device=images.device, dtype=torch.uint8)
            _lib.check(lib.cnl_resize_bilinear_u8(images.data_ptr(), out.data_ptr(), N, H, W, int(height), int(width), C,
                                                  ctypes.c_void_p(torch.cuda.current_stream(images.device).cuda_stream)), "cnl_resize_bilinear_u8")
        return out

    def forward_uint8(self, images: torch.Tensor, resize=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        """The reference's inference pre-processing fused into the path (README.md:79-101): uint8 HWC frames [N,H,W,3] ->
        [A.Resize(*resize) ->] A.Normalize -> forward().  The normalisation happens on the stem kernel's staged patch
        (cnl_stem_conv7x7_u8), so no fp32 image tensor is ever written: bit-identical to forward(preprocess_uint8(images))."""
        if resize is not None:
            images = self.resize_uint8(images, int(resize[0]), int(resize[1]))
        m, r = self._norm_constants(mean, std)
        out = self._engine.forward_u8(images.contiguous(), m, r, sigmoid=True)
        if "reid" in out:
            return TrackingOutput(out["heatmap"], out["box_2d"], out["reid"])
        return DetectionOutput(out["heatmap"], out["box_2d"])

    # ------------------------------------------------------------------ decode (Gen-A names)
    def gather_detection2d(self, heatmap, box_2d=None, num_detections=100, nms_kernel=3, normalize_bbox=False):
        """-> {"bboxes": [N,k,4] x1y1x2y2, "labels": [N,k] i64, "scores": [N,k]} (README.md:58-64,97-101).
        Accepts the forward() namedtuple as the single argument.  `heatmap` is the post-sigmoid heatmap."""
        if box_2d is None and isinstance(heatmap, (tuple, list)):
            heatmap, box_2d = heatmap[0], heatmap[1]
        out = _decode.decode(heatmap, box_2d, None, num_detections, nms_kernel, normalize_bbox, self.box_log,
                             self.box_multiplier, self.output_stride)
        return {"bboxes": out["boxes"], "labels": out["labels"], "scores": out["scores"]}

    def gather_tracking2d(self, heatmap, box_2d=None, reid=None, num_detections=100, nms_kernel=3, normalize_bbox=False):
        """FairMOT.gather_tracking2d (fairmot.py:138-151): adds "embeddings": [N,k,E]."""
        if box_2d is None and isinstance(heatmap, (tuple, list)):
            heatmap, box_2d, reid = heatmap[0], heatmap[1], heatmap[2]
        out = _decode.decode(heatmap, box_2d, reid, num_detections, nms_kernel, normalize_bbox, self.box_log,
                             self.box_multiplier, self.output_stride)
        return {"bboxes": out["boxes"], "labels": out["labels"], "scores": out["scores"], "embeddings": out["embeddings"]}

    # ------------------------------------------------------------------ decode (Gen-B names, centernet.py:229-304)
    def decode_detections(self, heatmap, box_offsets, normalize_boxes=False):
        out = _decode.decode(heatmap, box_offsets, None, self.num_detections, self.nms_kernel, normalize_boxes, self.box_log,
                             self.box_multiplier, self.stride)
        return {"boxes": out["boxes"], "scores": out["scores"], "labels": out["labels"]}

    def get_topk_from_heatmap(self, heatmap, pseudo_nms=True):
        return self.heads["heatmap"].gather_topk(heatmap, self.nms_kernel if pseudo_nms else 1, self.num_detections)

    @staticmethod
    def gather_and_decode_boxes(box_offsets, indices, normalize_boxes=False, box_log=False, box_multiplier=1.0, stride=4):
        return _decode.gather_boxes(box_offsets, indices, normalize_boxes, box_log, box_multiplier, stride)

    # ------------------------------------------------------------------ multi-GPU
    def collate(self, detections: Dict[str, torch.Tensor], group=None):
        """All-gather this rank's detections over the process group (RCCL on HIP) — eval/coco.py:10-18 precedent."""
        return collate_detections(detections, group)


def build_centernet(config: Union[str, Dict[str, Any]]) -> CenterNet:
    """Build from a YAML path or a dict (README.md:31-37).  Reads the `model:` section only."""
    m = model_section(config)
    extra = {k: m[k] for k in ("num_detections", "nms_kernel", "box_log", "box_multiplier") if k in m}
    return CenterNet(m["backbone"], m["neck"], m["output_heads"], m.get("task", "detection"), **extra)
