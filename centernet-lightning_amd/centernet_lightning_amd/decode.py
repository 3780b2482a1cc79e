"""Host-side wrappers of the fused HIP decode (cnl_decode_f32 and the standalone gathers).

Mirrors the reference's decode surface: CenterNet.decode_detections / get_topk_from_heatmap /
gather_and_decode_boxes (models/centernet.py:229-304), EmbeddingHead.gather_at_indices
(models/fairmot.py:63-73).  Inputs are logical-NCHW fp32 HIP tensors with ANY strides (NHWC views from
this package's forward are zero-copy; contiguous NCHW tensors from reference-style callers work too).
"""
import ctypes

import torch

from . import _lib
from ._lib import DecodeParams


def _require_cuda_f32(name, t, ndim=4):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a HIP device ('cuda'); there is no CPU fallback in the product path")
    if t.dtype != torch.float32 or t.dim() != ndim:
        raise ValueError(f"{name}: expected float32 with {ndim} dims, got {t.dtype} {tuple(t.shape)}")


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def decode(heatmap, box_2d, reid=None, num_detections=100, nms_kernel=3, normalize_boxes=False, box_log=False,
           box_multiplier=1.0, stride=4):
    """Fused pseudo-NMS + top-k + gathers.  Returns dict(scores [N,k] f32, indices [N,k] i64, labels [N,k] i64,
    boxes [N,k,4] f32 [, embeddings [N,k,E] f32]) on heatmap.device, detections sorted by score descending
    (ties: lower flat index first)."""
    lib = _lib.load()
    _require_cuda_f32("heatmap", heatmap)
    _require_cuda_f32("box_2d", box_2d)
    N, C, H, W = heatmap.shape
    if tuple(box_2d.shape) != (N, 4, H, W):
        raise ValueError(f"box_2d shape {tuple(box_2d.shape)} does not match heatmap {tuple(heatmap.shape)}")
    dev = heatmap.device
    k = int(num_detections)
    E = 0
    if reid is not None:
        _require_cuda_f32("reid", reid)
        if reid.shape[0] != N or tuple(reid.shape[2:]) != (H, W):
            raise ValueError(f"reid shape {tuple(reid.shape)} does not match heatmap {tuple(heatmap.shape)}")
        E = reid.shape[1]
    # ONE allocation for the outputs and the workspace, carved up AFTER the launch: the host-side tensor bookkeeping then overlaps the
    # kernels instead of delaying them (single-call latency: 84 -> 62 us at C1 came from the kernels, the rest is this ordering)
    ws_bytes = max(int(lib.cnl_decode_workspace_bytes(N, H, W)), 16)
    al = lambda n: (n + 255) & ~255
    o_idx, o_lab = 0, al(N * k * 8)
    o_box = o_lab + al(N * k * 8)
    o_sc = o_box + al(N * k * 16)
    o_emb = o_sc + al(N * k * 4)
    o_ws = o_emb + al(N * k * E * 4)
    switch = torch.cuda.current_device() != dev.index
    if switch:
        prev = torch.cuda.current_device()
        torch.cuda.set_device(dev)
    try:
        # the outputs share ONE allocation (carved after the launch); the stage-1 workspace (N*H*W*8 bytes: tens of MB at large N) is a
        # SEPARATE one, freed with this call — a caller that keeps `scores` of many batches alive must not pin a workspace per batch
        buf = torch.empty((o_ws,), device=dev, dtype=torch.uint8)
        ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
        base = buf.data_ptr()
        p = DecodeParams()
        p.heat = heatmap.data_ptr()
        p.heat_sn, p.heat_sc, p.heat_sh, p.heat_sw = heatmap.stride()
        p.box = box_2d.data_ptr()
        p.box_sn, p.box_sc, p.box_sh, p.box_sw = box_2d.stride()
        if E:
            p.reid = reid.data_ptr()
            p.reid_sn, p.reid_sc, p.reid_sh, p.reid_sw = reid.stride()
            p.emb = base + o_emb
        p.N, p.C, p.H, p.W, p.E = N, C, H, W, E
        p.k, p.nms_kernel = k, int(nms_kernel)
        p.normalize_boxes, p.box_log = int(bool(normalize_boxes)), int(bool(box_log))
        p.box_multiplier, p.stride = float(box_multiplier), float(stride)
        p.scores, p.indices, p.labels, p.boxes = base + o_sc, base + o_idx, base + o_lab, base + o_box
        p.workspace, p.workspace_bytes = ws.data_ptr(), ws_bytes
        _lib.check(lib.cnl_decode_f32(ctypes.byref(p), _stream(dev)), "cnl_decode_f32")
    finally:
        if switch:
            torch.cuda.set_device(prev)
    view = lambda off, nbytes, dtype, shape: buf[off:off + nbytes].view(dtype).view(shape)
    out = {"scores": view(o_sc, N * k * 4, torch.float32, (N, k)), "indices": view(o_idx, N * k * 8, torch.int64, (N, k)),
           "labels": view(o_lab, N * k * 8, torch.int64, (N, k)), "boxes": view(o_box, N * k * 16, torch.float32, (N, k, 4))}
    if E:
        out["embeddings"] = view(o_emb, N * k * E * 4, torch.float32, (N, k, E))
    return out


def gather_boxes(box_2d, indices, normalize_boxes=False, box_log=False, box_multiplier=1.0, stride=4):
    """CenterNet.gather_and_decode_boxes (centernet.py:263-304) at caller-supplied indices [N,k] (int64)."""
    lib = _lib.load()
    _require_cuda_f32("box_2d", box_2d)
    N, four, H, W = box_2d.shape
    if four != 4 or indices.dim() != 2 or indices.shape[0] != N:
        raise ValueError("gather_boxes: box_2d must be [N,4,H,W] and indices [N,k]")
    idx = indices.to(device=box_2d.device, dtype=torch.int64).contiguous()
    k = idx.shape[1]
    with torch.cuda.device(box_2d.device):
        boxes = torch.empty((N, k, 4), device=box_2d.device, dtype=torch.float32)
        sn, sc, sh, sw = box_2d.stride()
        _lib.check(lib.cnl_gather_boxes_f32(box_2d.data_ptr(), sn, sc, sh, sw, idx.data_ptr(), boxes.data_ptr(), N, H, W, k,
                                            int(bool(normalize_boxes)), int(bool(box_log)), float(box_multiplier), float(stride),
                                            _stream(box_2d.device)), "cnl_gather_boxes_f32")
    return boxes


def gather_embeddings(reid, indices):
    """EmbeddingHead.gather_at_indices (fairmot.py:63-73): reid [N,E,H,W], indices [N,k] -> [N,k,E]."""
    lib = _lib.load()
    _require_cuda_f32("reid", reid)
    N, E, H, W = reid.shape
    if indices.dim() != 2 or indices.shape[0] != N:
        raise ValueError("gather_embeddings: indices must be [N,k]")
    idx = indices.to(device=reid.device, dtype=torch.int64).contiguous()
    k = idx.shape[1]
    with torch.cuda.device(reid.device):
        emb = torch.empty((N, k, E), device=reid.device, dtype=torch.float32)
        sn, sc, sh, sw = reid.stride()
        _lib.check(lib.cnl_gather_embeddings_f32(reid.data_ptr(), sn, sc, sh, sw, idx.data_ptr(), emb.data_ptr(), N, E, H, W, k,
                                                 _stream(reid.device)), "cnl_gather_embeddings_f32")
    return emb
