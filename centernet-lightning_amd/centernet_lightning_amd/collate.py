"""Multi-GPU collation of detections: one RCCL all-gather of a fixed-size packed record per batch.

The reference merges per-rank prediction lists with `dist.all_gather_object` (pickle, variable size;
eval/coco.py:10-18).  Here every rank holds exactly [N_local, k] detections, so a plain all-gather of a
float32 record buffer [N_local, k, 6(+E)] is enough: x1 y1 x2 y2 score label-bits [embedding...].
Images are sharded contiguously: rank r owns global images [r*N_local, (r+1)*N_local), which is also the
order of the gathered buffer.  world_size == 1 is a no-op, like the reference (eval/coco.py:11-13).

`torch.distributed` backend "nccl" is RCCL on ROCm (xGMI within the node); the gloo backend is used by the
CPU tests of the protocol.  Pack / unpack are HIP kernels (cnl_pack_detections_f32 / cnl_unpack_detections_f32).
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib

RECORD_FIELDS = 6          # x1, y1, x2, y2, score, label


def shard_range(n_total: int, rank: int, world_size: int):
    """Contiguous batch split: rank r gets images [r*n/W, (r+1)*n/W)."""
    if n_total % world_size:
        raise ValueError(f"global batch {n_total} is not divisible by world size {world_size}")
    per = n_total // world_size
    return rank * per, (rank + 1) * per


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def pack_detections(dets: dict) -> torch.Tensor:
    """dict(bboxes|boxes [N,k,4], scores [N,k], labels [N,k] i64 [, embeddings [N,k,E]]) -> record [N,k,6+E] (HIP)."""
    lib = _lib.load()
    boxes = dets["bboxes"] if "bboxes" in dets else dets["boxes"]
    scores, labels, emb = dets["scores"], dets["labels"], dets.get("embeddings")
    if not boxes.is_cuda:
        raise RuntimeError("pack_detections runs on HIP devices only")
    N, k = scores.shape
    E = emb.shape[-1] if emb is not None else 0
    boxes, scores, labels = boxes.contiguous(), scores.contiguous(), labels.contiguous()
    emb = emb.contiguous() if emb is not None else None
    with torch.cuda.device(boxes.device):
        rec = torch.empty((N, k, RECORD_FIELDS + E), device=boxes.device, dtype=torch.float32)
        _lib.check(lib.cnl_pack_detections_f32(boxes.data_ptr(), scores.data_ptr(), labels.data_ptr(),
                                               emb.data_ptr() if emb is not None else None, rec.data_ptr(), N, k, E,
                                               _stream(boxes.device)), "cnl_pack_detections_f32")
    return rec


def unpack_detections(rec: torch.Tensor, box_key="bboxes") -> dict:
    lib = _lib.load()
    if not rec.is_cuda:
        raise RuntimeError("unpack_detections runs on HIP devices only")
    rec = rec.contiguous()
    N, k, R = rec.shape
    E = R - RECORD_FIELDS
    dev = rec.device
    with torch.cuda.device(dev):
        boxes = torch.empty((N, k, 4), device=dev, dtype=torch.float32)
        scores = torch.empty((N, k), device=dev, dtype=torch.float32)
        labels = torch.empty((N, k), device=dev, dtype=torch.int64)
        emb = torch.empty((N, k, E), device=dev, dtype=torch.float32) if E else None
        _lib.check(lib.cnl_unpack_detections_f32(rec.data_ptr(), boxes.data_ptr(), scores.data_ptr(), labels.data_ptr(),
                                                 emb.data_ptr() if emb is not None else None, N, k, E, _stream(dev)),
                   "cnl_unpack_detections_f32")
    out = {box_key: boxes, "labels": labels, "scores": scores}
    if E:
        out["embeddings"] = emb
    return out


def all_gather_records(rec: torch.Tensor, group=None, force: bool = False) -> torch.Tensor:
    """[N_local,k,R] on every rank -> [world*N_local,k,R] in rank order.  Device-agnostic torch.distributed call
    (RCCL for HIP tensors, gloo for the CPU protocol tests).  No-op without an initialised process group or at
    world size 1 (unless `force`, which runs the collective anyway — used to exercise the RCCL path on one GPU)."""
    if not (dist.is_available() and dist.is_initialized()):
        return rec
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return rec
    rec = rec.contiguous()
    out = torch.empty((world * rec.shape[0],) + tuple(rec.shape[1:]), device=rec.device, dtype=rec.dtype)
    dist.all_gather_into_tensor(out, rec, group=group)
    return out


def collate_detections(dets: dict, group=None, force: bool = False) -> dict:
    """All ranks call this with their local detections; every rank gets the global, rank-ordered detections."""
    box_key = "bboxes" if "bboxes" in dets else "boxes"
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return dets
    return unpack_detections(all_gather_records(pack_detections(dets), group, force), box_key=box_key)
