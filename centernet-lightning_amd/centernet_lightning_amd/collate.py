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


def pack_detections(dets: dict, out: torch.Tensor = None) -> torch.Tensor:
    """dict(bboxes|boxes [N,k,4], scores [N,k], labels [N,k] i64 [, embeddings [N,k,E]]) -> record [N,k,6+E] (HIP); `out`: a
    preallocated record to fill instead of a fresh one."""
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
        rec = out if out is not None else torch.empty((N, k, RECORD_FIELDS + E), device=boxes.device, dtype=torch.float32)
        if tuple(rec.shape) != (N, k, RECORD_FIELDS + E) or rec.dtype != torch.float32 or not rec.is_contiguous():
            raise ValueError(f"pack_detections: out must be a contiguous float32 [{N},{k},{RECORD_FIELDS + E}] tensor")
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


class Collator:
    """Pipelined collation for a steady stream of batches ("all-gather on a side stream, overlapped with the next
    batch"): submit() packs this rank's detections into a PERSISTENT record slot and starts the all-gather of that slot on a side
    stream behind an event; result() makes the caller's stream wait for the gather's event and unpacks.  With `depth` slots the
    gather of batch i runs while batch i+1 .. i+depth-1 are computed; nothing is allocated per step after the first.

        c = Collator(); pending = None
        for x in batches:
            h = c.submit(model.gather_detection2d(model(x)))
            if pending is not None: use(c.result(pending))
            pending = h
        use(c.result(pending))

    World size 1 (or no process group) is the reference's shortcut (eval/coco.py:11-13): the detections pass through untouched.
    CPU record tensors (the gloo protocol tests) take the same path without streams."""

    def __init__(self, group=None, depth: int = 2, force: bool = False):
        self.group, self.depth, self.force = group, max(1, int(depth)), force
        self._slots = {}            # (shape, device) -> list of [rec, out, done_event, claimed, generation] per slot
        self._next = 0
        self._side = None

    def _active(self):
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force)

    def _slot(self, shape, device, dtype):
        key = (tuple(shape), str(device))
        slots = self._slots.get(key)
        if slots is None:
            world = dist.get_world_size(self.group)
            slots = []
            for _ in range(self.depth):
                rec = torch.empty(shape, device=device, dtype=dtype)
                out = torch.empty((world * shape[0],) + tuple(shape[1:]), device=device, dtype=dtype)
                slots.append([rec, out, torch.cuda.Event() if device.type == "cuda" else None, False, -1])
            self._slots[key] = slots
        i = self._next % self.depth
        self._next += 1
        slots[i][4] = self._next                         # generation: which submission owns the slot now
        return slots[i]

    def _claim(self, shape, device, dtype):
        """Next slot; the caller's stream first waits for the slot's previous gather (it is about to overwrite its source)."""
        # (depth = 1 is valid for strictly sequential use — submit, result, submit ...; collecting a result behind a later submit is what
        #  the generation check of result_records() catches, at any depth)
        slot = self._slot(shape, device, dtype)
        if slot[2] is not None and slot[3]:
            torch.cuda.current_stream(device).wait_event(slot[2])
        slot[3] = True
        return slot

    def submit_records(self, rec: torch.Tensor, _slot=None):
        """Start the all-gather of a packed record [N_local,k,R]; returns a handle for result_records()."""
        if not self._active():
            return ("local", rec)
        slot = _slot
        if slot is None:
            slot = self._claim(rec.shape, rec.device, rec.dtype)
            slot[0].copy_(rec)
        if rec.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(rec.device)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(rec.device))
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                dist.all_gather_into_tensor(slot[1], slot[0], group=self.group)
                slot[2].record(self._side)
        else:
            dist.all_gather_into_tensor(slot[1], slot[0], group=self.group)
        return ("gathered", slot, slot[4])

    def result_records(self, handle) -> torch.Tensor:
        """The gathered records of a submit: a VIEW of the persistent slot (not a copy), valid until `depth` further submits re-claim
        the slot — a handle whose slot has been re-claimed raises instead of silently returning a later batch."""
        kind, payload = handle[0], handle[1]
        if kind == "local":
            return payload
        if payload[4] != handle[2]:
            raise RuntimeError("Collator: the slot of this handle was re-claimed by a later submit (results must be collected at most "
                               "depth - 1 submissions behind; raise depth for a deeper pipeline)")
        if payload[2] is not None:
            torch.cuda.current_stream(payload[1].device).wait_event(payload[2])
        return payload[1]

    def submit(self, dets: dict):
        if not self._active():
            return ("dets", dets)
        box_key = "bboxes" if "bboxes" in dets else "boxes"
        scores, emb = dets["scores"], dets.get("embeddings")
        N, k = scores.shape
        slot = self._claim((N, k, RECORD_FIELDS + (emb.shape[-1] if emb is not None else 0)), scores.device, torch.float32)
        pack_detections(dets, out=slot[0])                       # straight into the persistent record
        return ("records", self.submit_records(slot[0], _slot=slot), box_key)

    def result(self, handle) -> dict:
        if handle[0] == "dets":
            return handle[1]
        return unpack_detections(self.result_records(handle[1]), box_key=handle[2])
