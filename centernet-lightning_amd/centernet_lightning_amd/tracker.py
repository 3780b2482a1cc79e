"""Tracker — host side of the tracking association step.

Mirrors the reference's `Tracker` / `Track` / `match_with_threshold` / `build_tracker`
(centernet_lightning/models/tracker.py:27-43, 45-201, 217-353): same constructor arguments, `step_batch`, `step_single`,
`update`, `reset`, same track life cycle, same outputs ({"bboxes": [...], "track_ids": [...]} per frame).

What moved to the GPU (csrc/track.hip): the detection-threshold mask, the cosine (re-ID) and IoU / GIoU cost matrices, and the
track table itself — per-track embedding and box live in HBM and are updated there.  Per frame only the n x T cost matrices
come to the host, where the Hungarian assignment runs on scipy exactly as in the reference (tracker.py:28), and the short
match list goes back.  The reference instead copies every frame's k x (6+E) detections to the host (tracker.py:107).

`use_kalman=True` (tracker.py:243-262, 281-323): the per-track 8-state filter is a float64 numpy restatement of filterpy's
KalmanFilter (third-party, absent) on the host, beside the life cycle; the filtered boxes are uploaded to the device table.
Per frame: ONE kernel launch (cnl_track_frame_f32) writes the frame record — kept-detection indices, boxes / scores / labels, cost
matrices — straight into mapped host memory, ONE stream synchronisation makes it readable; no copy operation in either direction.
`reid_cost`: "cosine" / "euclidean" / "sqeuclidean" / "cityblock" / "chebyshev" / "canberra" / "braycurtis" / "correlation" have kernels; any other scipy
cdist name or a callable, and a callable `box_cost`
(tracker.py:51, 62-64), are computed on the host from copies only with `allow_host_cost=True` (otherwise the constructor raises).
There is no CPU fallback for the device path: without the HIP library or a GPU, `update` raises.
"""
import contextlib
import ctypes
import warnings
import weakref
from enum import Enum, auto
from typing import List

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from . import _lib
from .config import load_config

_BOX_MODES = {None: 0, "iou": 1, "giou": 2}
_LABEL_KINDS = {torch.int64: 1, torch.int32: 2, torch.float32: 3}     # det_label element types cnl_track_frame_f32 reads
_REID_METRICS = {"cosine": 0, "euclidean": 1, "sqeuclidean": 2, "cityblock": 3, "chebyshev": 4, "canberra": 5, "braycurtis": 6, "correlation": 7}
# ^ the scipy cdist metrics with a gfx950 kernel (float64, scipy's operation order); "manhattan" etc. are scipy aliases -> host path


_NO_GUARD = contextlib.nullcontext()


def _on(dev):
    """Device guard for the launches below — skipped when `dev` is the current device already (torch.cuda.device() costs ~6 us per use,
    twice per frame)."""
    return _NO_GUARD if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


class _Mapped:
    """Page-locked host memory that the device addresses through the same pointer (cnl_host_alloc): the frame record the association
    kernel writes and the index lists the table update reads cross PCIe as the kernels' own stores / loads — no copy operation, and one
    stream synchronisation per frame.  `np` is a uint8 view of the whole block (valid while this object lives)."""

    def __init__(self, nbytes):
        lib = _lib.load()
        p = ctypes.c_void_p()
        _lib.check(lib.cnl_host_alloc(nbytes, ctypes.byref(p)), "cnl_host_alloc")
        self.ptr, self.nbytes = p.value, nbytes
        self.np = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(self.ptr))
        self._finalizer = weakref.finalize(self, lib.cnl_host_free, ctypes.c_void_p(self.ptr))
        self._finalizer.atexit = False          # at interpreter exit the HIP runtime may be gone already; the process frees the pages


class TrackState(Enum):
    UNCONFIRMED = auto()
    ACTIVE = auto()
    INACTIVE = auto()
    TO_DELETE = auto()


def match_with_threshold(cost_matrix, threshold):
    """tracker.py:27-43: optimal assignment, keeping only pairs with cost < threshold.  Same result and order as the reference's loop
    (matches in row order, unmatched rows / columns ascending), vectorised: the loop over sets cost more than the Hungarian step itself."""
    row_ind, col_ind = linear_sum_assignment(cost_matrix)
    keep = cost_matrix[row_ind, col_ind] < threshold
    rows, cols = row_ind[keep], col_ind[keep]
    free_r = np.ones(cost_matrix.shape[0], dtype=bool)
    free_c = np.ones(cost_matrix.shape[1], dtype=bool)
    free_r[rows] = False
    free_c[cols] = False
    return list(zip(rows.tolist(), cols.tolist())), np.flatnonzero(free_r).tolist(), np.flatnonzero(free_c).tolist()


class BoxKalman:
    """The 8-state constant-velocity Kalman filter the reference builds per track with filterpy (tracker.py:243-262, 281-301, 317-323):
    state = box corners x1 y1 x2 y2 + their velocities, measurement = the corners.  filterpy is third-party and absent from the image;
    its published predict / update equations (filterpy/kalman/kalman_filter.py: x = Fx, P = FPF' + Q;  y = z - Hx, S = HPH' + R,
    K = PH'S^-1, x += Ky, P = (I-KH)P(I-KH)' + KRK') are restated in float64 numpy — "parity unpinned" (no reference test pins it)."""

    def __init__(self, bbox):
        self.x = np.zeros(8)
        self.x[:4] = bbox
        self.F = np.eye(8)
        self.F[:4, 4:] = np.eye(4)
        self.H = np.eye(4, 8)
        wh = np.asarray(bbox[2:], np.float64) - np.asarray(bbox[:2], np.float64)
        std = np.tile(wh, 4)                                  # adapted from DeepSORT (tracker.py:256-260)
        std[:4] /= 10
        std[4:] /= 16
        self.P = np.diag(std ** 2)

    def predict(self):
        wh = self.x[2:4] - self.x[:2]
        std = np.tile(wh, 4)                                  # tracker.py:284-289
        std[:4] /= 20
        std[4:] /= 160
        self.x = self.F @ self.x
        self.P = self.F @ self.P @ self.F.T + np.diag(np.square(std))

    def update(self, z):
        wh = self.x[2:4] - self.x[:2]
        R = np.diag((np.tile(wh, 2) / 20) ** 2)               # tracker.py:318-320
        y = np.asarray(z, np.float64) - self.H @ self.x
        PHT = self.P @ self.H.T
        S = self.H @ PHT + R
        K = PHT @ np.linalg.inv(S)
        self.x = self.x + K @ y
        I_KH = np.eye(8) - K @ self.H
        self.P = I_KH @ self.P @ I_KH.T + K @ R @ K.T
        return self.x[:4].copy()


class Track:
    """Host record of one track (tracker.py:217-347).  bbox / label live here (they are reported every frame); the embedding
    lives in the tracker's device table and is fetched on access."""

    def __init__(self, tracker, track_id, bbox, label, min_birth_age=2, max_inactive_age=30, smoothing_factor=0.9, use_kalman=False):
        self._tracker = tracker
        self.kf = BoxKalman(bbox) if use_kalman else None
        self._row = -1
        self.track_id = track_id
        self.state = TrackState.UNCONFIRMED
        self.birth_age = 0
        self.inactive_age = 0
        self.bbox = bbox
        self.label = label
        self.min_birth_age = min_birth_age
        self.max_inactive_age = max_inactive_age
        self.smoothing_factor = smoothing_factor

    @property
    def active(self):
        return self.state == TrackState.ACTIVE

    @property
    def confirmed(self):
        return self.state != TrackState.UNCONFIRMED

    @property
    def to_delete(self):
        return self.state == TrackState.TO_DELETE

    @property
    def embedding(self):
        return self._tracker._emb[self._row].cpu().numpy()

    def update_matched(self, bbox):
        if self.state == TrackState.UNCONFIRMED:
            self.birth_age += 1
            if self.birth_age >= self.min_birth_age:
                self.state = TrackState.ACTIVE
        elif self.state == TrackState.INACTIVE:
            self.state = TrackState.ACTIVE
            self.inactive_age = 0
        # tracker.py:311-323: the detection's box, or the filtered state when the track carries a Kalman filter
        self.bbox = bbox if self.kf is None else self.kf.update(bbox)

    def kalman_predict(self):
        """tracker.py:281-290 (called at the end of every Tracker.update; `bbox` keeps the last UPDATED state, as in the reference,
        where it is a view of the array that filterpy's predict replaces)."""
        if self.kf is not None:
            self.kf.predict()

    def update_unmatched(self):
        if self.state == TrackState.UNCONFIRMED:
            self.state = TrackState.TO_DELETE
        elif self.state == TrackState.ACTIVE:
            self.state = TrackState.INACTIVE
            self.inactive_age = 0
        elif self.state == TrackState.INACTIVE:
            self.inactive_age += 1
            if self.inactive_age >= self.max_inactive_age:
                self.state = TrackState.TO_DELETE

    def __repr__(self):
        return f"track id: {self.track_id}, bbox: {self.bbox}, label: {self.label}, state: {self.state.name}"


class Tracker:
    """Multiple-object tracking on top of `CenterNet.gather_tracking2d` (tracker.py:45-201)."""

    def __init__(self, model=None, nms_kernel=3, num_detections=300, detection_threshold=0.3, reid_cost="cosine",
                 reid_threshold=0.2, box_cost="iou", box_threshold=0.5, smoothing_factor=0.5, use_kalman=False,
                 max_inactive_age=30, min_birth_age=2, device=None, allow_host_cost=False):
        """reid_cost: "cosine" (default), "euclidean", "sqeuclidean", "cityblock", "chebyshev", "canberra", "braycurtis", "correlation" run on the device.  The reference accepts ANY scipy cdist metric name or
        a callable (tracker.py:51, 62-64), and a callable box_cost: those are computed on the HOST from copies of the frame's kept embeddings /
        boxes and the track table (two more device -> host copies per frame) — only with allow_host_cost=True, otherwise they raise: a silent
        CPU detour is not what a caller of a gfx950 tracker expects."""
        self.model = model
        if model is None:
            warnings.warn("A model was not provided. Only `.update()` will work")
        self._host_reid = None if reid_cost in _REID_METRICS else reid_cost
        self._host_box = box_cost if callable(box_cost) else None
        if (self._host_reid is not None or self._host_box is not None) and not allow_host_cost:
            raise ValueError(f"reid_cost={reid_cost!r} / box_cost={box_cost!r}: only {sorted(_REID_METRICS)} and 'iou' / 'giou' / None have gfx950 "
                             "kernels; pass allow_host_cost=True to compute other scipy metrics or callables on the host (slower: the embeddings "
                             "then travel to the host every frame)")
        if self._host_box is None and box_cost not in _BOX_MODES:
            raise ValueError(f"box_cost={box_cost!r}: expected 'iou', 'giou', None or (with allow_host_cost=True) a callable")
        self.nms_kernel = nms_kernel
        self.num_detections = num_detections
        self.detection_threshold = detection_threshold
        self.reid_cost = reid_cost
        self.reid_threshold = reid_threshold
        self.box_cost = box_cost
        self.box_threshold = box_threshold
        self.smoothing_factor = smoothing_factor
        self.use_kalman = bool(use_kalman)
        self.max_inactive_age = max_inactive_age
        self.min_birth_age = min_birth_age
        self._device = torch.device(device) if device is not None else None
        self.reset()

    # ------------------------------------------------------------------ state
    def reset(self):
        self.frame = 0
        self.next_track_id = 0
        self.tracks: List[Track] = []
        self._emb = None            # device track table [capacity, E] / [capacity, 4]; rows 0..len(tracks)-1 are live
        self._box = None
        self._spare = None          # the other half of the ping-pong pair
        self.last_costs = None      # (reid [n,T] f64, box [n,T] f32 | None, det_index [n]) of the last update (host numpy)
        self._rec = None            # mapped host memory the association kernel writes the frame record into (cnl_track_frame_f32)
        self._src = None            # mapped host memory holding the two index lists cnl_track_apply_f32 reads
        self._apply_stream = None   # the stream the last cnl_track_apply_f32 was launched on

    @property
    def device(self):
        if self._device is None:
            if self.model is not None:
                self._device = next(self.model.parameters()).device
            else:
                self._device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        if self._device.type != "cuda":
            raise RuntimeError("the tracker's association kernels need a HIP device ('cuda'); there is no CPU fallback")
        return self._device

    def _tables(self, rows, E):
        """Return (new_emb, new_box) with room for `rows` rows, distinct from the live table."""
        dev = self.device
        if self._spare is None or self._spare[0].shape[0] < rows or self._spare[0].shape[1] != E:
            cap = max(64, 1 << (max(rows, 1) - 1).bit_length())
            self._spare = (torch.empty((cap, E), device=dev, dtype=torch.float32), torch.empty((cap, 4), device=dev, dtype=torch.float32))
        return self._spare

    # ------------------------------------------------------------------ inference
    @torch.no_grad()
    def step_batch(self, images: torch.Tensor, **kwargs):
        """Run the model on a batch of consecutive frames and update the tracks frame by frame (tracker.py:84-121).
        Returns {"bboxes": [...], "track_ids": [...]} with one list per frame (active tracks only)."""
        nms_kernel = kwargs.get("nms_kernel", self.nms_kernel)
        num_detections = kwargs.get("num_detections", self.num_detections)
        self.model.eval()
        images = images.to(self.device)
        heatmap, box_2d, reid = self.model(images)
        det = self.model.gather_tracking2d(heatmap, box_2d, reid, nms_kernel=nms_kernel, num_detections=num_detections,
                                           normalize_bbox=True)
        # the kernel that computes a frame's costs also writes its boxes / scores / labels into the frame record: no separate copy
        out = {"bboxes": [], "track_ids": []}
        for i in range(images.shape[0]):
            self._update_device(det["bboxes"][i], det["scores"][i], det["embeddings"][i], det["labels"][i], None, **kwargs)
            self.frame += 1
            out["bboxes"].append([x.bbox for x in self.tracks if x.active])
            out["track_ids"].append([x.track_id for x in self.tracks if x.active])
        return out

    @torch.no_grad()
    def step_single(self, img: torch.Tensor, **kwargs):
        out = self.step_batch(img.unsqueeze(0), **kwargs)
        return {k: v[0] for k, v in out.items()}

    def update(self, bboxes, labels, scores, embeddings, **kwargs):
        """Update current tracks with one frame's detections (tracker.py:123-201).  Accepts numpy arrays (as the reference) or
        torch tensors; the arrays are moved to the HIP device, where the association costs are computed."""
        dev = self.device

        def to_dev(a):
            if isinstance(a, torch.Tensor) and a.device == dev and a.dtype == torch.float32 and a.is_contiguous():
                return a                                   # already where the kernels read it (each .to() costs ~7 us of host time)
            return torch.as_tensor(a).to(device=dev, dtype=torch.float32).contiguous()
        host = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        if all(isinstance(a, torch.Tensor) and a.is_cuda for a in (bboxes, labels, scores)) and bboxes.dim() == 2:
            # device inputs: the three small arrays the host-side life cycle reads come back inside the frame record
            self._update_device(to_dev(bboxes), to_dev(scores), to_dev(embeddings), labels.to(dev), None, **kwargs)
        else:
            self._update_device(to_dev(bboxes), to_dev(scores), to_dev(embeddings), None, (host(bboxes), host(labels), host(scores)), **kwargs)

    # ------------------------------------------------------------------ one frame
    def _update_device(self, d_box, d_score, d_emb, d_label, host, **kwargs):
        """`host`: (boxes, labels, scores) as numpy arrays when the caller holds them on the host already; None: the frame record
        brings them (d_label: the labels on the device)."""
        detection_threshold = kwargs.get("detection_threshold", self.detection_threshold)
        reid_threshold = kwargs.get("reid_threshold", self.reid_threshold)
        box_threshold = kwargs.get("box_threshold", self.box_threshold)
        lib = _lib.load()
        dev = self.device
        k, E = d_emb.shape
        if d_box.shape != (k, 4) or d_score.shape != (k,):
            raise ValueError(f"detections: boxes {tuple(d_box.shape)}, scores {tuple(d_score.shape)}, embeddings {tuple(d_emb.shape)}")
        d_box, d_score, d_emb = d_box.contiguous(), d_score.contiguous(), d_emb.contiguous()
        T = len(self.tracks)
        box_mode = 0 if self._host_box is not None else _BOX_MODES[self.box_cost]
        reid_metric = _REID_METRICS.get(self.reid_cost, 0)
        with_dets = host is None
        label_kind = 0
        if with_dets and d_label is not None:
            if d_label.shape != (k,):
                raise ValueError(f"detections: labels {tuple(d_label.shape)}, expected ({k},)")
            label_kind = _LABEL_KINDS.get(d_label.dtype, 0)
            if not label_kind:
                d_label, label_kind = d_label.to(torch.int64), 1
            d_label = d_label.contiguous()
        # cnl_track_frame_bytes(k, T, with_dets), in Python (a ctypes call costs ~2 us of a 150 us frame)
        need = ((((32 + 4 * k + 7) & ~7) + (24 * k if with_dets else 0) + 7) & ~7) + 12 * k * T
        with _on(dev):
            cur = torch.cuda.current_stream(dev)
            stream = ctypes.c_void_p(cur.cuda_stream)
            if self._apply_stream is not None and self._apply_stream != cur:
                # the caller changed streams between frames: the previous frame's table update (which reads the mapped index lists and
                # writes the tables this frame reads) ran on another stream — finish it first (same stream: stream order does it)
                self._apply_stream.synchronize()
            if self._rec is None or self._rec.nbytes < need:
                self._rec = _Mapped(max(2 * need, 1 << 18))          # persistent: nothing allocated per frame
            rec = self._rec
            # ONE launch writes the record [header | det_index | boxes scores labels | reid f64 n*T | box f32 n*T] straight into host
            # memory (packed by the n the kernel finds), ONE synchronisation makes it readable: the frame's only device -> host traffic
            _lib.check(lib.cnl_track_frame_f32(d_emb.data_ptr(), d_box.data_ptr(), d_score.data_ptr(), d_label.data_ptr() if label_kind else None,
                                               label_kind, k, E, float(detection_threshold), self._emb.data_ptr() if T else None,
                                               self._box.data_ptr() if T else None, T, box_mode, reid_metric, int(with_dets), rec.ptr, rec.nbytes,
                                               stream), "cnl_track_frame_f32")
            cur.synchronize()
        h = rec.np
        hdr = h[:32].view(np.int32)
        n, off_dets, off_reid, off_box = int(hdr[0]), int(hdr[5]), int(hdr[6]), int(hdr[7])
        if with_dets:
            # copies: a Track keeps its box, and the record is overwritten by the next frame
            f = h[off_dets:off_dets + 20 * k].view(np.float32)
            h_box, h_score = f[:4 * k].reshape(k, 4).copy(), f[4 * k:].copy()
            h_label = h[off_dets + 20 * k:off_dets + 24 * k].view(np.int32).astype(np.int64)
        else:
            h_box, h_label, h_score = host
            n_host = int(np.count_nonzero(np.asarray(h_score, dtype=np.float32) >= np.float32(detection_threshold)))
            if n_host != n:
                raise RuntimeError(f"detection count mismatch between host ({n_host}) and device ({n}): scores on the host and on the device differ")
        det_index = h[32:32 + 4 * n].view(np.int32).copy()
        self.d2h_bytes = 32 + 4 * n + (24 * k if with_dets else 0) + (12 if box_mode else 8) * n * T      # bytes the kernel stored over PCIe
        self.last_costs = None

        # ---- assignment on the host: tracker.py:139-176 ----
        if T == 0:
            matches, unmatched_dets, unmatched_tracks = [], list(range(n)), []
        else:
            reid = h[off_reid:off_reid + 8 * n * T].view(np.float64).reshape(n, T)
            if self._host_reid is not None or self._host_box is not None:
                # opt-in host fallback (allow_host_cost=True): the reference's own expressions on copies of the operands
                from scipy.spatial.distance import cdist
                sel = torch.from_numpy(np.ascontiguousarray(det_index[:n]).astype(np.int64)).to(dev)
                if self._host_reid is not None:
                    h_de, h_te = d_emb.index_select(0, sel).cpu().numpy(), self._emb[:T].cpu().numpy()
                    reid = self._host_reid(h_de, h_te) if callable(self._host_reid) else cdist(h_de, h_te, self._host_reid)
                    reid = np.asarray(reid, np.float64).reshape(n, T)
            matches, unmatched_dets, unmatched_tracks = match_with_threshold(reid, reid_threshold)
            box = None
            if self._host_box is not None:
                # tracker.py:157-162 as written: the callable sees the REMAINING detections' and tracks' boxes
                h_db, h_tb = d_box.index_select(0, sel).cpu().numpy(), self._box[:T].cpu().numpy()
                sub = np.asarray(self._host_box(h_db[unmatched_dets], h_tb[unmatched_tracks])).reshape(len(unmatched_dets), len(unmatched_tracks))
                new_matches, ud, ut = match_with_threshold(sub, box_threshold)
                matches.extend((unmatched_dets[x], unmatched_tracks[y]) for x, y in new_matches)
                unmatched_dets, unmatched_tracks = [unmatched_dets[x] for x in ud], [unmatched_tracks[y] for y in ut]
            elif box_mode:
                box = h[off_box:off_box + 4 * n * T].view(np.float32).reshape(n, T)
            if box is not None:
                # element-wise costs: the remaining-pairs matrix of tracker.py:157-162 is a sub-matrix of the full one
                sub = box[np.ix_(unmatched_dets, unmatched_tracks)]
                new_matches, ud, ut = match_with_threshold(sub, box_threshold)
                matches.extend((unmatched_dets[x], unmatched_tracks[y]) for x, y in new_matches)
                unmatched_dets, unmatched_tracks = [unmatched_dets[x] for x in ud], [unmatched_tracks[y] for y in ut]
            self.last_costs = (reid.copy(), None if box is None else box.copy(), det_index.copy())

        # ---- track life cycle (host) + the rows of the new device table ----
        row_det = {}                                  # old track index -> detection row feeding its update
        for det_idx, track_idx in matches:
            # reference quirk kept (tracker.py:171): the match indexes the thresholded arrays but the update reads the
            # unfiltered ones at the same position; identical when scores are sorted descending (gather_tracking2d output)
            self.tracks[track_idx].update_matched(h_box[det_idx])
            row_det[track_idx] = det_idx
        for track_idx in unmatched_tracks:
            self.tracks[track_idx].update_unmatched()
        old_rows = list(range(T))
        for det_idx in unmatched_dets:
            src = int(det_index[det_idx])
            self.tracks.append(Track(self, self.next_track_id, h_box[src], h_label[src], min_birth_age=self.min_birth_age,
                                     max_inactive_age=self.max_inactive_age, smoothing_factor=self.smoothing_factor,
                                     use_kalman=self.use_kalman))
            self.next_track_id += 1
            old_rows.append(-1)
            row_det[len(self.tracks) - 1] = src
        keep = [i for i, t in enumerate(self.tracks) if not t.to_delete]
        self.tracks = [self.tracks[i] for i in keep]
        T_new = len(self.tracks)
        if T_new:
            src = np.empty((2, T_new), np.int32)
            src[0] = [old_rows[i] for i in keep]
            src[1] = [row_det.get(i, -1) for i in keep]
            with _on(dev):
                # the two index lists sit in mapped host memory that the kernel reads directly (2 x T_new int32 over PCIe): no host ->
                # device copy; the synchronisation at the top of the next frame orders the kernel's reads before the next overwrite
                if self._src is None or self._src.nbytes < 8 * T_new:
                    self._src = _Mapped(8 * max(256, 1 << (T_new - 1).bit_length()))
                cap = self._src.nbytes // 8
                self._src.np.view(np.int32).reshape(2, cap)[:, :T_new] = src
                d_src = (self._src.ptr, self._src.ptr + 4 * cap)
                new_emb, new_box = self._tables(T_new, E)
                _lib.check(lib.cnl_track_apply_f32(self._emb.data_ptr() if T else None, self._box.data_ptr() if T else None,
                                                   d_emb.data_ptr(), d_box.data_ptr(), d_src[0], d_src[1],
                                                   T_new, E, float(self.smoothing_factor), new_emb.data_ptr(), new_box.data_ptr(),
                                                   stream), "cnl_track_apply_f32")
                self._apply_stream = cur
            self._spare, (self._emb, self._box) = ((self._emb, self._box) if self._emb is not None else None), (new_emb, new_box)
        for r, t in enumerate(self.tracks):
            t._row = r
        if self.use_kalman and T_new:
            # the device table's boxes are the detections' (cnl_track_apply_f32); a Kalman track's box is its filtered state, computed on
            # the host with the rest of the life cycle (8x8 float64 algebra per track): upload the T x 4 boxes (16 B per track)
            boxes = np.asarray([np.asarray(t.bbox, np.float64) for t in self.tracks], np.float32).reshape(T_new, 4)
            with torch.cuda.device(dev):
                self._box[:T_new].copy_(torch.from_numpy(boxes), non_blocking=False)
            for t in self.tracks:
                t.kalman_predict()

    def track_embeddings(self):
        """Device view [T, E] of the live track table (row order = self.tracks)."""
        return self._emb[:len(self.tracks)] if self._emb is not None else None


def build_tracker(config, model=None):
    """tracker.py:349-353."""
    if isinstance(config, str):
        config = load_config(config)["tracker"]
    return Tracker(model=model, **config)
