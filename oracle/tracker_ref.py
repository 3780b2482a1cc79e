"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product package.

CPU (numpy / scipy) restatement of the reference's tracking association step:

  * box_inter_union / iou / giou distance matrices <- centernet_lightning/utils/box.py:49-92
  * cosine cost matrix                            <- scipy.spatial.distance.cdist(metric="cosine") as bound at
                                                     centernet_lightning/models/tracker.py:55 (third-party; scipy is
                                                     present in the image, so it is called, not restated)
  * match_with_threshold                          <- centernet_lightning/models/tracker.py:27-43
  * Tracker.update                                <- centernet_lightning/models/tracker.py:123-201
  * Track state machine + embedding smoothing     <- centernet_lightning/models/tracker.py:217-347 (use_kalman=False)

Parity status: PINNED.  tests/golden/track_*.npz hold cost matrices and per-frame track ids / boxes produced by the
reference's own `Tracker.update`, `box_iou_distance_matrix`, `box_giou_distance_matrix` (imported in the build container by
oracle/make_golden_tracker.py); tests/test_oracle_tracker.py checks this restatement against them.

The Kalman option (`use_kalman=True`): filterpy is absent from the image, so its KalmanFilter cannot be run to pin this part —
BoxKalman restates filterpy's published predict / update equations with the reference's matrices and noise schedule
(tracker.py:243-262, 281-323): "parity unpinned" for the filter; known-answer checks in tests/test_oracle_tracker.py.

Reference quirk kept on purpose (results must be identical): matches index the *thresholded* detection arrays, but
`update_matched` is fed `bboxes[det_idx]` / `embeddings[det_idx]` from the UNfiltered arrays (tracker.py:171).  The two agree
whenever the scores are sorted descending (always true for gather_tracking2d output).
"""
from enum import Enum, auto

import numpy as np
from scipy.optimize import linear_sum_assignment
from scipy.spatial import distance


# ---------------------------------------------------------------- cost matrices
def box_inter_union_matrix(b1, b2):
    """utils/box.py:49-62. b1 (n,4), b2 (m,4) x1y1x2y2 -> inter, union (n,m), in the arrays' own dtype."""
    area1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    area2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    lt = np.maximum(b1[:, None, :2], b2[:, :2])
    rb = np.minimum(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clip(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = area1[:, None] + area2 - inter
    return inter, union


def box_iou_distance_matrix(b1, b2):
    """utils/box.py:65-68, 84-87: 1 - inter/union."""
    inter, union = box_inter_union_matrix(b1, b2)
    with np.errstate(invalid="ignore", divide="ignore"):
        return 1 - inter / union


def box_giou_distance_matrix(b1, b2):
    """utils/box.py:71-81, 90-92: 1 - (iou - (hull - union)/hull)."""
    inter, union = box_inter_union_matrix(b1, b2)
    with np.errstate(invalid="ignore", divide="ignore"):
        iou = inter / union
        lti = np.minimum(b1[:, None, :2], b2[:, :2])
        rbi = np.maximum(b1[:, None, 2:], b2[:, 2:])
        whi = (rbi - lti).clip(min=0)
        areai = whi[:, :, 0] * whi[:, :, 1]
        return 1 - (iou - (areai - union) / areai)


BOX_COSTS = {"iou": box_iou_distance_matrix, "giou": box_giou_distance_matrix}


def cosine_distance_matrix(a, b):
    """tracker.py:55: partial(scipy cdist, metric="cosine") -> float64 (n,m)."""
    return distance.cdist(a, b, metric="cosine")


def match_with_threshold(cost, threshold):
    """tracker.py:27-43."""
    rows, cols = linear_sum_assignment(cost)
    matches, mr, mc = [], set(), set()
    for r, c in zip(rows, cols):
        if cost[r, c] < threshold:
            matches.append((int(r), int(c)))
            mr.add(int(r))
            mc.add(int(c))
    return (matches, [x for x in range(cost.shape[0]) if x not in mr], [x for x in range(cost.shape[1]) if x not in mc])


# ---------------------------------------------------------------- track objects
class TrackState(Enum):
    UNCONFIRMED = auto()
    ACTIVE = auto()
    INACTIVE = auto()
    TO_DELETE = auto()


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
    """tracker.py:217-347."""

    def __init__(self, track_id, bbox, label, embedding, min_birth_age=2, max_inactive_age=30, smoothing_factor=0.9, use_kalman=False):
        self.kf = BoxKalman(bbox) if use_kalman else None
        self.track_id = track_id
        self.state = TrackState.UNCONFIRMED
        self.birth_age = 0
        self.inactive_age = 0
        self.bbox = bbox
        self.label = label
        self.embedding = embedding / np.linalg.norm(embedding)
        self.min_birth_age = min_birth_age
        self.max_inactive_age = max_inactive_age
        self.smoothing_factor = smoothing_factor

    active = property(lambda self: self.state == TrackState.ACTIVE)
    to_delete = property(lambda self: self.state == TrackState.TO_DELETE)

    def update_matched(self, bbox, embedding):
        if self.state == TrackState.UNCONFIRMED:
            self.birth_age += 1
            if self.birth_age >= self.min_birth_age:
                self.state = TrackState.ACTIVE
        elif self.state == TrackState.INACTIVE:
            self.state = TrackState.ACTIVE
            self.inactive_age = 0
        self.bbox = bbox if self.kf is None else self.kf.update(bbox)
        embedding = embedding / np.linalg.norm(embedding)
        self.embedding = (1 - self.smoothing_factor) * self.embedding + self.smoothing_factor * embedding

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


class Tracker:
    """tracker.py:45-201 (`update` only — the model-driven `step_batch` lives in the product)."""

    def __init__(self, detection_threshold=0.3, reid_cost="cosine", reid_threshold=0.2, box_cost="iou", box_threshold=0.5,
                 smoothing_factor=0.5, max_inactive_age=30, min_birth_age=2, use_kalman=False):
        # tracker.py:51, 62-64: a scipy cdist metric name, or a callable (det_embeddings, track_embeddings) -> matrix
        self.reid_cost = reid_cost if callable(reid_cost) else (lambda a, b, m=reid_cost: distance.cdist(a, b, metric=m))
        self.use_kalman = use_kalman
        self.detection_threshold = detection_threshold
        self.reid_threshold = reid_threshold
        self.box_cost = box_cost if callable(box_cost) else (BOX_COSTS[box_cost] if box_cost is not None else None)
        self.box_threshold = box_threshold
        self.smoothing_factor = smoothing_factor
        self.max_inactive_age = max_inactive_age
        self.min_birth_age = min_birth_age
        self.frame = 0
        self.next_track_id = 0
        self.tracks = []
        self.last_costs = None          # (reid_cost_matrix, full box cost matrix) of the last update, for the tests

    def update(self, bboxes, labels, scores, embeddings):
        mask = scores >= self.detection_threshold
        det_bboxes, det_labels, det_embeddings = bboxes[mask], labels[mask], embeddings[mask]
        self.last_costs = None
        if len(self.tracks) == 0:
            unmatched_dets = range(len(det_bboxes))
        else:
            trk_emb = np.stack([t.embedding for t in self.tracks], axis=0)
            trk_box = np.stack([t.bbox for t in self.tracks], axis=0)
            reid = self.reid_cost(det_embeddings, trk_emb)
            matches, unmatched_dets, unmatched_tracks = match_with_threshold(reid, self.reid_threshold)
            full_box = None
            if self.box_cost is not None:
                full_box = self.box_cost(det_bboxes, trk_box)       # kept for the tests only (element-wise: sub == full[ix_])
                sub = self.box_cost(det_bboxes[unmatched_dets], trk_box[unmatched_tracks])      # tracker.py:157-160
                new_matches, ud, ut = match_with_threshold(sub, self.box_threshold)
                matches.extend((unmatched_dets[x], unmatched_tracks[y]) for x, y in new_matches)
                unmatched_dets, unmatched_tracks = [unmatched_dets[x] for x in ud], [unmatched_tracks[y] for y in ut]
            self.last_costs = (reid, full_box)
            for d, t in matches:
                self.tracks[t].update_matched(bboxes[d], embeddings[d])     # unfiltered arrays: reference quirk (header)
            for t in unmatched_tracks:
                self.tracks[t].update_unmatched()
        for d in unmatched_dets:
            self.tracks.append(Track(self.next_track_id, det_bboxes[d], det_labels[d], det_embeddings[d],
                                     min_birth_age=self.min_birth_age, max_inactive_age=self.max_inactive_age,
                                     smoothing_factor=self.smoothing_factor, use_kalman=self.use_kalman))
            self.next_track_id += 1
        self.tracks = [t for t in self.tracks if not t.to_delete]
        for t in self.tracks:                                   # tracker.py:199-201
            if t.kf is not None:
                t.kf.predict()
        self.frame += 1

    def active(self):
        return ([t.track_id for t in self.tracks if t.active],
                [np.asarray(t.bbox) for t in self.tracks if t.active])


# ---------------------------------------------------------------- seeded synthetic sequences
def synth_sequence(seed, frames=24, objects=12, k=48, emb_dim=64, sort_scores=True):
    """A small MOT-like scene: `objects` boxes drifting across a unit image with fixed identity embeddings plus noise;
    each frame yields k detections (real ones with high scores, clutter with low scores), normalised boxes as in
    tracker.py:104.  Objects blink (missed detections) so that INACTIVE / re-activation / deletion paths run.
    Returns a list of (bboxes (k,4) f32, labels (k,) i64, scores (k,) f32, embeddings (k,E) f32)."""
    rng = np.random.default_rng(seed)
    centre = rng.random((objects, 2)) * 0.6 + 0.2
    vel = (rng.random((objects, 2)) - 0.5) * 0.02
    size = rng.random((objects, 2)) * 0.08 + 0.05
    ident = rng.standard_normal((objects, emb_dim))
    ident[1] = ident[0] + 0.35 * rng.standard_normal(emb_dim)       # two look-alikes: the box stage must split them
    out = []
    for f in range(frames):
        centre = centre + vel
        present = rng.random(objects) > 0.15
        if f in (5, 6, 7):
            present[2] = False                                      # a longer gap
        boxes, scores, embs = [], [], []
        for o in range(objects):
            if not present[o]:
                continue
            c = centre[o] + rng.standard_normal(2) * 0.002
            s = size[o] * (1 + rng.standard_normal(2) * 0.02)
            boxes.append([c[0] - s[0] / 2, c[1] - s[1] / 2, c[0] + s[0] / 2, c[1] + s[1] / 2])
            scores.append(0.45 + 0.5 * rng.random())
            noise = 0.25 if (o == 3 and f % 4 == 0) else 0.08      # object 3 sometimes looks different -> IoU fallback
            embs.append(ident[o] * (1.0 + 0.1 * rng.random()) + noise * 6 * rng.standard_normal(emb_dim) * (noise > 0.2)
                        + 0.08 * rng.standard_normal(emb_dim))
        n_real = len(boxes)
        for _ in range(k - n_real):                                 # clutter
            c = rng.random(2)
            s = rng.random(2) * 0.05 + 0.01
            boxes.append([c[0] - s[0] / 2, c[1] - s[1] / 2, c[0] + s[0] / 2, c[1] + s[1] / 2])
            scores.append(0.29 * rng.random() if rng.random() > 0.04 else 0.31 + 0.1 * rng.random())
            embs.append(rng.standard_normal(emb_dim))
        boxes = np.asarray(boxes, np.float32)
        scores = np.asarray(scores, np.float32)
        embs = np.asarray(embs, np.float32)
        order = np.argsort(-scores, kind="stable") if sort_scores else rng.permutation(k)
        out.append((boxes[order], np.zeros(k, np.int64), scores[order], embs[order]))
    return out
