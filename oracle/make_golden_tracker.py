"""ORACLE — TEST INFRASTRUCTURE ONLY.  Generates tests/golden/track_*.npz by RUNNING THE REFERENCE'S OWN tracking code in
the build container (it cannot travel to the GPU box):

  * box cost matrices: box_iou_distance_matrix / box_giou_distance_matrix (centernet_lightning/utils/box.py:84-92)
  * association:       Tracker.update, Track, match_with_threshold (centernet_lightning/models/tracker.py:27-43, 123-347)

Import notes.  tracker.py:12 does `from ..utils import box_iou_distance_matrix, box_giou_distance_matrix, load_config`, but the
reference's utils/__init__.py has those exports commented out, so the module does not import as shipped.  This generator
imports the real utils/box.py and binds its two functions (and a dummy load_config) onto the real `centernet_lightning.utils`
package before importing tracker.py; `filterpy` (absent) is stubbed — the Kalman branch is not exercised (use_kalman=False).

Run:  python oracle/make_golden_tracker.py     (only where /root/reference exists)
Fixtures are data only: the seeded recipe of the inputs (oracle/tracker_ref.synth_sequence) + SHA-256 of the input bytes +
the reference's outputs; small KAT inputs are stored in full.
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipes                      # noqa: E402
import tracker_ref                  # noqa: E402
from _ref_import import import_reference_centernet, _stub, _Any   # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def import_reference_tracker():
    import_reference_centernet()                       # installs the third-party stubs + sys.path
    fp = _stub("filterpy")
    fp.kalman = _stub("filterpy.kalman", KalmanFilter=_Any)
    utils = importlib.import_module("centernet_lightning.utils")
    box = importlib.import_module("centernet_lightning.utils.box")
    utils.box_iou_distance_matrix = box.box_iou_distance_matrix
    utils.box_giou_distance_matrix = box.box_giou_distance_matrix
    utils.load_config = lambda *a, **k: {}
    trk = importlib.import_module("centernet_lightning.models.tracker")
    return trk, box


def run_reference(trk, seq, **kw):
    tracker = trk.Tracker(model=None, **kw)
    ids, boxes, n_tracks = [], [], []
    for bboxes, labels, scores, emb in seq:
        tracker.update(bboxes, labels, scores, emb)
        tracker.frame += 1
        ids.append(np.array([t.track_id for t in tracker.tracks if t.active], np.int64))
        boxes.append(np.array([t.bbox for t in tracker.tracks if t.active], np.float32).reshape(-1, 4))
        n_tracks.append(len(tracker.tracks))
    final_emb = np.stack([t.embedding for t in tracker.tracks]) if tracker.tracks else np.zeros((0, 1), np.float32)
    final_ids = np.array([t.track_id for t in tracker.tracks], np.int64)
    return ids, boxes, np.array(n_tracks), final_ids, final_emb


def run_oracle(seq, **kw):
    tracker = tracker_ref.Tracker(**kw)
    ids, boxes, n_tracks = [], [], []
    for bboxes, labels, scores, emb in seq:
        tracker.update(bboxes, labels, scores, emb)
        i, b = tracker.active()
        ids.append(np.array(i, np.int64))
        boxes.append(np.array(b, np.float32).reshape(-1, 4))
        n_tracks.append(len(tracker.tracks))
    final_emb = np.stack([t.embedding for t in tracker.tracks]) if tracker.tracks else np.zeros((0, 1), np.float32)
    final_ids = np.array([t.track_id for t in tracker.tracks], np.int64)
    return ids, boxes, np.array(n_tracks), final_ids, final_emb


def pack_ragged(prefix, arrays, payload):
    payload[f"{prefix}_len"] = np.array([len(a) for a in arrays], np.int64)
    payload[f"{prefix}_cat"] = np.concatenate(arrays, axis=0) if arrays else np.zeros((0,))


def main():
    import warnings
    warnings.simplefilter("ignore")
    os.makedirs(OUT, exist_ok=True)
    trk, box = import_reference_tracker()

    # ---------------- box cost matrices (utils/box.py) ----------------
    rng = np.random.default_rng(5)
    def rand_boxes(n):
        c = rng.random((n, 2)); s = rng.random((n, 2)) * 0.3 + 0.01
        return np.concatenate([c - s / 2, c + s / 2], axis=1).astype(np.float32)
    b1, b2 = rand_boxes(37), rand_boxes(23)
    b2[3] = b1[5]                                    # identical boxes -> distance 0
    b2[4] = [0.1, 0.1, 0.1, 0.1]                     # degenerate (zero area)
    b1[6] = [0.1, 0.1, 0.1, 0.1]                     # degenerate vs degenerate -> 0/0 = nan
    b1[7] = [0.9, 0.9, 0.2, 0.2]                     # inverted corners (negative extents)
    with np.errstate(all="ignore"):
        iou_d = box.box_iou_distance_matrix(b1, b2)
        giou_d = box.box_giou_distance_matrix(b1, b2)
        assert np.array_equal(iou_d, tracker_ref.box_iou_distance_matrix(b1, b2), equal_nan=True)
        assert np.array_equal(giou_d, tracker_ref.box_giou_distance_matrix(b1, b2), equal_nan=True)
    assert iou_d.dtype == np.float32
    np.savez_compressed(os.path.join(OUT, "track_boxcost.npz"), b1=b1, b2=b2, iou=iou_d, giou=giou_d)
    print("track_boxcost:", iou_d.shape, "nan:", int(np.isnan(iou_d).sum()), int(np.isnan(giou_d).sum()))

    # ---------------- match_with_threshold (tracker.py:27-43) ----------------
    cost = rng.random((9, 6))
    m, ur, uc = trk.match_with_threshold(cost, 0.35)
    assert (m, ur, uc) == tracker_ref.match_with_threshold(cost, 0.35)
    np.savez_compressed(os.path.join(OUT, "track_match.npz"), cost=cost, threshold=0.35, matches=np.array(m, np.int64),
                        unmatched_rows=np.array(ur, np.int64), unmatched_cols=np.array(uc, np.int64))

    # ---------------- whole association sequences (Tracker.update) ----------------
    cases = [
        # name, seed, recipe kwargs, tracker kwargs
        ("seq_iou", 0, dict(frames=24, objects=12, k=48), dict(box_cost="iou")),
        ("seq_giou", 1, dict(frames=24, objects=10, k=40), dict(box_cost="giou", box_threshold=0.7, smoothing_factor=0.3)),
        ("seq_reid_only", 2, dict(frames=16, objects=8, k=32), dict(box_cost=None, reid_threshold=0.25, min_birth_age=1)),
        ("seq_unsorted", 3, dict(frames=12, objects=8, k=32, sort_scores=False), dict(box_cost="iou", max_inactive_age=3)),
        ("seq_k300", 4, dict(frames=10, objects=40, k=300), dict(box_cost="iou", detection_threshold=0.4)),
    ]
    for name, seed, rk, tk in cases:
        seq = tracker_ref.synth_sequence(seed, **rk)
        ids, boxes, n_tracks, fin_ids, fin_emb = run_reference(trk, seq, **tk)
        o_ids, o_boxes, o_n, o_fin_ids, o_fin_emb = run_oracle(seq, **tk)
        assert all(np.array_equal(a, b) for a, b in zip(ids, o_ids)), name
        assert all(np.array_equal(a, b) for a, b in zip(boxes, o_boxes)), name
        assert np.array_equal(n_tracks, o_n) and np.array_equal(fin_ids, o_fin_ids), name
        assert np.array_equal(fin_emb, o_fin_emb), name
        payload = dict(seed=seed, recipe=np.array(repr(sorted(rk.items()))), tracker=np.array(repr(sorted(tk.items(), key=str))),
                       sha=recipes.sha256(*[a for fr in seq for a in fr]), n_tracks=n_tracks,
                       final_ids=fin_ids, final_emb=fin_emb.astype(np.float32))
        assert fin_emb.dtype == np.float32, fin_emb.dtype
        pack_ragged("ids", ids, payload)
        pack_ragged("boxes", boxes, payload)
        np.savez_compressed(os.path.join(OUT, f"track_{name}.npz"), **payload)
        print(f"track_{name}: frames={len(seq)} tracks/frame={n_tracks.tolist()} active(last)={ids[-1].tolist()[:10]} "
              f"next_id={int(fin_ids.max()) + 1 if len(fin_ids) else 0}")


if __name__ == "__main__":
    main()
