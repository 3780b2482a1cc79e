"""CPU: the tracker oracle (oracle/tracker_ref.py) against the golden vectors produced by the reference's own
Tracker.update / box distance functions (oracle/make_golden_tracker.py), plus the product's host-side tracker logic that needs
no GPU."""
import ast
import glob
import os
import warnings

import numpy as np
import pytest

import recipes
import tracker_ref

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
SEQS = sorted(glob.glob(os.path.join(GOLDEN, "track_seq_*.npz")))


def load_case(path):
    g = dict(np.load(path))
    rk = dict(ast.literal_eval(str(g["recipe"])))
    tk = dict(ast.literal_eval(str(g["tracker"])))
    seq = tracker_ref.synth_sequence(int(g["seed"]), **rk)
    assert recipes.sha256(*[a for fr in seq for a in fr]) == str(g["sha"]), "input recipe drifted"
    return g, seq, tk


def unpack(g, prefix):
    lens = g[f"{prefix}_len"]
    cat = g[f"{prefix}_cat"]
    out, o = [], 0
    for n in lens:
        out.append(cat[o:o + n])
        o += n
    return out


def test_box_cost_matrices_bit_exact():
    g = np.load(os.path.join(GOLDEN, "track_boxcost.npz"))
    iou = tracker_ref.box_iou_distance_matrix(g["b1"], g["b2"])
    giou = tracker_ref.box_giou_distance_matrix(g["b1"], g["b2"])
    assert iou.dtype == np.float32 and giou.dtype == np.float32
    assert np.array_equal(iou, g["iou"], equal_nan=True)
    assert np.array_equal(giou, g["giou"], equal_nan=True)
    assert np.isnan(g["iou"]).sum() == 1                      # the degenerate-vs-degenerate pair is 0/0 in the reference too
    assert g["iou"][5, 3] == 0.0                              # identical boxes


def test_match_with_threshold_golden():
    g = np.load(os.path.join(GOLDEN, "track_match.npz"))
    m, ur, uc = tracker_ref.match_with_threshold(g["cost"], float(g["threshold"]))
    assert np.array_equal(np.array(m), g["matches"])
    assert ur == g["unmatched_rows"].tolist() and uc == g["unmatched_cols"].tolist()
    from centernet_lightning_amd.tracker import match_with_threshold          # the product's host copy of the same step
    assert match_with_threshold(g["cost"], float(g["threshold"])) == (m, ur, uc)


@pytest.mark.parametrize("path", SEQS, ids=lambda p: os.path.basename(p)[6:-4])
def test_sequences_match_reference(path):
    g, seq, tk = load_case(path)
    trk = tracker_ref.Tracker(**tk)
    ids, boxes, n_tracks = [], [], []
    for fr in seq:
        trk.update(*fr)
        i, b = trk.active()
        ids.append(np.array(i, np.int64))
        boxes.append(np.array(b, np.float32).reshape(-1, 4))
        n_tracks.append(len(trk.tracks))
    assert np.array_equal(np.array(n_tracks), g["n_tracks"])
    for a, b in zip(ids, unpack(g, "ids")):
        assert np.array_equal(a, b)
    for a, b in zip(boxes, unpack(g, "boxes")):
        assert np.array_equal(a, b.reshape(-1, 4))
    assert np.array_equal(np.array([t.track_id for t in trk.tracks]), g["final_ids"])
    assert np.array_equal(np.stack([t.embedding for t in trk.tracks]), g["final_emb"])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree only exists in the build container")
def test_oracle_against_live_reference():
    import make_golden_tracker as mg
    trk_mod, box_mod = mg.import_reference_tracker()
    rng = np.random.default_rng(99)
    c = rng.random((50, 2)); s = rng.random((50, 2)) * 0.4
    b = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
    with np.errstate(all="ignore"):
        assert np.array_equal(box_mod.box_iou_distance_matrix(b[:30], b[20:]), tracker_ref.box_iou_distance_matrix(b[:30], b[20:]), equal_nan=True)
        assert np.array_equal(box_mod.box_giou_distance_matrix(b[:30], b[20:]), tracker_ref.box_giou_distance_matrix(b[:30], b[20:]), equal_nan=True)
    seq = tracker_ref.synth_sequence(77, frames=10, objects=9, k=30)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r_ids, r_boxes, r_n, r_fin, r_emb = mg.run_reference(trk_mod, seq, box_cost="giou")
    o_ids, o_boxes, o_n, o_fin, o_emb = mg.run_oracle(seq, box_cost="giou")
    assert all(np.array_equal(x, y) for x, y in zip(r_ids, o_ids)) and np.array_equal(r_n, o_n)
    assert np.array_equal(r_emb, o_emb)


def test_product_tracker_host_surface(configs_dir):
    import centernet_lightning_amd as cl
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t = cl.build_tracker(os.path.join(configs_dir, "tracking_resnet34_fpn.yaml"))
        assert (t.num_detections, t.detection_threshold, t.reid_threshold, t.box_cost, t.box_threshold) == (300, 0.3, 0.2, "iou", 0.5)
        assert (t.smoothing_factor, t.max_inactive_age, t.min_birth_age, t.frame, t.next_track_id, t.tracks) == (0.5, 30, 2, 0, 0, [])
        assert cl.Tracker(use_kalman=True).use_kalman is True          # tracker.py:243-262: host-side filter (BoxKalman)
        with pytest.raises(ValueError):
            cl.Tracker(reid_cost="minkowski")                # no gfx950 kernel: needs allow_host_cost=True
        for name in ("euclidean", "sqeuclidean", "cityblock", "chebyshev", "canberra", "braycurtis"):      # these have one
            assert cl.Tracker(reid_cost=name).reid_cost == name and cl.Tracker(reid_cost=name)._host_reid is None
        assert cl.Tracker(reid_cost="minkowski", allow_host_cost=True)._host_reid == "minkowski"
        with pytest.raises(ValueError):
            cl.Tracker(box_cost="diou")
        import torch
        if not torch.cuda.is_available():                      # no CPU fallback: the association needs the HIP device
            with pytest.raises(RuntimeError):
                cl.Tracker().update(np.zeros((2, 4), np.float32), np.zeros(2, np.int64), np.ones(2, np.float32), np.ones((2, 8), np.float32))
    # life cycle of the host record (tracker.py:295-347)
    tr = cl.Track(None, 0, np.zeros(4), 0, min_birth_age=2, max_inactive_age=2)
    assert tr.state == cl.TrackState.UNCONFIRMED and not tr.confirmed
    tr.update_matched(np.ones(4)); assert tr.state == cl.TrackState.UNCONFIRMED
    tr.update_matched(np.ones(4)); assert tr.active
    tr.update_unmatched(); assert tr.state == cl.TrackState.INACTIVE and tr.inactive_age == 0
    tr.update_unmatched(); assert tr.inactive_age == 1
    tr.update_matched(np.ones(4)); assert tr.active and tr.inactive_age == 0
    tr.update_unmatched(); tr.update_unmatched(); tr.update_unmatched(); assert tr.to_delete
    t2 = cl.Track(None, 1, np.zeros(4), 0)
    t2.update_unmatched(); assert t2.to_delete


def test_box_kalman_known_answers():
    """BoxKalman (filterpy restated): a measurement equal to the state leaves the state; repeated measurements of a box moving at constant
    velocity make the velocity estimate converge to it; the covariance stays symmetric positive definite; predict moves the corners by
    the velocities."""
    box = np.array([0.2, 0.3, 0.4, 0.6])
    kf = tracker_ref.BoxKalman(box)
    assert np.allclose(kf.update(box), box) and np.allclose(kf.x[4:], 0)
    v = np.array([0.01, -0.005, 0.01, -0.005])
    kf = tracker_ref.BoxKalman(box)
    for t in range(1, 60):
        kf.predict()
        kf.update(box + v * t)
    assert np.allclose(kf.x[4:], v, atol=2e-4) and np.allclose(kf.x[:4], box + v * 59, atol=2e-3)
    assert np.allclose(kf.P, kf.P.T) and np.all(np.linalg.eigvalsh(kf.P) > 0)
    before = kf.x.copy()
    kf.predict()
    assert np.allclose(kf.x[:4], before[:4] + before[4:]) and np.allclose(kf.x[4:], before[4:])
