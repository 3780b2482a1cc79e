"""ORACLE — TEST INFRASTRUCTURE ONLY.  Generates tests/golden/formats.json by RUNNING THE REFERENCE'S OWN record builders in the
build container:

  * detections_to_coco_results                 centernet_lightning/eval/utils.py:83-103
  * CocoEvaluator.create_coco (annotations)    centernet_lightning/eval/coco.py:78-107   (pycocotools.COCO stubbed: only the
                                               `dataset` dict the reference assembles is read back)
  * evaluate_mot_tracking_sequence             centernet_lightning/eval/mot_challenge.py:9-83 (trackeval stubbed; the gt.txt and
                                               tracker file the reference WRITES are read back before its temp dir vanishes)
  * validation_step's xyxy -> xywh + split     centernet_lightning/models/centernet.py:207-209 needs torchvision.ops.box_convert
                                               (absent): its published definition (x1, y1, x2-x1, y2-y1) is restated -> "parity
                                               unpinned" for that one line.

Run:  python oracle/make_golden_formats.py     (only where /root/reference exists)
"""
import importlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import import_reference_centernet, _stub, _Any   # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "formats.json")


class _COCO:
    def createIndex(self):
        pass


def main():
    import_reference_centernet()
    sys.modules["pycocotools.coco"].COCO = _COCO
    te = _stub("trackeval")
    te.datasets = _stub("trackeval.datasets", MotChallenge2DBox=_Any)
    te.metrics = _stub("trackeval.metrics", HOTA=_Any, CLEAR=_Any, Identity=_Any)
    utils = importlib.import_module("centernet_lightning.eval.utils")
    coco = importlib.import_module("centernet_lightning.eval.coco")
    mot = importlib.import_module("centernet_lightning.eval.mot_challenge")
    coco.COCO = _COCO

    rng = np.random.default_rng(12)
    N, k = 3, 5
    xy = rng.random((N, k, 2)) * 400
    wh = rng.random((N, k, 2)) * 100 + 1
    boxes_xywh = np.concatenate([xy, wh], -1).astype(np.float32)
    scores = np.sort(rng.random((N, k)).astype(np.float32), axis=1)[:, ::-1].copy()
    labels = rng.integers(0, 80, (N, k)).astype(np.int64)
    out = {"inputs": {"boxes_xywh": boxes_xywh.tolist(), "scores": scores.tolist(), "labels": labels.tolist()}}

    # eval/utils.py:83-103 (plain Python numbers in, as its json.dump requires)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "results.json")
        res = utils.detections_to_coco_results([10, 11, 12], boxes_xywh.tolist(), labels.tolist(), scores.tolist(), path, score_threshold=0.3)
        out["coco_results"] = {"image_ids": [10, 11, 12], "score_threshold": 0.3, "results": res, "file_text": open(path).read()}

    # eval/coco.py:78-107
    preds = [{"boxes": boxes_xywh[i], "scores": scores[i], "labels": labels[i]} for i in range(N)]
    c = coco.CocoEvaluator.create_coco(preds, [0, 1, 2], 80, prediction=True)
    t = coco.CocoEvaluator.create_coco([{"boxes": boxes_xywh[i], "labels": labels[i]} for i in range(N)], [0, 1, 2], 80, prediction=False)
    out["coco_annotations"] = {"prediction": c.dataset["annotations"], "target": t.dataset["annotations"]}

    # eval/mot_challenge.py:9-83: capture what it writes
    captured = {}

    def fake_eval(gt_folder, trackers_folder, trackers_to_eval=None, seqmap_file=None, skip_split_fol=False, **kw):
        captured["gt"] = open(os.path.join(gt_folder, "sequence_0", "gt", "gt.txt")).read()
        captured["pred"] = open(os.path.join(trackers_folder, trackers_to_eval[0], "data", "sequence_0.txt")).read()
        return {"tracker_0": {"sequence_0": {"HOTA": np.array([0.5]), "MOTA": 0.1, "IDF1": 0.2}}}

    mot.evaluate_mot_tracking_from_file = fake_eval
    frames = 4
    pred_b = [[[float(v) for v in rng.random(4) * 50] for _ in range(int(n))] for n in (2, 0, 3, 1)]
    pred_i = [[int(v) for v in rng.integers(0, 9, len(b))] for b in pred_b]
    tgt_b = [[[float(v) for v in rng.random(4) * 50] for _ in range(2)] for _ in range(frames)]
    tgt_i = [[0, 1] for _ in range(frames)]
    # numpy float32 boxes too: the f-string prints numpy scalars with their own repr rules
    pred_b_np = [np.asarray(b, np.float32).reshape(-1, 4) for b in pred_b]
    mot.evaluate_mot_tracking_sequence(pred_b, pred_i, tgt_b, tgt_i)
    out["mot"] = {"pred_bboxes": pred_b, "pred_track_ids": pred_i, "target_bboxes": tgt_b, "target_track_ids": tgt_i,
                  "gt_txt": captured["gt"], "pred_txt": captured["pred"]}
    mot.evaluate_mot_tracking_sequence(pred_b_np, pred_i, tgt_b, tgt_i)
    out["mot"]["pred_txt_float32"] = captured["pred"]

    with open(OUT, "w") as f:
        json.dump(out, f, indent=0)
    print("saved", OUT, os.path.getsize(OUT), "bytes;", len(res), "results,", len(captured["pred"].splitlines()), "mot lines")
    print(captured["pred"].splitlines()[0])


if __name__ == "__main__":
    main()
