"""Wire / on-disk formats around the hot path: what the reference's evaluation and export flows exchange
with the detector, produced from this package's device-resident outputs.

  to_coco_predictions          <- CenterNet.validation_step (models/centernet.py:204-209): xyxy -> xywh, numpy, list per image
  coco_annotations             <- CocoEvaluator.create_coco's annotation records (eval/coco.py:78-97)
  detections_to_coco_results   <- eval/utils.py:83-103 (COCO "results" json)
  mot_challenge_lines / write_mot_challenge <- eval/mot_challenge.py:59-79 (gt.txt / tracker .txt lines, 1-based, xywh)
  load_checkpoint              <- CenterNet.load_from_checkpoint as used by tools/export.py:8,15: Lightning `.ckpt` -> this model

Only the box conversion is device work (cnl_boxes_xyxy_to_xywh_f32); the rest is host-side record building, kept identical to
the reference's so that its evaluators (pycocotools / TrackEval — not in this image) read the same bytes.
"""
import ctypes
import json

import numpy as np
import torch

from . import _lib


def boxes_xyxy_to_xywh(boxes: torch.Tensor) -> torch.Tensor:
    """torchvision box_convert(boxes, 'xyxy', 'xywh') (centernet.py:207) on the HIP device; any leading shape, last dim 4."""
    if not (isinstance(boxes, torch.Tensor) and boxes.is_cuda):
        raise RuntimeError("boxes_xyxy_to_xywh: expected a HIP ('cuda') tensor; there is no CPU fallback in the product path")
    if boxes.dtype != torch.float32 or boxes.shape[-1] != 4:
        raise ValueError(f"boxes_xyxy_to_xywh: expected float32 [..., 4], got {boxes.dtype} {tuple(boxes.shape)}")
    b = boxes.contiguous()
    out = torch.empty_like(b)
    with torch.cuda.device(b.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(b.device).cuda_stream)
        _lib.check(_lib.load().cnl_boxes_xyxy_to_xywh_f32(b.data_ptr(), out.data_ptr(), b.numel() // 4, stream), "cnl_boxes_xyxy_to_xywh_f32")
    return out


def to_coco_predictions(dets: dict):
    """{"bboxes"|"boxes" [N,k,4] xyxy, "scores" [N,k], "labels" [N,k]} (device) -> list of N dicts {"boxes" xywh, "scores",
    "labels"} of numpy arrays: exactly what validation_step hands to CocoEvaluator.update (centernet.py:207-212)."""
    boxes = dets["bboxes"] if "bboxes" in dets else dets["boxes"]
    preds = {"boxes": boxes_xyxy_to_xywh(boxes), "scores": dets["scores"], "labels": dets["labels"]}
    preds = {k: v.cpu().numpy() for k, v in preds.items()}
    return [{k: v[i] for k, v in preds.items()} for i in range(boxes.shape[0])]


def coco_annotations(detections, image_ids, prediction=False):
    """The annotation records CocoEvaluator.create_coco builds (eval/coco.py:80-97): ids start at 1, area = w*h, iscrowd 0,
    `score` only for predictions."""
    annotations = []
    ann_id = 1
    for img_id, det in zip(image_ids, detections):
        det = {k: np.asarray(v).tolist() for k, v in det.items()}
        for i, (box, label) in enumerate(zip(det["boxes"], det["labels"])):
            ann = {"id": ann_id, "image_id": img_id, "category_id": label, "bbox": box, "area": box[2] * box[3], "iscrowd": 0}
            if prediction:
                ann["score"] = det["scores"][i]
            annotations.append(ann)
            ann_id += 1
    return annotations


def detections_to_coco_results(image_ids, bboxes, labels, scores, save_path=None, score_threshold=0):
    """eval/utils.py:83-103.  numpy inputs are converted to plain Python numbers (the reference's json.dump only accepts those)."""
    results = []
    for img_id, img_bboxes, img_labels, img_scores in zip(image_ids, bboxes, labels, scores):
        for box, label, score in zip(np.asarray(img_bboxes).tolist(), np.asarray(img_labels).tolist(), np.asarray(img_scores).tolist()):
            if score < score_threshold:
                continue
            results.append({"image_id": img_id, "category_id": int(label), "bbox": box, "score": score})
    if save_path is not None:
        with open(save_path, "w") as f:
            json.dump(results, f)
    return results


def mot_challenge_lines(bboxes, track_ids, ground_truth=False):
    """eval/mot_challenge.py:59-64 (gt.txt) / :73-79 (tracker file): one line per (frame, track), 1-based frame / id / x / y,
    boxes xywh.  bboxes: per frame, a sequence of boxes; track_ids: per frame, a sequence of ids."""
    tail = "1,1,1" if ground_truth else "1,-1,-1,-1"
    lines = []
    for i, (frame_bboxes, frame_track_ids) in enumerate(zip(bboxes, track_ids)):
        for box, track_id in zip(frame_bboxes, frame_track_ids):
            lines.append(f"{i+1},{track_id+1},{box[0]+1},{box[1]+1},{box[2]},{box[3]},{tail}\n")
    return lines


def write_mot_challenge(path, bboxes, track_ids, ground_truth=False):
    with open(path, "w") as f:
        f.writelines(mot_challenge_lines(bboxes, track_ids, ground_truth))


_PREFIXES = ("model.", "module.", "net.")
_RENAMES = (("output_heads.", "heads."),)


# training-only entries of a reference checkpoint that the inference model has no counterpart for: EmbeddingHead.classifier (the
# track-id classification layers "used during training only", models/fairmot.py:25-31: Linear / BatchNorm1d / ReLU / Linear ->
# classifier.{0,1,3}.*) and whatever the loss modules register (loss_function.*)
_TRAINING_ONLY = (".classifier.", ".loss_function.")


def is_training_only_key(key):
    return any(t in "." + key for t in _TRAINING_ONLY)


def checkpoint_state_dict(ckpt, keep_training_only=False):
    """Lightning `.ckpt` (dict with "state_dict") or a bare state_dict -> tensors keyed like this package's CenterNet:
    wrapper prefixes dropped ("model." of GenericModel inside the LightningModule, models/meta.py:66; "module." of DDP),
    Gen-A's `output_heads.` -> `heads.`; training-only entries (see _TRAINING_ONLY) dropped unless asked for."""
    sd = ckpt.get("state_dict", ckpt) if isinstance(ckpt, dict) else ckpt
    out = {}
    for k, v in sd.items():
        if not keep_training_only and is_training_only_key(k):
            continue
        changed = True
        while changed:
            changed = False
            for p in _PREFIXES:
                if k.startswith(p):
                    k, changed = k[len(p):], True
        for a, b in _RENAMES:
            if k.startswith(a):
                k = b + k[len(a):]
        out[k] = v
    return out


def load_checkpoint(model, ckpt, strict=True, map_location="cpu"):
    """CenterNet.load_from_checkpoint's weight-loading half (tools/export.py:8): `ckpt` is a path or an already-loaded dict.
    Returns the (missing, unexpected) key lists; with strict=True a mismatch raises, listing both.  Training-only entries of a
    reference tracking checkpoint (heads.reid.classifier.*, fairmot.py:25-31) are not "unexpected": they are skipped."""
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
        ckpt = torch.load(ckpt, map_location=map_location, weights_only=True)
    sd = checkpoint_state_dict(ckpt)
    own = model.state_dict()
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    if strict and (missing or unexpected):
        raise KeyError(f"checkpoint does not match the model: {len(missing)} missing (e.g. {missing[:3]}), "
                       f"{len(unexpected)} unexpected (e.g. {unexpected[:3]})")
    model.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    return missing, unexpected
