"""Wire formats: the product's record builders against golden output of the reference's own
detections_to_coco_results / CocoEvaluator.create_coco / evaluate_mot_tracking_sequence (oracle/make_golden_formats.py), the
checkpoint key mapping, and — on the GPU — the xyxy -> xywh kernel and to_coco_predictions."""
import json
import os

import numpy as np
import pytest
import torch

import centernet_lightning_amd as cl
from centernet_lightning_amd import formats

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "formats.json")


@pytest.fixture(scope="module")
def g():
    return json.load(open(GOLDEN))


def test_coco_results_match_reference(g, tmp_path):
    i = g["inputs"]
    c = g["coco_results"]
    path = tmp_path / "r.json"
    res = formats.detections_to_coco_results(c["image_ids"], i["boxes_xywh"], i["labels"], i["scores"], str(path), c["score_threshold"])
    assert res == c["results"]
    assert path.read_text() == c["file_text"]
    # numpy inputs (what to_coco_predictions yields) give the same records
    res_np = formats.detections_to_coco_results(c["image_ids"], np.array(i["boxes_xywh"]), np.array(i["labels"]), np.array(i["scores"]),
                                                None, c["score_threshold"])
    assert res_np == c["results"]


def test_coco_annotations_match_reference(g):
    i = g["inputs"]
    b, s, l = np.array(i["boxes_xywh"], np.float32), np.array(i["scores"], np.float32), np.array(i["labels"], np.int64)
    preds = [{"boxes": b[n], "scores": s[n], "labels": l[n]} for n in range(len(b))]
    assert formats.coco_annotations(preds, [0, 1, 2], prediction=True) == g["coco_annotations"]["prediction"]
    assert formats.coco_annotations([{"boxes": b[n], "labels": l[n]} for n in range(len(b))], [0, 1, 2]) == g["coco_annotations"]["target"]


def test_mot_challenge_lines_match_reference(g, tmp_path):
    m = g["mot"]
    assert "".join(formats.mot_challenge_lines(m["pred_bboxes"], m["pred_track_ids"])) == m["pred_txt"]
    assert "".join(formats.mot_challenge_lines(m["target_bboxes"], m["target_track_ids"], ground_truth=True)) == m["gt_txt"]
    f32 = [np.asarray(b, np.float32).reshape(-1, 4) for b in m["pred_bboxes"]]
    assert "".join(formats.mot_challenge_lines(f32, m["pred_track_ids"])) == m["pred_txt_float32"]
    p = tmp_path / "seq.txt"
    formats.write_mot_challenge(str(p), m["pred_bboxes"], m["pred_track_ids"])
    assert p.read_text() == m["pred_txt"]


def test_checkpoint_key_mapping(tmp_path):
    cfg = {"backbone": {"name": "resnet34"}, "neck": {"name": "fpn"}, "output_heads": {"heatmap": {"num_classes": 3}, "box_2d": {}}}
    torch.manual_seed(0)
    src = cl.CenterNet(cfg["backbone"], cfg["neck"], cfg["output_heads"], "detection")
    dst = cl.CenterNet(cfg["backbone"], cfg["neck"], cfg["output_heads"], "detection")
    sd = src.state_dict()
    # a Lightning checkpoint of the reference: LightningModule.model = GenericModel -> "model." prefix (models/meta.py:66);
    # Gen-A names the heads container `output_heads`
    ckpt = {"epoch": 3, "state_dict": {"model." + k.replace("heads.", "output_heads.", 1): v for k, v in sd.items()}, "hyper_parameters": {}}
    path = tmp_path / "last.ckpt"
    torch.save(ckpt, path)
    missing, unexpected = formats.load_checkpoint(dst, str(path))
    assert missing == [] and unexpected == []
    for k, v in dst.state_dict().items():
        assert torch.equal(v, sd[k]), k
    mapped = formats.checkpoint_state_dict({"module.model.backbone.conv1.weight": 1, "output_heads.heatmap.out_conv.bias": 2})
    assert set(mapped) == {"backbone.conv1.weight", "heads.heatmap.out_conv.bias"}
    bad = dict(ckpt["state_dict"])
    bad.pop("model.backbone.conv1.weight")
    bad["model.extra.weight"] = torch.zeros(1)
    with pytest.raises(KeyError):
        formats.load_checkpoint(dst, {"state_dict": bad})
    missing, unexpected = formats.load_checkpoint(dst, {"state_dict": bad}, strict=False)
    assert missing == ["backbone.conv1.weight"] and unexpected == ["extra.weight"]


def test_no_cpu_fallback_for_device_work():
    with pytest.raises(RuntimeError):
        formats.boxes_xyxy_to_xywh(torch.zeros(2, 4))


@pytest.mark.gpu
def test_xyxy_to_xywh_and_coco_predictions_gpu():
    g = torch.Generator().manual_seed(1)
    b = torch.rand(3, 7, 4, generator=g) * 300
    b[..., 2:] += b[..., :2]
    ref = torch.stack([b[..., 0], b[..., 1], b[..., 2] - b[..., 0], b[..., 3] - b[..., 1]], dim=-1)    # box_convert xyxy -> xywh
    out = formats.boxes_xyxy_to_xywh(b.cuda())
    assert torch.equal(out.cpu(), ref)
    assert formats.boxes_xyxy_to_xywh(torch.empty(0, 4, device="cuda")).shape == (0, 4)
    dets = {"bboxes": b.cuda(), "scores": torch.rand(3, 7, generator=g).cuda(), "labels": torch.randint(0, 80, (3, 7), generator=g).cuda()}
    preds = formats.to_coco_predictions(dets)
    assert len(preds) == 3 and set(preds[0]) == {"boxes", "scores", "labels"}
    assert isinstance(preds[1]["boxes"], np.ndarray) and np.array_equal(preds[1]["boxes"], ref[1].numpy())
    assert preds[2]["labels"].dtype == np.int64 and np.array_equal(preds[2]["scores"], dets["scores"][2].cpu().numpy())
    ann = formats.coco_annotations(preds, [0, 1, 2], prediction=True)
    assert len(ann) == 21 and ann[0]["id"] == 1 and ann[-1]["image_id"] == 2 and isinstance(ann[0]["bbox"], list)


def test_reference_tracking_checkpoint_with_training_only_keys_loads_strict():
    """a reference FairMOT checkpoint carries EmbeddingHead.classifier.{0,1,3}.* ("used during training only",
    fairmot.py:25-31).  They are dropped, not reported as unexpected, so strict loading works; genuinely foreign keys still raise."""
    import os
    import torch
    import centernet_lightning_amd as cl
    from centernet_lightning_amd import formats
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "centernet-lightning_amd", "configs", "tracking_resnet34_fpn.yaml")
    model = cl.build_centernet(cfg)
    sd = {"model." + k: v.clone() for k, v in model.state_dict().items()}
    E, ids = 64, 800                                         # the reference EmbeddingHead's key set (max_track_ids = 800 in the config)
    sd.update({"model.heads.reid.classifier.0.weight": torch.zeros(E, E), "model.heads.reid.classifier.1.weight": torch.ones(E),
               "model.heads.reid.classifier.1.bias": torch.zeros(E), "model.heads.reid.classifier.1.running_mean": torch.zeros(E),
               "model.heads.reid.classifier.1.running_var": torch.ones(E), "model.heads.reid.classifier.1.num_batches_tracked": torch.tensor(0),
               "model.heads.reid.classifier.3.weight": torch.zeros(ids, E), "model.heads.reid.classifier.3.bias": torch.zeros(ids)})
    sd["model.heads.reid.out_conv.bias"] = sd["model.heads.reid.out_conv.bias"] + 1.0
    missing, unexpected = formats.load_checkpoint(model, {"state_dict": sd}, strict=True)
    assert missing == [] and unexpected == []
    assert float(model.heads["reid"].out_conv.bias[0]) == 1.0
    assert any(formats.is_training_only_key(k) for k in sd) and "heads.reid.classifier.0.weight" in formats.checkpoint_state_dict(sd, keep_training_only=True)
    sd["model.some_other_module.weight"] = torch.zeros(1)
    import pytest
    with pytest.raises(KeyError):
        formats.load_checkpoint(model, {"state_dict": sd}, strict=True)
