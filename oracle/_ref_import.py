"""TEST INFRASTRUCTURE ONLY (oracle). Imports the reference's decode code from /root/reference
with stub modules for its absent third-party dependencies. Works only in the build container
(/root/reference does not exist on the GPU box); used by oracle/make_golden.py and by the
`-m "not gpu"` test that pins oracle/decode_ref.py against the live reference when present.

Nothing here is imported by the product package.
"""
import sys
import types
from types import SimpleNamespace

REF_ROOT = "/root/reference"


class _Any:
    """Permissive placeholder class for third-party symbols the decode path never executes."""
    def __init__(self, *a, **k):
        pass


class _StubModule(types.ModuleType):
    def __getattr__(self, item):          # any other attribute -> placeholder class
        if item.startswith("__"):
            raise AttributeError(item)
        return _Any


def _stub(name, **attrs):
    m = _StubModule(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference_centernet():
    """Return the reference `CenterNet` class (centernet_lightning/models/centernet.py:68)."""
    import os
    if not os.path.isdir(REF_ROOT):
        raise FileNotFoundError(REF_ROOT)
    import torch
    from torch import nn

    tv = _stub("torchvision")
    tv.ops = _stub("torchvision.ops", box_convert=lambda *a, **k: None, batched_nms=lambda *a, **k: None,
                   DeformConv2d=_Any)
    pl = _stub("pytorch_lightning", LightningModule=nn.Module, Callback=object, Trainer=_Any)
    pl.callbacks = _stub("pytorch_lightning.callbacks", Callback=object)
    pl.loggers = _stub("pytorch_lightning.loggers", WandbLogger=_Any, TensorBoardLogger=_Any)
    vt = _stub("vision_toolbox")
    vt.backbones = _stub("vision_toolbox.backbones", BaseBackbone=nn.Module)
    vt.necks = _stub("vision_toolbox.necks", BaseNeck=nn.Module)
    vt.components = _stub("vision_toolbox.components", ConvBnAct=_Any)
    al = _stub("albumentations", Compose=_Any, BboxParams=_Any)
    al.pytorch = _stub("albumentations.pytorch", ToTensorV2=_Any)
    pc = _stub("pycocotools")
    pc.coco = _stub("pycocotools.coco", COCO=_Any)
    pc.cocoeval = _stub("pycocotools.cocoeval", COCOeval=_Any)
    _stub("cv2")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import importlib
    mod = importlib.import_module("centernet_lightning.models.centernet")
    return mod.CenterNet


def make_fake_self(CenterNet, nms_kernel=3, num_detections=100, box_log=False, box_multiplier=1.0, stride=4):
    fs = SimpleNamespace(
        hparams=SimpleNamespace(nms_kernel=nms_kernel, num_detections=num_detections, box_log=box_log,
                                box_multiplier=box_multiplier),
        stride=stride)
    fs.get_topk_from_heatmap = lambda h, pseudo_nms=True: CenterNet.get_topk_from_heatmap(fs, h, pseudo_nms)
    return fs
