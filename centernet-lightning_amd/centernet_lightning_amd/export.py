"""TorchScript export of the HIP forward (reference tools/export.py:7-12: LightningModule.to_torchscript(method="trace")).

The forward here is a launch plan, not a graph of ATen ops, so tracing needs ONE dispatcher-visible op that stands for it:
`centernet_gfx950::forward(x, params, config, sigmoid) -> Tensor[]`, registered with torch.library.  `TraceableCenterNet` wraps a
CenterNet so that torch.jit.trace records exactly that node, with the model's parameters / buffers as the traced module's own
tensors (they are saved inside the .pt) and the `model:` config section as a constant JSON string.  The saved file is therefore
self-contained: a process that has imported `centernet_lightning_amd` (which registers the op) can torch.jit.load() it and run it —
the op rebuilds a CenterNet from the JSON, adopts the tensors it is handed (no copy) and replays the same launch plan, bit for bit.

ONNX export (tools/export.py:14-19) is not offered: an ONNX graph has no way to carry a HIP launch plan (and neither `onnx` nor a
consumer for it exists in this image); export_onnx() says so instead of writing a file that could not run.
"""
import json
from typing import List

import torch
from torch import nn

_CACHE = {}          # (config json, data_ptrs of the tensors) -> CenterNet whose parameters ARE those tensors


def _state_tensors(model) -> List[torch.Tensor]:
    return [t for _, t in model.state_dict(keep_vars=True).items()]


def _model_for(config: str, params: List[torch.Tensor]):
    key = (config, tuple(int(p.data_ptr()) for p in params))
    model = _CACHE.get(key)
    if model is None:
        from .models import build_centernet
        model = build_centernet({"model": json.loads(config)})
        keys = list(model.state_dict().keys())
        if len(keys) != len(params):
            raise RuntimeError(f"centernet_gfx950::forward: {len(params)} tensors for a model with {len(keys)} state entries")
        # adopt the caller's tensors (no copy): the engine folds BatchNorm / packs weights from them on first use
        sd = dict(zip(keys, params))
        for name, mod in model.named_modules():
            for pname, p in list(mod._parameters.items()):
                if p is not None:
                    mod._parameters[pname] = nn.Parameter(sd[(name + "." if name else "") + pname].detach(), requires_grad=False)
            for bname, b in list(mod._buffers.items()):
                if b is not None:
                    mod._buffers[bname] = sd[(name + "." if name else "") + bname].detach()
        model._engine.invalidate()
        if len(_CACHE) > 16:
            _CACHE.pop(next(iter(_CACHE)))
        _CACHE[key] = model
    return model


@torch.library.custom_op("centernet_gfx950::forward", mutates_args=())
def forward_op(x: torch.Tensor, params: List[torch.Tensor], config: str, sigmoid: bool) -> List[torch.Tensor]:
    model = _model_for(config, params)
    return [t for t in model._engine.forward(x, sigmoid).values()]


@forward_op.register_fake
def _(x, params, config, sigmoid):
    m = json.loads(config)
    n, _, h, w = x.shape
    res = []
    for name, cfg in m["output_heads"].items():
        c = int(cfg["num_classes"]) if name == "heatmap" else (4 if name == "box_2d" else int(cfg.get("emb_dim", 64)))
        res.append(x.new_empty((n, h // 4, w // 4, c)).permute(0, 3, 1, 2))      # logical NCHW view of NHWC storage, like the op
    return res


class TraceableCenterNet(nn.Module):
    """forward(x) -> tuple(heatmap after sigmoid, box_2d[, reid]) through the dispatcher-visible op; trace or script THIS module."""

    def __init__(self, model):
        super().__init__()
        self.config = json.dumps(model.config_section)          # insertion order matters: heads are built (and their state laid out) in config order
        tensors = _state_tensors(model)
        self.params = nn.ParameterList([nn.Parameter(t.detach(), requires_grad=False) for t in tensors if t.is_floating_point()])
        self._float_slots = [i for i, t in enumerate(tensors) if t.is_floating_point()]
        self._ints = [(i, t.detach()) for i, t in enumerate(tensors) if not t.is_floating_point()]
        for j, (_, t) in enumerate(self._ints):          # num_batches_tracked etc.: plain buffers
            self.register_buffer(f"int_{j}", t)
        self.n_state = len(tensors)
        key = (self.config, tuple(int(t.data_ptr()) for t in tensors))
        _CACHE[key] = model                               # the live model serves its own traced calls (no rebuild)

    def _tensors(self) -> List[torch.Tensor]:
        out: List[torch.Tensor] = [torch.empty(0)] * self.n_state
        for slot, p in zip(self._float_slots, self.params):
            out[slot] = p
        for j, (slot, _) in enumerate(self._ints):
            out[slot] = getattr(self, f"int_{j}")
        return out

    def forward(self, x: torch.Tensor):
        outs = torch.ops.centernet_gfx950.forward(x, self._tensors(), self.config, True)
        return tuple(outs)


def export_torchscript(model, save_path=None, input_size=512, example_inputs=None):
    """tools/export.py:7-12 for this package: trace -> (optionally) save.  Returns the ScriptModule."""
    dev = next(model.parameters()).device
    x = example_inputs if example_inputs is not None else torch.rand(1, 3, input_size, input_size, device=dev)
    with torch.no_grad():
        traced = torch.jit.trace(TraceableCenterNet(model), x, check_trace=False)
    if save_path is not None:
        torch.jit.save(traced, save_path)
    return traced


def export_onnx(*_a, **_k):
    raise NotImplementedError("ONNX export is not offered: the forward is a HIP launch plan (centernet_gfx950::forward), which an ONNX graph "
                              "cannot carry; use export_torchscript() (the .pt replays through the registered op)")
