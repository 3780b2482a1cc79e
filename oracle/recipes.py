"""ORACLE — TEST INFRASTRUCTURE ONLY.  Seeded synthetic-input recipes shared by oracle/make_golden.py
(build container) and the tests/bench (GPU box), so large golden cases store only
{recipe name, seed, shape, SHA-256 of the input bytes} + the reference's outputs.

Conventions follow SURVEY.md §8(d) / BASELINE.md §5: heat = sigmoid(randn - 2.19) (init_bias of
configs/base_resnet34.yaml:16), box = rand * 16, reid = randn; images = rand in [0,1]
(tests/test_models.py:12 of the reference).
"""
import hashlib

import numpy as np
import torch


def _gen(seed):
    return torch.Generator(device="cpu").manual_seed(int(seed))


def decode_inputs(seed, shape, emb_dim=0):
    """-> heat (N,C,H,W) f32 in (0,1), box (N,4,H,W) f32 in [0,16) [, reid (N,E,H,W)] as torch CPU."""
    N, C, H, W = shape
    g = _gen(seed)
    heat = torch.randn(N, C, H, W, generator=g).sub_(2.19).sigmoid_()
    box = torch.rand(N, 4, H, W, generator=g).mul_(16.0)
    out = [heat, box]
    if emb_dim:
        out.append(torch.randn(N, emb_dim, H, W, generator=g))
    return out


def images(seed, shape):
    return torch.rand(*shape, generator=_gen(seed))


def sha256(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        a = t.detach().cpu().contiguous().numpy() if isinstance(t, torch.Tensor) else np.ascontiguousarray(t)
        h.update(a.tobytes())
    return h.hexdigest()
