"""CPU: pins the oracle (oracle/decode_ref.py, oracle/ref_cpu.py) against the golden vectors produced by the
reference's own code (oracle/make_golden.py), and against the live reference where /root/reference exists."""
import glob
import os

import numpy as np
import pytest
import torch

import decode_ref
import recipes
import ref_cpu


def _load(path):
    return dict(np.load(path, allow_pickle=False))


def _inputs(g):
    if "heat" in g:
        ins = [torch.from_numpy(g["heat"]), torch.from_numpy(g["box"])] + ([torch.from_numpy(g["reid"])] if "reid" in g else [])
    else:
        ins = recipes.decode_inputs(int(g["seed"]), tuple(int(v) for v in g["shape"]), int(g["emb"]))
    assert recipes.sha256(*recipes.decode_inputs(int(g["seed"]), tuple(int(v) for v in g["shape"]), int(g["emb"]))) == str(g["sha"]), \
        "seeded input recipe no longer reproduces the bytes the golden outputs were generated from"
    return ins


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "decode_*.npz"))),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_decode_oracle_matches_reference_golden(path):
    g = _load(path)
    ins = _inputs(g)
    o = decode_ref.decode_detections(ins[0].numpy(), ins[1].numpy(), int(g["k"]), int(g["nms"]), bool(g["normalize"]),
                                     bool(g["box_log"]), float(g["mult"]), int(g["stride"]),
                                     reid=ins[2].numpy() if len(ins) > 2 else None)
    assert np.array_equal(o["scores"], g["scores"])
    assert np.array_equal(o["indices"], g["indices"])          # top-k indices bit-exact
    assert np.array_equal(o["labels"], g["labels"])
    if bool(g["tie_free"]):
        assert np.array_equal(g["raw_indices"], g["indices"])   # canonical order == torch.topk order when tie-free
    if bool(g["box_log"]):
        np.testing.assert_allclose(o["boxes"], g["boxes"], rtol=2e-6, atol=1e-5)   # exp() is libm-dependent
    else:
        assert np.array_equal(o["boxes"].view(np.uint32), g["boxes"].view(np.uint32))
    if "embeddings_sha" in g:
        assert recipes.sha256(o["embeddings"]) == str(g["embeddings_sha"])
        assert np.array_equal(o["embeddings"][:, :4], g["embeddings_head"])


@pytest.mark.parametrize("name", ["plateau", "allequal", "signed", "kfull"])
def test_decode_oracle_kats(name, golden_dir):
    g = _load(os.path.join(golden_dir, f"kat_{name}.npz"))
    o = decode_ref.decode_detections(g["heat"], g["box"], int(g["k"]), int(g["nms"]))
    for key in ("scores", "indices", "labels"):
        assert np.array_equal(o[key], g[key]), key
    assert np.array_equal(o["boxes"].view(np.uint32), g["boxes"].view(np.uint32))
    # against what the reference itself returned: same score multiset; its labels/boxes agree per index
    assert np.array_equal(np.sort(g["ref_scores"], axis=1)[:, ::-1], o["scores"])
    for n in range(g["heat"].shape[0]):
        ref = {int(i): (int(l), b.tobytes()) for i, l, b in zip(g["ref_indices"][n], g["ref_labels"][n], g["ref_boxes"][n])}
        hits = 0
        for i, l, b in zip(o["indices"][n], o["labels"][n], o["boxes"][n]):
            if int(i) in ref:
                hits += 1
                assert ref[int(i)] == (int(l), b.tobytes())
        assert hits > 0 or name == "allequal"      # all-tied: the reference's arbitrary pick may not overlap


def test_plateau_semantics(golden_dir):
    g = _load(os.path.join(golden_dir, "kat_plateau.npz"))
    o = decode_ref.decode_detections(g["heat"], g["box"], 10, 3)
    idx = list(o["indices"][0])
    assert idx[0] == 0 and o["scores"][0, 0] == np.float32(0.9)              # corner peak
    assert idx[1] == 7 * 8 + 5                                                # bottom border
    assert idx[2:4] == [3 * 8 + 3, 3 * 8 + 4]                                 # plateau: both kept, index order
    assert idx[4] == 5 * 8 + 1 and o["labels"][0, 4] == 0                     # class tie -> lowest class
    assert 6 * 8 + 6 not in idx[:7]                                           # suppressed neighbour


def test_canonicalize_is_identity_on_tie_free():
    h, b = [t.numpy() for t in recipes.decode_inputs(5, (1, 4, 16, 16))]
    o = decode_ref.decode_detections(h, b, 20)
    s, i, l = decode_ref.canonicalize(o["scores"], o["indices"], o["labels"])
    assert np.array_equal(i, o["indices"]) and np.array_equal(l, o["labels"])


def test_torch_decode_used_by_the_cpu_baseline_agrees_with_the_numpy_oracle():
    """bench.py's cpu_baseline times decode_detections_torch (the reference's torch op sequence, multi-threaded); it must be the same
    function as the numpy oracle up to tie order."""
    g = torch.Generator().manual_seed(5)
    heat = torch.sigmoid(torch.randn(3, 7, 24, 40, generator=g) * 2)
    box = torch.rand(3, 4, 24, 40, generator=g) * 9 - 0.5
    reid = torch.randn(3, 16, 24, 40, generator=g)
    a = decode_ref.decode_detections(heat.numpy(), box.numpy(), 50, 3, reid=reid.numpy())
    b = decode_ref.decode_detections_torch(heat, box, 50, 3, reid=reid)
    ca = decode_ref.canonicalize(a["scores"], a["indices"], a["boxes"], a["labels"], a["embeddings"])
    cb = decode_ref.canonicalize(b["scores"].numpy(), b["indices"].numpy(), b["boxes"].numpy(), b["labels"].numpy(), b["embeddings"].numpy())
    for x, y in zip(ca, cb):
        np.testing.assert_array_equal(x, y)
    c = decode_ref.decode_detections_torch(heat, box, 50, 3, normalize_boxes=True)
    d = decode_ref.decode_detections(heat.numpy(), box.numpy(), 50, 3, normalize_boxes=True)
    np.testing.assert_allclose(decode_ref.canonicalize(c["scores"].numpy(), c["indices"].numpy(), c["boxes"].numpy())[2],
                               decode_ref.canonicalize(d["scores"], d["indices"], d["boxes"])[2], rtol=1e-6, atol=1e-7)


def test_pack_unpack_roundtrip():
    h, b, r = [t.numpy() for t in recipes.decode_inputs(3, (2, 5, 16, 24), 8)]
    o = decode_ref.decode_detections(h, b, 30, reid=r)
    rec = decode_ref.pack_detections(o["boxes"], o["scores"], o["labels"], o["embeddings"])
    assert rec.shape == (2, 30, 14) and rec.dtype == np.float32
    u = decode_ref.unpack_detections(rec)
    for k in ("boxes", "scores", "labels", "embeddings"):
        assert np.array_equal(u[k], o[k])


def test_head_wiring_matches_reference_generic_head(golden_dir):
    """oracle/ref_cpu.head_forward vs outputs of the reference's GenericHead / GenericModel (meta.py:21-47)."""
    g = _load(os.path.join(golden_dir, "head_wiring.npz"))
    sd = {}
    for k, v in g.items():
        if k.startswith("sd."):
            # reference block = Sequential(conv, bn, relu): block_i.0 -> block_i.conv, block_i.1 -> block_i.bn
            name = k[3:].replace(".0.", ".conv.").replace(".1.", ".bn.")
            sd["heads." + name] = torch.from_numpy(v)
    neck = torch.from_numpy(g["neck"])
    assert list(g["head_order"]) == ref_cpu.head_names(sd)
    for name in ref_cpu.head_names(sd):
        out = ref_cpu.head_forward(sd, name, neck)
        np.testing.assert_allclose(out.numpy(), g[f"out.{name}"], rtol=1e-5, atol=1e-5)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree only exists in the build container")
def test_decode_oracle_matches_live_reference():
    from _ref_import import import_reference_centernet, make_fake_self
    CenterNet = import_reference_centernet()
    for seed, shape, k, nms, norm in [(11, (2, 7, 20, 28), 40, 3, False), (12, (1, 80, 64, 64), 100, 3, True),
                                      (13, (2, 3, 32, 32), 64, 5, False)]:
        heat, box = recipes.decode_inputs(seed, shape)
        fs = make_fake_self(CenterNet, nms_kernel=nms, num_detections=k)
        s, i, l = CenterNet.get_topk_from_heatmap(fs, heat)
        b = CenterNet.decode_detections(fs, heat, box, normalize_boxes=norm)["boxes"]
        o = decode_ref.decode_detections(heat.numpy(), box.numpy(), k, nms, norm)
        cs, ci, cl, cb = decode_ref.canonicalize(s.numpy(), i.numpy(), l.numpy(), b.numpy())
        assert np.array_equal(cs, o["scores"]) and np.array_equal(ci, o["indices"]) and np.array_equal(cl, o["labels"])
        assert np.array_equal(cb.view(np.uint32), o["boxes"].view(np.uint32))


def test_normalize_u8_known_answers():
    """albumentations-style normalisation (oracle restatement): hand-checkable values."""
    img = np.array([[[[0, 0, 0], [255, 255, 255], [124, 116, 104]]]], dtype=np.uint8)
    out = decode_ref.normalize_u8(img)
    assert out.dtype == np.float32 and out.shape == (1, 1, 3, 3)
    np.testing.assert_allclose(out[0, 0, 0], [-0.485 / 0.229, -0.456 / 0.224, -0.406 / 0.225], rtol=1e-6)
    np.testing.assert_allclose(out[0, 0, 1], [(1 - 0.485) / 0.229, (1 - 0.456) / 0.224, (1 - 0.406) / 0.225], rtol=1e-6)
    np.testing.assert_allclose(out[0, 0, 2], [(124 / 255 - 0.485) / 0.229, (116 / 255 - 0.456) / 0.224, (104 / 255 - 0.406) / 0.225], rtol=1e-5, atol=1e-6)
