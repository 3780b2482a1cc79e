"""ORACLE — TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN
CODE in the build container (it cannot travel to the GPU box):

  * decode: CenterNet.decode_detections / get_topk_from_heatmap / gather_and_decode_boxes
            (centernet_lightning/models/centernet.py:229-304), imported via oracle/_ref_import.py
  * reid gather: EmbeddingHead.gather_at_indices (centernet_lightning/models/fairmot.py:63-73); that
            module does not import (fairmot.py:5 needs a missing symbol), so the function object is
            compiled from the reference file's AST at generation time and executed — nothing of
            its text is stored in this repo.
  * head / model wiring: GenericHead, GenericModel (centernet_lightning/models/meta.py:21-47) with an
            explicit conv3x3+BN+ReLU `block`.

Run:  python oracle/make_golden.py      (only where /root/reference exists)
Fixtures are data only: inputs (or their seeded recipe + SHA-256) and the reference's outputs.
"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipes                      # noqa: E402
import decode_ref                   # noqa: E402
from _ref_import import import_reference_centernet, make_fake_self, REF_ROOT   # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def ref_gather_at_indices():
    src = open(os.path.join(REF_ROOT, "centernet_lightning/models/fairmot.py")).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "gather_at_indices":
            node.decorator_list = []
            mod = ast.Module(body=[node], type_ignores=[])
            ns = {"torch": torch}
            exec(compile(mod, "fairmot.py", "exec"), ns)
            return lambda reid, idx: ns["gather_at_indices"](None, reid, idx)
    raise RuntimeError("gather_at_indices not found")


def tie_free(scores_sorted_kplus1):
    s = np.asarray(scores_sorted_kplus1)
    return bool(np.all(s[:, :-1] > s[:, 1:]))


def run_ref_decode(CenterNet, heat, box, k, nms, normalize, box_log, mult, stride):
    fs = make_fake_self(CenterNet, nms_kernel=nms, num_detections=k, box_log=box_log, box_multiplier=mult,
                        stride=stride)
    scores, indices, labels = CenterNet.get_topk_from_heatmap(fs, heat)
    out = CenterNet.decode_detections(fs, heat, box, normalize_boxes=normalize)
    assert torch.equal(out["scores"], scores) and torch.equal(out["labels"], labels)
    return scores.numpy(), indices.numpy(), labels.numpy(), out["boxes"].numpy()


def main():
    os.makedirs(OUT, exist_ok=True)
    CenterNet = import_reference_centernet()
    gather_ref = ref_gather_at_indices()
    meta = {"torch": torch.__version__}

    # ---------------- randomized decode cases (input by recipe) ----------------
    cases = [
        # name, seed, shape, emb, k, nms, normalize, box_log, mult
        ("det128_s0", 0, (2, 80, 128, 128), 0, 100, 3, False, False, 1.0),
        ("det128_s1", 1, (2, 80, 128, 128), 0, 100, 3, True, False, 1.0),
        ("det128_s2_log16", 2, (1, 80, 128, 128), 0, 100, 3, False, True, 16.0),
        ("det128_s3_k300_nms5", 3, (1, 80, 128, 128), 0, 300, 5, False, False, 1.0),
        ("mot_s0", 0, (2, 2, 152, 272), 64, 100, 3, False, False, 1.0),
        ("mot_s1_k300", 1, (1, 2, 152, 272), 64, 300, 3, True, False, 1.0),
        ("small_s0", 0, (1, 20, 32, 32), 0, 100, 3, False, False, 1.0),
        ("small_s1_c1", 1, (3, 1, 24, 40), 8, 50, 3, False, False, 1.0),
        ("small_s2_c3", 2, (2, 3, 16, 16), 0, 20, 1, False, False, 2.0),
    ]
    for name, seed, shape, emb, k, nms, normalize, box_log, mult in cases:
        ins = recipes.decode_inputs(seed, shape, emb)
        heat, box = ins[0], ins[1]
        s, i, l, b = run_ref_decode(CenterNet, heat, box, k, nms, normalize, box_log, mult, 4)
        # tie-freeness of the top-(k+1): the only regime in which torch.topk's order is defined
        fs = make_fake_self(CenterNet, nms_kernel=nms, num_detections=k + 1)
        s1, _, _ = CenterNet.get_topk_from_heatmap(fs, heat)
        tf = tie_free(s1.numpy())
        # the restatement must agree bit-for-bit (after canonicalisation when ties exist)
        o = decode_ref.decode_detections(heat.numpy(), box.numpy(), k, nms, normalize, box_log, mult, 4)
        cs, ci, cl, cb = decode_ref.canonicalize(s, i, l, b)
        assert np.array_equal(cs, o["scores"]) and np.array_equal(ci, o["indices"]), name
        assert np.array_equal(cl, o["labels"]), name
        if box_log:      # exp() is not bit-reproducible across libm/SLEEF/HIP: tolerance, not bit-equality
            assert np.allclose(cb, o["boxes"], rtol=2e-6, atol=1e-5), name
        else:
            assert np.array_equal(cb.view(np.uint32), o["boxes"].view(np.uint32)), name
        if tf:
            assert np.array_equal(i, ci), name
        payload = dict(seed=seed, shape=np.array(shape), emb=emb, k=k, nms=nms, normalize=normalize,
                       box_log=box_log, mult=mult, stride=4, tie_free=tf,
                       sha=recipes.sha256(*ins), scores=cs, indices=ci, labels=cl, boxes=cb,
                       raw_indices=i)
        if emb:
            e = gather_ref(ins[2], torch.from_numpy(ci)).numpy()
            assert np.array_equal(e, decode_ref.gather_at_indices(ins[2].numpy(), ci)), name
            payload["embeddings_sha"] = recipes.sha256(e)
            payload["embeddings_head"] = e[:, :4].copy()          # first 4 detections in full
        if np.prod(shape) <= 32 * 1024:                           # small: keep the input itself too
            payload["heat"] = heat.numpy()
            payload["box"] = box.numpy()
            if emb:
                payload["reid"] = ins[2].numpy()
        np.savez_compressed(os.path.join(OUT, f"decode_{name}.npz"), **payload)
        print(f"decode_{name}: tie_free={tf} top={cs[0, :3]} idx={ci[0, :3]}")

    # ---------------- hand-made known-answer tests, stored in full ----------------
    kats = {}
    H = W = 8
    # (a) plateau: two equal neighbours both survive; border peaks; fewer than k peaks
    h = np.zeros((1, 3, H, W), np.float32)
    h[0, 0, 0, 0] = 0.9                      # corner
    h[0, 1, 3, 3] = 0.7; h[0, 1, 3, 4] = 0.7  # plateau (both kept -> a tie)
    h[0, 2, 7, 5] = 0.8                      # bottom border
    h[0, 0, 5, 1] = 0.5; h[0, 2, 5, 1] = 0.5  # same pixel, two classes tie -> lowest class wins
    h[0, 1, 5, 2] = 0.4                      # suppressed? different class than neighbour -> survives
    h[0, 0, 6, 6] = 0.3; h[0, 0, 6, 7] = 0.31  # 0.3 suppressed by its neighbour
    kats["plateau"] = (h, 10, 3)
    # (b) all-equal heatmap: everything is a plateau, all ties
    kats["allequal"] = (np.full((1, 2, H, W), 0.25, np.float32), 12, 3)
    # (c) negative values + zeros (non-sigmoid input): h*mask gives -0.0 at suppressed negatives
    rng = np.random.default_rng(7)
    kats["signed"] = (rng.standard_normal((2, 4, H, W)).astype(np.float32), 16, 3)
    # (d) k == H*W
    kats["kfull"] = (rng.random((1, 2, 4, 4)).astype(np.float32), 16, 3)
    for name, (h, k, nms) in kats.items():
        box = (rng.random((h.shape[0], 4) + h.shape[2:]) * 8 - 1).astype(np.float32)   # some negative -> clamp
        s, i, l, b = run_ref_decode(CenterNet, torch.from_numpy(h), torch.from_numpy(box), k, nms, False, False, 1.0, 4)
        o = decode_ref.decode_detections(h, box, k, nms, False, False, 1.0, 4)
        assert np.array_equal(np.sort(s, axis=1)[:, ::-1], o["scores"]), name       # same multiset, sorted
        cs, ci, cl, cb = decode_ref.canonicalize(s, i, l, b)
        assert np.array_equal(cs, o["scores"]), name
        # Inside a tie group that straddles position k the reference may pick different members than the
        # canonical rule; compare what is well-defined: for every returned index, label/box match.
        for n in range(h.shape[0]):
            ref_map = {int(ii): (int(ll), bb.tobytes()) for ii, ll, bb in zip(i[n], l[n], b[n])}
            for ii, ll, bb in zip(o["indices"][n], o["labels"][n], o["boxes"][n]):
                if int(ii) in ref_map:
                    assert ref_map[int(ii)] == (int(ll), bb.tobytes()), (name, ii)
        np.savez_compressed(os.path.join(OUT, f"kat_{name}.npz"), heat=h, box=box, k=k, nms=nms,
                            ref_scores=s, ref_indices=i, ref_labels=l, ref_boxes=b,
                            scores=o["scores"], indices=o["indices"], labels=o["labels"], boxes=o["boxes"])
        print(f"kat_{name}: ref idx {i[0, :6]} canon idx {o['indices'][0, :6]}")

    # ---------------- head / model wiring (meta.py:21-47) ----------------
    from torch import nn
    import importlib
    metamod = importlib.import_module("centernet_lightning.models.meta")

    def block(in_c, out_c):
        return nn.Sequential(nn.Conv2d(in_c, out_c, 3, padding=1, bias=False), nn.BatchNorm2d(out_c), nn.ReLU(inplace=True))

    torch.manual_seed(11)
    heads = nn.ModuleDict({
        "heatmap": metamod.GenericHead(8, 5, width=16, depth=2, block=block, init_bias=-2.19),
        "box_2d": metamod.GenericHead(8, 4, width=16, depth=3, block=block, init_bias=10.0),
        "reid": metamod.GenericHead(8, 6, width=16, depth=1, block=block, init_bias=None),
    })
    for m in heads.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    heads.eval()

    class BB(nn.Module):
        def forward_features(self, x):
            return [x, x[:, :, ::2, ::2] * 2.0]

    class NK(nn.Module):
        def forward(self, feats):
            return feats[0] + torch.nn.functional.interpolate(feats[1], scale_factor=2, mode="nearest")

    model = metamod.GenericModel(BB(), NK(), heads)
    x = torch.randn(2, 8, 12, 20)
    with torch.no_grad():
        out = model(x)
        neck = NK()(BB().forward_features(x))
    payload = {"x": x.numpy(), "neck": neck.numpy(), "head_order": np.array(list(out.keys()))}
    for k_, v in out.items():
        payload[f"out.{k_}"] = v.numpy()
    for k_, v in heads.state_dict().items():
        payload[f"sd.{k_}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "head_wiring.npz"), **payload)
    print("head_wiring:", {k_: tuple(v.shape) for k_, v in out.items()})
    print("meta", meta)


if __name__ == "__main__":
    main()
