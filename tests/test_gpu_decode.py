"""GPU: parity of the fused HIP decode (through the C ABI) with the oracle and the reference-generated golden
vectors.  Bar: scores / indices / labels / boxes / embeddings BIT-EXACT (compare-select and one-rounding fp32
arithmetic); only box_log=True (exp) is compared with a tolerance."""
import glob
import os

import numpy as np
import pytest
import torch

import decode_ref
import recipes
import centernet_lightning_amd as cl
from centernet_lightning_amd import decode as hip_decode

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _np(d):
    return {k: v.cpu().numpy() for k, v in d.items()}


def _check(o, ref, box_tol=False):
    for key in ("scores", "indices", "labels"):
        assert np.array_equal(o[key], ref[key]), key
    if box_tol:
        np.testing.assert_allclose(o["boxes"], ref["boxes"], rtol=2e-6, atol=1e-5)
    else:
        assert np.array_equal(o["boxes"].view(np.uint32), ref["boxes"].view(np.uint32))
    if "embeddings" in ref and ref["embeddings"] is not None:
        assert np.array_equal(o["embeddings"], ref["embeddings"])


def _layouts(t):
    """same logical NCHW tensor as (a) contiguous NCHW, (b) channels_last / NHWC storage."""
    return [t.cuda().contiguous(), t.cuda().contiguous(memory_format=torch.channels_last)]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "decode_*.npz"))), ids=lambda p: os.path.basename(p)[7:-4])
def test_decode_matches_reference_golden(path):
    g = dict(np.load(path))
    shape = tuple(int(v) for v in g["shape"])
    ins = recipes.decode_inputs(int(g["seed"]), shape, int(g["emb"]))
    assert recipes.sha256(*ins) == str(g["sha"])
    for li in range(2):
        heat, box = _layouts(ins[0])[li], _layouts(ins[1])[li]
        reid = _layouts(ins[2])[li] if len(ins) > 2 else None
        o = _np(hip_decode.decode(heat, box, reid, int(g["k"]), int(g["nms"]), bool(g["normalize"]), bool(g["box_log"]),
                                  float(g["mult"]), int(g["stride"])))
        _check(o, {k: g[k] for k in ("scores", "indices", "labels", "boxes")}, box_tol=bool(g["box_log"]))
        if reid is not None:
            assert recipes.sha256(o["embeddings"]) == str(g["embeddings_sha"])


@pytest.mark.parametrize("name", ["plateau", "allequal", "signed", "kfull"])
def test_decode_kats(name):
    g = dict(np.load(os.path.join(GOLDEN, f"kat_{name}.npz")))
    for heat, box in zip(_layouts(torch.from_numpy(g["heat"])), _layouts(torch.from_numpy(g["box"]))):
        o = _np(hip_decode.decode(heat, box, None, int(g["k"]), int(g["nms"])))
        _check(o, g)


@pytest.mark.parametrize("shape,k,nms,emb", [((3, 80, 128, 128), 100, 3, 0), ((2, 2, 152, 272), 300, 3, 64), ((2, 5, 33, 47), 77, 5, 3),
                                            ((1, 1, 8, 200), 9, 7, 0), ((2, 6, 16, 16), 256, 1, 0), ((1, 81, 20, 20), 50, 3, 0),
                                            ((1, 130, 12, 12), 20, 3, 0), ((2, 16, 40, 36), 50, 7, 0), ((1, 80, 20, 24), 30, 7, 0), ((2, 8, 33, 66), 64, 5, 0)])
def test_decode_random_vs_oracle(shape, k, nms, emb):
    ins = recipes.decode_inputs(sum(shape) + k, shape, emb)
    ref = decode_ref.decode_detections(ins[0].numpy(), ins[1].numpy(), k, nms, reid=ins[2].numpy() if emb else None)
    for li in range(2):
        heat, box = _layouts(ins[0])[li], _layouts(ins[1])[li]
        reid = _layouts(ins[2])[li] if emb else None
        o = _np(hip_decode.decode(heat, box, reid, k, nms))
        _check(o, ref)


def _sweep_cases():
    """Edge shapes of the C % 8 == 0 stage-1 kernel (runs of 4 pixels, 64-pixel blocks, 16- / 4-row strips) and of stage 2's pruning bound
    (k from 1 to H*W, maps smaller than a workgroup, more than 16 keys per thread), seeded."""
    rng = np.random.default_rng(2024)
    cases = []
    for W in (8, 9, 13, 63, 64, 65, 130):
        for H in (1, 3, 16, 17, 33):
            C = int(rng.choice([8, 16, 40, 80, 88]))
            N = int(rng.choice([1, 2, 3]))
            nms = int(rng.choice([1, 3, 5]))
            k = int(rng.integers(1, min(H * W, 400) + 1))
            cases.append((N, C, H, W, k, nms))
    cases += [(40, 8, 48, 20, 33, 3), (20, 16, 70, 64, 1024, 3), (1, 80, 152, 272, 1000, 3), (2, 8, 4, 8, 32, 7), (1, 8, 6, 1500, 200, 3), (2, 16, 3, 700, 64, 5)]
    return cases


@pytest.mark.parametrize("case", _sweep_cases(), ids=lambda c: "N{}C{}_{}x{}_k{}_nms{}".format(*c))
def test_decode_shape_sweep_vs_oracle(case):
    N, C, H, W, k, nms = case
    ins = recipes.decode_inputs(N * 7 + C + H * 3 + W + k, (N, C, H, W), 0)
    heat = ins[0]
    if (H + W) % 3 == 0:                                 # a third of the cases: quantised scores -> ties across the k boundary and plateaus
        heat = (heat * 16).floor() / 16
    ref = decode_ref.decode_detections(heat.numpy(), ins[1].numpy(), k, nms)
    _check(_np(hip_decode.decode(_layouts(heat)[1], _layouts(ins[1])[1], None, k, nms)), ref)       # NHWC storage: the channel-minor kernels
    if C * H * W <= 40000:
        _check(_np(hip_decode.decode(_layouts(heat)[0], _layouts(ins[1])[0], None, k, nms)), ref)   # contiguous NCHW: the generic kernel


def test_decode_quantised_scores_many_ties():
    """Scores on a coarse grid produce large tie groups everywhere, including across the k boundary."""
    g = torch.Generator().manual_seed(3)
    heat = (torch.rand(2, 4, 24, 24, generator=g) * 8).floor() / 8
    box = torch.rand(2, 4, 24, 24, generator=g) * 5
    ref = decode_ref.decode_detections(heat.numpy(), box.numpy(), 60, 3)
    for h, b in zip(_layouts(heat), _layouts(box)):
        _check(_np(hip_decode.decode(h, b, None, 60, 3)), ref)


@pytest.mark.parametrize("shape,k", [((2, 4, 128, 128), 100), ((2, 3, 96, 96), 100), ((1, 2, 152, 272), 300), ((2, 2, 200, 200), 120), ((1, 4, 128, 128), 400)])
@pytest.mark.parametrize("kind", ["random", "quantised", "plateau_rows", "all_equal", "one_slab_only"])
def test_decode_topk_on_large_maps_with_ties(shape, k, kind):
    """Maps of 9 K - 41 K pixels (keys in registers / in LDS) with the tie structures that stress the selection: quantised scores, whole rows that are
    plateaus (the radix-select path), everything equal, all winners in the last rows.  Bit-exact against the oracle (scores, indices in canonical
    order, labels, boxes); three runs give identical bytes.  (Written for round 6's experiment r6b — the top-k on several workgroups per image, merged by
    the last arriver: bit-exact on these cases too, but slower, profiles/r06_experiments.txt — and kept for the kernel that stayed.)"""
    N, C, H, W = shape
    g = torch.Generator().manual_seed(H * W + k)
    if kind == "random":
        heat = torch.rand(N, C, H, W, generator=g)
    elif kind == "quantised":
        heat = (torch.rand(N, C, H, W, generator=g) * 16).floor() / 16
    elif kind == "plateau_rows":
        heat = torch.rand(N, C, H, 1, generator=g).expand(N, C, H, W).contiguous()      # every row constant: every pixel of a row-maximum row survives the pool... in x
        heat = (heat * 8).floor() / 8
    elif kind == "all_equal":
        heat = torch.full((N, C, H, W), 0.25)
    else:
        heat = torch.rand(N, C, H, W, generator=g) * 0.01
        heat[:, :, H - 20:H - 4, :] += 0.5                                                  # all winners in the last slab
    box = torch.rand(N, 4, H, W, generator=g) * 7
    ref = decode_ref.decode_detections(heat.numpy(), box.numpy(), k, 3)
    outs = [_np(hip_decode.decode(_layouts(heat)[1], _layouts(box)[1], None, k, 3)) for _ in range(3)]
    _check(outs[0], ref)
    for o in outs[1:]:
        assert all((o[key] == outs[0][key]).all() for key in outs[0])


def _bound_cases():
    """Seeded random (shape, k, nms, score distribution) cases for stage 2's two-step pruning bound: every key storage (16 registers, 48 registers, LDS, 16-bit LDS
    prefilter, neither) x k from 1 to 1024 x distributions that put the k-th largest thread maximum at either end of the bins, in one bin, or on a tie."""
    rng = np.random.default_rng(606)
    shapes = [(96, 96), (128, 128), (100, 131), (152, 272), (160, 256), (200, 200), (199, 201), (64, 700), (152, 300), (31, 33), (256, 256)]
    kinds = ["uniform", "heavy_tail", "near_constant", "two_level", "wide_range", "few_distinct"]
    cases = []
    for i in range(36):
        H, W = shapes[int(rng.integers(len(shapes)))]
        cases.append((int(rng.choice([1, 2])), int(rng.choice([1, 2, 3])), H, W, int(min(rng.choice([1, 7, 100, 300, 511, 1024]), H * W)), int(rng.choice([1, 3])),
                      kinds[i % len(kinds)], i))
    return cases


@pytest.mark.parametrize("case", _bound_cases(), ids=lambda c: "N{}C{}_{}x{}_k{}_nms{}_{}_{}".format(*c))
def test_decode_topk_bound_random_distributions(case):
    """Round 6's second step of the pruning bound (the k-th largest thread maximum to within one of 512-1024 bins between the first bound and the largest maximum) and the
    four-candidates-per-lane rank: bit-exact against the oracle whatever the distribution does to the bins."""
    N, C, H, W, k, nms, kind, seed = case
    g = torch.Generator().manual_seed(1000 + seed)
    u = torch.rand(N, C, H, W, generator=g)
    if kind == "uniform":
        heat = u
    elif kind == "heavy_tail":
        heat = u ** 12                                    # a handful of large scores, the k-th far below the maximum
    elif kind == "near_constant":
        heat = 0.5 + u * 1e-6                             # the whole range inside a few hundred float steps: bins of width 1
    elif kind == "two_level":
        heat = torch.where(u > 0.999, 0.5 + u, u * 1e-3)  # a few winners, then a cliff
    elif kind == "wide_range":
        heat = (u - 0.3) * torch.exp((torch.rand(N, C, H, W, generator=g) - 0.5) * 40)      # signs, zeros' neighbourhood, 17 decades
    else:
        heat = (u * 5).floor() / 5                        # five values: every bound lands on a tie
    box = torch.rand(N, 4, H, W, generator=g) * 9
    ref = decode_ref.decode_detections(heat.numpy(), box.numpy(), k, nms)
    _check(_np(hip_decode.decode(_layouts(heat)[1], _layouts(box)[1], None, k, nms)), ref)
    if seed % 3 == 0:
        _check(_np(hip_decode.decode(_layouts(heat)[0], _layouts(box)[0], None, k, nms)), ref)


def _planes_cases():
    """Contiguous NCHW maps (the reference's own layout) of >= 4 classes with W % 4 == 0 and the 3 x 3 pool run stage 1's class-planes kernel (round 6): strips of 8 rows x
    64 pixels and 16 class groups (128 / 256 pixels and 8 / 4 groups below 16 / 8 classes).  Edge shapes: one row, partial strips, partial / several x tiles, class counts around the group count; a slice of a wider tensor (row pitch > W)."""
    cases = []
    for C in (4, 7, 8, 15, 16, 17, 31, 80, 100):
        for H, W in ((1, 4), (7, 8), (8, 64), (9, 60), (33, 68), (16, 132)):
            cases.append((1 + (C + H) % 2, C, H, W))
    return cases


@pytest.mark.parametrize("case", _planes_cases(), ids=lambda c: "N{}C{}_{}x{}".format(*c))
def test_decode_class_planes_kernel_shapes(case):
    N, C, H, W = case
    g = torch.Generator().manual_seed(C * 1000 + H * 10 + W)
    heat = torch.rand(N, C, H, W, generator=g)
    if (C + W) % 3 == 0:
        heat = (heat * 8).floor() / 8                     # ties across classes and pixels: first class / lowest index must win
    box = torch.rand(N, 4, H, W, generator=g) * 6
    k = min(50, H * W)
    ref = decode_ref.decode_detections(heat.numpy(), box.numpy(), k, 3)
    _check(_np(hip_decode.decode(heat.cuda(), box.cuda(), None, k, 3)), ref)
    # the same maps as a window of a wider, taller buffer: row pitch and plane pitch no longer W and H * W (still multiples of 4)
    big = torch.zeros(N, C, H + 3, W + 8)
    big[:, :, 1:H + 1, 4:W + 4] = heat
    _check(_np(hip_decode.decode(big.cuda()[:, :, 1:H + 1, 4:W + 4], box.cuda(), None, k, 3)), ref)


@pytest.mark.parametrize("nms", [1, 5, 7])
@pytest.mark.parametrize("shape", [(2, 16, 9, 60), (1, 80, 33, 68), (1, 7, 16, 132), (2, 100, 8, 64), (1, 31, 3, 8)], ids=lambda s: "N{}C{}_{}x{}".format(*s))
def test_decode_class_planes_kernel_pools(shape, nms):
    """The class-planes kernel's other instantiations: no pool, 5 x 5, 7 x 7 (strips of 4 rows from 5 x 5 up)."""
    N, C, H, W = shape
    g = torch.Generator().manual_seed(C * 100 + H + W + nms)
    heat = torch.rand(N, C, H, W, generator=g)
    if (H + nms) % 2 == 0:
        heat = (heat * 6).floor() / 6
    box = torch.rand(N, 4, H, W, generator=g) * 6
    k = min(40, H * W)
    ref = decode_ref.decode_detections(heat.numpy(), box.numpy(), k, nms)
    _check(_np(hip_decode.decode(heat.cuda(), box.cuda(), None, k, nms)), ref)


def test_decode_full_size_properties():
    """BASELINE C1 size (32x80x128x128): properties that need no oracle run — sortedness, peak-ness, top-k-ness."""
    N, C, H, W, k = 32, 80, 128, 128, 100
    g = torch.Generator(device="cuda").manual_seed(0)
    heat = torch.randn(N, H, W, C, device="cuda", generator=g).sub_(2.19).sigmoid_().permute(0, 3, 1, 2)
    box = (torch.rand(N, H, W, 4, device="cuda", generator=g) * 16).permute(0, 3, 1, 2)
    o = hip_decode.decode(heat, box, None, k, 3)
    s, i, l, b = o["scores"], o["indices"], o["labels"], o["boxes"]
    assert bool((s[:, :-1] >= s[:, 1:]).all())
    flat = heat.reshape(N, C, H * W)
    got = torch.gather(flat, 2, i.unsqueeze(1).expand(-1, C, -1))                       # values at winners: N,C,k
    assert torch.equal(got.gather(1, l.unsqueeze(1)).squeeze(1), s)                     # score == heat[label, idx]
    pooled = torch.nn.functional.max_pool2d(heat, 3, 1, 1)
    assert bool((pooled.reshape(N, C, -1).gather(2, i.unsqueeze(1).expand(-1, C, -1)).gather(1, l.unsqueeze(1)).squeeze(1) == s).all())
    masked = (heat * (pooled == heat)).amax(dim=1).reshape(N, -1)
    kth = masked.topk(k, dim=1).values
    assert torch.equal(kth, s)                                                          # same score multiset as torch
    cx = (i % W).float() + 0.5
    ltrb = torch.gather(box.reshape(N, 4, -1), 2, i.unsqueeze(1).expand(-1, 4, -1)).clamp_min(0)
    assert torch.equal(b[..., 0], (cx - ltrb[:, 0]) * 4) and torch.equal(b[..., 2], (cx + ltrb[:, 2]) * 4)
    # idempotence / determinism
    o2 = hip_decode.decode(heat, box, None, k, 3)
    assert all(torch.equal(o[key], o2[key]) for key in o)
    # layout independence at full size: the same logical maps as contiguous NCHW tensors (the reference's layout: the class-planes stage 1) give the same bytes,
    # for every pool size both layouts have a kernel of their own for
    for nms in (3, 1, 5, 7):
        a_ = hip_decode.decode(heat, box, None, k, nms)
        b_ = hip_decode.decode(heat.contiguous(), box.contiguous(), None, k, nms)
        assert all(torch.equal(a_[key], b_[key]) for key in a_), nms


def test_decode_tracking_shape_layout_independence():
    """BASELINE C4's maps (2 x 152 x 272 + 64-d embeddings, k = 300; 8 images): channel-minor and contiguous NCHW inputs — different stage-1 kernels, 16-byte against strided
    embedding gathers — give the same bytes, run after run."""
    N, C, H, W, k, E = 8, 2, 152, 272, 300, 64
    g = torch.Generator(device="cuda").manual_seed(4)
    heat = torch.randn(N, H, W, C, device="cuda", generator=g).sub_(2.19).sigmoid_().permute(0, 3, 1, 2)
    box = (torch.rand(N, H, W, 4, device="cuda", generator=g) * 16).permute(0, 3, 1, 2)
    emb = torch.randn(N, H, W, E, device="cuda", generator=g).permute(0, 3, 1, 2)
    a_ = hip_decode.decode(heat, box, emb, k, 3)
    b_ = hip_decode.decode(heat.contiguous(), box.contiguous(), emb.contiguous(), k, 3)
    c_ = hip_decode.decode(heat, box, emb, k, 3)
    for key in a_:
        assert torch.equal(a_[key], b_[key]), key
        assert torch.equal(a_[key], c_[key]), key
    flat = emb.reshape(N, E, H * W)
    assert torch.equal(a_["embeddings"], torch.gather(flat, 2, a_["indices"].unsqueeze(1).expand(-1, E, -1)).permute(0, 2, 1))


def test_standalone_gathers_and_model_surface():
    ins = recipes.decode_inputs(8, (2, 3, 24, 32), 16)
    heat, box, reid = [t.cuda() for t in ins]
    m = cl.CenterNet({"name": "resnet34"}, {"name": "fpn"}, {"heatmap": {"num_classes": 3}, "box_2d": {}, "reid": {}}, "tracking")
    ref = decode_ref.decode_detections(ins[0].numpy(), ins[1].numpy(), 40, 3, reid=ins[2].numpy())
    d = m.gather_tracking2d(heat, box, reid, num_detections=40)
    assert set(d) == {"bboxes", "labels", "scores", "embeddings"} and d["labels"].dtype == torch.int64
    _check({"scores": d["scores"].cpu().numpy(), "indices": ref["indices"], "labels": d["labels"].cpu().numpy(),
            "boxes": d["bboxes"].cpu().numpy(), "embeddings": d["embeddings"].cpu().numpy()}, ref)
    # Gen-A per-head calls (fairmot.py:141-143)
    s, i, l = m.heads["heatmap"].gather_topk(heat, nms_kernel=3, num_detections=40)
    assert np.array_equal(i.cpu().numpy(), ref["indices"]) and np.array_equal(l.cpu().numpy(), ref["labels"])
    bb = m.heads["box_2d"].gather_at_indices(box, i, normalize_bbox=False, stride=4)
    assert np.array_equal(bb.cpu().numpy().view(np.uint32), ref["boxes"].view(np.uint32))
    ee = m.heads["reid"].gather_at_indices(reid, i)
    assert np.array_equal(ee.cpu().numpy(), ref["embeddings"])
    # Gen-B names (centernet.py:229-304); namedtuple accepted as the single argument (README.md:97-98)
    d2 = m.gather_detection2d((heat, box), num_detections=40)
    assert torch.equal(d2["bboxes"], d["bboxes"])
    m.num_detections = 40
    d3 = m.decode_detections(heat, box)
    assert set(d3) == {"boxes", "scores", "labels"} and torch.equal(d3["boxes"], d["bboxes"])
    bn = cl.CenterNet.gather_and_decode_boxes(box, i, normalize_boxes=True)
    refn = decode_ref.gather_and_decode_boxes(ins[1].numpy(), ref["indices"], normalize_boxes=True)
    assert np.array_equal(bn.cpu().numpy().view(np.uint32), refn.view(np.uint32))


def test_decode_argument_errors():
    heat = torch.rand(1, 2, 8, 8, device="cuda")
    box = torch.rand(1, 4, 8, 8, device="cuda")
    with pytest.raises(ValueError):
        hip_decode.decode(heat, box, None, 65, 3)            # k > H*W
    with pytest.raises(ValueError):
        hip_decode.decode(heat, box, None, 10, 4)            # even kernel (reference breaks too)
    with pytest.raises(ValueError):
        hip_decode.decode(heat, box[:, :3], None, 10, 3)


def test_pack_unpack_and_collate_single_process():
    ins = recipes.decode_inputs(2, (2, 3, 16, 16), 8)
    o = hip_decode.decode(*[t.cuda() for t in ins], 20, 3)
    dets = {"bboxes": o["boxes"], "scores": o["scores"], "labels": o["labels"], "embeddings": o["embeddings"]}
    rec = cl.pack_detections(dets)
    ref = decode_ref.pack_detections(o["boxes"].cpu().numpy(), o["scores"].cpu().numpy(), o["labels"].cpu().numpy(),
                                     o["embeddings"].cpu().numpy())
    assert np.array_equal(rec.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    u = cl.unpack_detections(rec)
    for key in dets:
        assert torch.equal(u[key], dets[key])
    assert cl.collate_detections(dets) is dets               # world size 1: no-op (eval/coco.py:11-13)


def test_collate_through_rccl_single_rank():
    """The real collective path (pack kernel -> RCCL all_gather_into_tensor on HIP memory -> unpack kernel) on the one GPU
    this box has: a world-size-1 "nccl" group with the gather forced.  The multi-rank ordering is covered on CPU (gloo)."""
    import socket
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        ins = recipes.decode_inputs(4, (3, 2, 16, 24), 8)
        o = hip_decode.decode(*[t.cuda() for t in ins], 25, 3)
        dets = {"bboxes": o["boxes"], "scores": o["scores"], "labels": o["labels"], "embeddings": o["embeddings"]}
        g = cl.collate_detections(dets, force=True)
        torch.cuda.synchronize()
        assert g is not dets
        for key in dets:
            assert torch.equal(g[key], dets[key]), key
        # the pipelined form: persistent slots, gather on the side stream behind an event, result one step behind submit
        c = cl.Collator(depth=2, force=True)
        batches = []
        for i in range(5):
            o = hip_decode.decode(*[t.cuda() for t in recipes.decode_inputs(10 + i, (3, 2, 16, 24), 8)], 25, 3)
            batches.append({"bboxes": o["boxes"], "scores": o["scores"], "labels": o["labels"], "embeddings": o["embeddings"]})
        pending, got = None, []
        for d in batches:
            h = c.submit(d)
            if pending is not None:
                got.append({k: v.clone() for k, v in c.result(pending).items()})
            pending = h
        got.append(c.result(pending))
        torch.cuda.synchronize()
        assert len(next(iter(c._slots.values()))) == 2 and len(c._slots) == 1          # two slots, allocated once
        for d, g in zip(batches, got):
            for key in d:
                assert torch.equal(g[key], d[key]), key
    finally:
        dist.destroy_process_group()
