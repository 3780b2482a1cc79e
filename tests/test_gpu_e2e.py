"""GPU: end-to-end parity of CenterNet.forward / get_encoded_outputs / gather_detection2d with the CPU oracle
(oracle/ref_cpu.py + oracle/decode_ref.py) on the BASELINE configs at sizes the CPU finishes in seconds.

Level A (decode on identical tensors): bit-exact — test_gpu_decode.py.
Level B (this file): conv accumulation order differs from oneDNN, so logits / scores / boxes must agree within
rtol = atol = 1e-4 and top-k indices wherever the oracle's neighbouring score gap exceeds that tolerance."""
import ctypes
import os

import numpy as np
import pytest
import torch

import decode_ref
import recipes
import ref_cpu
import centernet_lightning_amd as cl

pytestmark = pytest.mark.gpu
CONFIGS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "centernet-lightning_amd", "configs")
TOL = 1e-4          # north star: outputs within 1e-4 fp32
GAP = 2e-5          # an oracle top-k position is "well separated" when both neighbouring score gaps exceed this


def build(cfg_name, mutate=None, **options):
    torch.manual_seed(0)
    model = cl.build_centernet(os.path.join(CONFIGS, cfg_name))
    sd = ref_cpu.synth_state_dict(model.state_dict(), seed=0, calib_shape=(2, 3, 128, 128))
    if mutate is not None:
        mutate(sd)
    model.load_state_dict(sd)
    if options:
        model.set_kernel_options(**options)
    return model.cuda(), sd


def compare_detections(dets, indices, ref_heat, ref_box, k, ref_reid=None):
    """End-to-end detections against the oracle's, EVERY rank pinned (VERDICT r5 #4b; rounds 3-5 pinned only the ranks whose neighbouring oracle scores lie
    more than GAP apart — 80-90 % of them).  Two fp32 summation orders may legitimately swap detections whose scores differ by less than the conv
    round-off, and the k-th may trade places with the (k+1)-th; so the oracle decodes k + 64 candidates from ITS maps and, rank by rank, the GPU's detection
    must BE one of them — same pixel — whose oracle score lies within GAP of the oracle's score at that rank; labels, boxes and embeddings are then compared
    through that matching.  Returns the mask of ranks where GPU and oracle agree on the pixel outright."""
    s, l, b = dets["scores"].cpu().numpy(), dets["labels"].cpu().numpy(), dets["bboxes"].cpu().numpy()
    gi = indices.cpu().numpy()
    N, HW = s.shape[0], ref_heat.shape[2] * ref_heat.shape[3]
    ref = decode_ref.decode_detections(ref_heat, ref_box, k, 3, reid=ref_reid)
    ext = decode_ref.decode_detections(ref_heat, ref_box, min(HW, k + 64), 3, reid=ref_reid)
    np.testing.assert_allclose(s, ref["scores"], rtol=TOL, atol=TOL)
    same = gi == ref["indices"]
    pinned = 0
    for n in range(N):
        assert len(set(gi[n].tolist())) == k                                     # k distinct pixels
        where = {int(ix): j for j, ix in enumerate(ext["indices"][n])}
        for r in range(k):
            j = where.get(int(gi[n, r]))
            assert j is not None, f"image {n} rank {r}: pixel {gi[n, r]} is not among the oracle's first {len(where)} candidates"
            assert abs(float(ext["scores"][n, j]) - float(ref["scores"][n, r])) <= GAP, (n, r, j, float(ext["scores"][n, j]), float(ref["scores"][n, r]))
            assert l[n, r] == ext["labels"][n, j], (n, r)
            np.testing.assert_allclose(b[n, r], ext["boxes"][n, j], rtol=TOL, atol=TOL * 4)      # boxes are in pixels (x stride 4)
            if ref_reid is not None:
                np.testing.assert_allclose(dets["embeddings"][n, r].cpu().numpy(), ext["embeddings"][n, j], rtol=TOL, atol=TOL)
            pinned += 1
    print(f"compare_detections: safe fraction {pinned / (N * k):.3f} of {N * k} ranks (tolerance-aware matching: every rank pinned); "
          f"identical pixel at the same rank: {same.mean():.3f}")
    assert pinned == N * k
    return same


@pytest.mark.parametrize("cfg,shape", [("resnet34_simple.yaml", (2, 3, 128, 160)), ("resnet34_fpn.yaml", (2, 3, 160, 128)),
                                       ("tracking_resnet34_fpn.yaml", (1, 3, 96, 160))])
def test_forward_and_decode_match_cpu_oracle(cfg, shape):
    model, sd = build(cfg)
    x = recipes.images(1234, shape)
    ref_logits = ref_cpu.forward(sd, x, sigmoid=False)
    ref_sig = ref_cpu.forward(sd, x, sigmoid=True)
    xd = x.cuda()
    enc = model.get_encoded_outputs(xd)
    assert list(enc.keys()) == list(ref_logits.keys())
    for name, r in ref_logits.items():
        o = enc[name]
        assert tuple(o.shape) == tuple(r.shape) and o.shape[2] == shape[2] // model.output_stride
        torch.testing.assert_close(o.cpu(), r, rtol=TOL, atol=TOL)
    out = model(xd)                                           # namedtuple, heatmap post-sigmoid
    heat, box = out[0], out[1]
    assert float(heat.min()) >= 0 and float(heat.max()) <= 1                  # tests/test_models.py:93-95
    torch.testing.assert_close(heat.cpu(), ref_sig["heatmap"], rtol=TOL, atol=TOL)
    heat_std = float(ref_sig["heatmap"].std())
    assert heat_std > 1e-3, "degenerate synthetic heatmap: the comparison would be meaningless"
    k = 50
    if model.task == "tracking":
        dets = model.gather_tracking2d(out, num_detections=k)
    else:
        dets = model.gather_detection2d(out, num_detections=k)
    idx = cl.decode.decode(heat, box, None, k, 3, stride=model.output_stride)["indices"]
    compare_detections(dets, idx, ref_sig["heatmap"].numpy(), ref_sig["box_2d"].numpy(), k, ref_sig["reid"].numpy() if model.task == "tracking" else None)
    # Level A on the GPU's own tensors: indices bit-exact against the oracle decode of the same bytes
    refA = decode_ref.decode_detections(heat.cpu().numpy(), box.cpu().numpy(), k, 3,
                                        reid=out[2].cpu().numpy() if model.task == "tracking" else None)
    assert np.array_equal(idx.cpu().numpy(), refA["indices"])                  # indices bit-exact on identical bytes
    assert np.array_equal(dets["scores"].cpu().numpy(), refA["scores"])
    assert np.array_equal(dets["labels"].cpu().numpy(), refA["labels"])
    assert np.array_equal(dets["bboxes"].cpu().numpy().view(np.uint32), refA["boxes"].view(np.uint32))
    if model.task == "tracking":
        assert np.array_equal(dets["embeddings"].cpu().numpy(), refA["embeddings"])


def test_c0_config_512(request):
    """BASELINE C0: configs/base_resnet34.yaml-equivalent, 1x3x512x512."""
    model, sd = build("resnet34_simple.yaml")
    x = recipes.images(1234, (1, 3, 512, 512))
    ref = ref_cpu.forward(sd, x, sigmoid=True)
    heat, box = model(x.cuda())
    assert tuple(heat.shape) == (1, 80, 128, 128) and tuple(box.shape) == (1, 4, 128, 128)
    torch.testing.assert_close(heat.cpu(), ref["heatmap"], rtol=TOL, atol=TOL)
    torch.testing.assert_close(box.cpu(), ref["box_2d"], rtol=TOL, atol=TOL)
    dets = model.gather_detection2d(heat, box)
    assert tuple(dets["bboxes"].shape) == (1, 100, 4) and dets["labels"].dtype == torch.int64
    idx = cl.decode.decode(heat, box, None, 100, 3, stride=model.output_stride)["indices"]
    compare_detections(dets, idx, ref["heatmap"].numpy(), ref["box_2d"].numpy(), 100)


def test_batch_shard_equals_full_batch():
    """The N>1 partitioning: running each contiguous shard separately gives the same bytes as the
    full batch (images are independent; kernels are batch-invariant)."""
    model, _ = build("resnet34_fpn.yaml")
    x = recipes.images(7, (4, 3, 128, 128)).cuda()
    full = model.gather_detection2d(model(x), num_detections=30)
    parts = []
    for r in range(2):
        lo, hi = cl.shard_range(4, r, 2)
        parts.append(model.gather_detection2d(model(x[lo:hi]), num_detections=30))
    for key in full:
        assert torch.equal(full[key], torch.cat([p[key] for p in parts], dim=0)), key


def test_batch_shard_equals_full_batch_with_unequal_images():
    """The same with images of very different magnitude in one batch and maps large enough for the fp16-split kernel (whose
    input scale is per image for exactly this reason)."""
    model, _ = build("resnet34_simple.yaml")
    x = (recipes.images(9, (4, 3, 256, 256)) * torch.tensor([1.0, 0.02, 30.0, 1.0]).view(4, 1, 1, 1)).cuda()
    full = model.get_encoded_outputs(x)
    for r in range(4):
        part = model.get_encoded_outputs(x[r:r + 1])
        for key in full:
            assert torch.equal(full[key][r:r + 1], part[key]), (key, r)


def test_channels_last_input_and_weight_reload():
    model, sd = build("resnet34_simple.yaml")
    x = recipes.images(3, (1, 3, 128, 128)).cuda()
    a = model.get_encoded_outputs(x)
    b = model.get_encoded_outputs(x.contiguous(memory_format=torch.channels_last))
    for key in a:
        assert torch.equal(a[key], b[key])
    sd2 = {k: (v * 1.5 if k == "heads.box_2d.out_conv.bias" else v) for k, v in sd.items()}
    model.load_state_dict(sd2)
    c = model.get_encoded_outputs(x)
    assert not torch.equal(a["box_2d"], c["box_2d"]) and torch.equal(a["heatmap"], c["heatmap"])


def test_direct_and_winograd_paths_agree():
    """The same model through the two conv implementations (winograd=False: every conv on the direct implicit-GEMM kernels;
    default: 3x3/stride-1 layers on Winograd): both within 1e-4 of the CPU oracle and ~1e-5 of each other."""
    x = recipes.images(11, (2, 3, 128, 128))
    model_d, sd = build("resnet34_fpn.yaml", winograd=False)
    out_d = model_d.get_encoded_outputs(x.cuda())
    assert not any("winograd" in L.what for plan in model_d._engine.plans.values() for L in plan.launches)
    model_w, _ = build("resnet34_fpn.yaml")
    out_w = model_w.get_encoded_outputs(x.cuda())
    assert sum("winograd" in L.what for plan in model_w._engine.plans.values() for L in plan.launches) >= 30
    ref = ref_cpu.forward(sd, x, sigmoid=False)
    for name in ref:
        torch.testing.assert_close(out_d[name].cpu(), ref[name], rtol=TOL, atol=TOL)
        torch.testing.assert_close(out_w[name].cpu(), ref[name], rtol=TOL, atol=TOL)
        torch.testing.assert_close(out_w[name], out_d[name], rtol=2e-5, atol=2e-5)


def test_absmax_handover_matches_own_pass():
    """The fp16-split Winograd launches scale their input by a power of two taken from the image's maximum magnitude.  The engine
    hands that maximum over from the producing launch (cnl_conv_params.x_absmax / y_absmax); without the hand-over each launch
    makes its own pass over its input.  Both give the same network outputs up to fp32 rounding (the handed-over maximum may cover
    a superset of the consumer's channels, i.e. a scale that differs by a power of two; and the direct convs take the fp16-split
    kernel only with the hand-over, the fp32 matrix-core one without), and the slots really are filled."""
    x = recipes.images(13, (2, 3, 256, 256)).cuda()
    model_h, sd = build("resnet34_fpn.yaml")
    out_h = model_h.get_encoded_outputs(x)
    plan = next(iter(model_h._engine.plans.values()))
    wired = [L for L in plan.launches if getattr(L.args, "x_absmax", None)]
    assert plan.absmax is not None and len(wired) >= 10 and bool((plan.absmax[..., 0] > 0).all())
    model_o, _ = build("resnet34_fpn.yaml", absmax_handover=False)
    out_o = model_o.get_encoded_outputs(x)
    assert next(iter(model_o._engine.plans.values())).absmax is None
    ref = ref_cpu.forward(sd, x.cpu(), sigmoid=False)
    for name in ref:
        torch.testing.assert_close(out_h[name], out_o[name], rtol=TOL, atol=TOL)
        torch.testing.assert_close(out_h[name].cpu(), ref[name], rtol=TOL, atol=TOL)


def _feature_errors(cfg, shape, algo, mutate=None, x=None):
    """max |feature - float64 oracle| / max |float64 oracle| at the neck output and at every head's last 256-channel block output (the
    tensors out_conv reads), for the HIP path under `algo` — and for the CPU fp32 oracle itself (algo = "cpu")."""
    x = recipes.images(4242, shape) if x is None else x
    if algo == "cpu":
        model, sd = build(cfg, mutate)
        _, _, neck, heads = ref_cpu.forward(sd, x, sigmoid=False, return_intermediates="heads")
    else:
        model, sd = build(cfg, mutate, algo=algo, reuse_buffers=False)
        model.get_encoded_outputs(x.cuda())
        torch.cuda.synchronize()
        plan = model._engine.plan_for(x.cuda(), sigmoid=False)
        nb, nh, nw, nc, nup = plan.neck_out
        neck = plan.tensor(nb)[..., :nc].permute(0, 3, 1, 2).cpu()
        if nup:                                                       # a pending nearest upsample is folded into the heads
            neck = torch.nn.functional.interpolate(neck, scale_factor=2, mode="nearest")
        heads = {}
        for name, (buf, ld, off, c, _, _, _) in plan.head_features.items():
            heads[name] = plan.tensor(buf)[..., off:off + c].permute(0, 3, 1, 2).cpu()
    _, _, neck64, heads64 = ref_cpu.forward_float64(sd, x, sigmoid=False, return_intermediates="heads")
    errs = {"neck": float((neck.double() - neck64).abs().max() / neck64.abs().max())}
    for name in heads64:
        errs["head." + name] = float((heads[name].double() - heads64[name]).abs().max() / heads64[name].abs().max())
    return errs


@pytest.mark.parametrize("cfg,shape", [("resnet34_simple.yaml", (2, 3, 512, 512)),              # C1
                                       ("resnet34_fpn.yaml", (2, 3, 512, 512)),                 # C2 / C3
                                       ("tracking_resnet34_fpn.yaml", (2, 3, 608, 1088))])      # C4
def test_feature_level_error_against_float64(cfg, shape):
    """The sharp end-to-end gate.  The outputs behind out_conv (sigma = 0.01 weights, sigmoid' = 0.09) hide feature errors
    by three orders of magnitude, so this test looks at the FEATURES: the neck output and each head's last block output, against the
    float64 oracle, as max |err| / max |ref|, for the arithmetic classes of the plan:
      f32   every conv on the fp32 matrix cores (no split operands)          — the yardstick
      auto  the default: fp16-split matrix cores, Winograd F(2x2)            — must be <= 1.25 x f32 (+ 1e-6)
    and all of them <= 1e-4, the path's tolerance (fp32 rounding through 33 conv layers is itself ~1e-5 of the maximum: the CPU
    oracle's own distance from float64 is printed and used as the second yardstick)."""
    e = {a: _feature_errors(cfg, shape, a) for a in ("f32", "auto", "cpu")}
    print("feature errors vs float64 (max err / max ref):", cfg, e)
    for key in e["f32"]:
        assert e["auto"][key] <= 1.25 * e["f32"][key] + 1e-6, (key, e)
        for a in ("f32", "auto"):
            assert e[a][key] <= 1e-4, (a, key, e)


def _trained_checkpoint_like(sd):
    """BatchNorm scales as a trained checkpoint has them: every BN gamma times 10^U(-2, 0.5) per channel, 5 % of the channels dead
    (gamma = beta = 0: the folded filter and its bias are exactly zero) — the FOLDED conv weights then spread over 2.5 decades per output
    channel and so do the activations the next layer reads; running statistics re-calibrated so that the network stays O(1)."""
    g = torch.Generator().manual_seed(77)
    for k in list(sd):
        if k.endswith("running_var"):
            base = k[: -len("running_var")]
            c = sd[k].numel()
            s = torch.pow(10.0, torch.rand(c, generator=g) * 2.5 - 2.0)
            s[torch.randperm(c, generator=g)[: max(1, c // 20)]] = 0.0
            sd[base + "weight"].mul_(s)
            sd[base + "bias"].mul_((s > 0).float())
    ref_cpu.forward(sd, torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(99)) * 4.2 - 2.1, stats=True)


def test_feature_level_error_with_trained_checkpoint_like_scales():
    """VERDICT r2 #4, end to end: per-output-channel scales over 2.5 decades with 5 % dead channels (BN gamma, folded into the conv
    weights), inputs in the range of a normalised image (negative values) — the fp16-split plan must stay within 1.25 x the fp32
    matrix-core plan's distance from the float64 oracle at the neck output and at every head's last block, and within 1e-4."""
    cfg, shape = "resnet34_simple.yaml", (2, 3, 256, 256)
    x = recipes.images(515, shape) * 4.2 - 2.1                        # (x / 255 - mean) / std of an 8-bit image spans about [-2.1, 2.6]
    e = {a: _feature_errors(cfg, shape, a, mutate=_trained_checkpoint_like, x=x) for a in ("f32", "auto", "cpu")}
    print("feature errors vs float64, trained-checkpoint-like scales:", e)
    for key in e["f32"]:
        assert e["auto"][key] <= 1.25 * max(e["f32"][key], e["cpu"][key]) + 1e-6, (key, e)
        assert e["auto"][key] <= 1e-4 and e["f32"][key] <= 1e-4, (key, e)


def test_two_streams_do_not_share_plan_state():
    """Plans (arena, absmax slots, patched output pointers) are per stream: the same model run concurrently on two streams gives,
    on each, the bytes of a lone run."""
    model, _ = build("resnet34_simple.yaml")
    xa = recipes.images(21, (2, 3, 256, 256)).cuda()
    xb = (recipes.images(22, (2, 3, 256, 256)) * 7.0).cuda()
    ra, rb = model.get_encoded_outputs(xa), model.get_encoded_outputs(xb)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(s1):
            oa = model.get_encoded_outputs(xa)
        with torch.cuda.stream(s2):
            ob = model.get_encoded_outputs(xb)
        torch.cuda.synchronize()
        for k in ra:
            assert torch.equal(oa[k], ra[k]) and torch.equal(ob[k], rb[k]), k
    assert len({key[-1] for key in model._engine.plans}) == 3          # default stream + two side streams


def test_forward_from_two_threads_on_the_default_stream():
    """Serving code calls model(x) from several Python threads, all on the default stream: the engine serialises the host-side enqueue
    (a plan patches its output pointers per call), so every thread gets the bytes of a single-threaded call."""
    import threading
    model, _ = build("resnet34_simple.yaml")
    xs = [recipes.images(40 + i, (2, 3, 96, 96)).cuda() for i in range(4)]
    want = [{k: v.clone() for k, v in model.get_encoded_outputs(x).items()} for x in xs]
    got, errs = [None] * 4, []

    def work(i):
        try:
            for _ in range(10):
                o = model.get_encoded_outputs(xs[i])
            got[i] = o
        except Exception as e:                                   # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.cuda.synchronize()
    assert not errs, errs
    for i in range(4):
        for k in want[i]:
            assert torch.equal(got[i][k], want[i][k]), (i, k)


@pytest.mark.parametrize("cfg,shape", [("resnet34_simple.yaml", (1, 3, 512, 512)), ("resnet34_fpn.yaml", (2, 3, 256, 256)),
                                       ("tracking_resnet34_fpn.yaml", (1, 3, 224, 416))])
def test_split_small_latency_mode_matches_oracle(cfg, shape):
    """KernelOptions(split_small=True): the small maps of a small batch run as split-reduction direct convs.  Same parity bars as the
    default plan: outputs within 1e-4 of the CPU oracle, features fp32-grade against float64; and the split launches are really there."""
    model, sd = build(cfg, split_small=True, reuse_buffers=False)
    x = recipes.images(77, shape)
    ref = ref_cpu.forward(sd, x, sigmoid=False)
    enc = model.get_encoded_outputs(x.cuda())
    for name, r in ref.items():
        torch.testing.assert_close(enc[name].cpu(), r, rtol=TOL, atol=TOL)
    plan = model._engine.plan_for(x.cuda(), sigmoid=False)
    what = [L.what for L in plan.launches]
    assert sum("[split x" in w for w in what) >= 5, what
    assert not any("[split x" in w for w in what if w.startswith("heads.") and "out_conv" not in w and shape[2] >= 512)
    _, _, neck64, heads64 = ref_cpu.forward_float64(sd, x, sigmoid=False, return_intermediates="heads")
    nb, _, _, nc, nup = plan.neck_out
    neck = plan.tensor(nb)[..., :nc].permute(0, 3, 1, 2).cpu()
    if nup:
        neck = torch.nn.functional.interpolate(neck, scale_factor=2, mode="nearest")
    assert float((neck.double() - neck64).abs().max() / neck64.abs().max()) <= 5e-5
    # a batch large enough to fill the chip takes the default kernels under the same option
    model.get_encoded_outputs(recipes.images(78, (16,) + tuple(shape[1:])).cuda()) if shape[2] <= 256 else None


@pytest.mark.parametrize("cfg,shape", [("resnet34_simple.yaml", (1, 3, 512, 512)), ("resnet34_fpn.yaml", (1, 3, 256, 256)),
                                       ("tracking_resnet34_fpn.yaml", (2, 3, 96, 160))])
def test_latency_class_matches_oracle(cfg, shape):
    """KernelOptions(latency=True): the 3x3 / stride-1 layers run on winograd10.hip's 4-row x 32-cout work items (variant 11).
    Same parity bars as the default plan — outputs within 1e-4 of the CPU oracle, features fp32-grade against float64 — the class really is
    in the plan, it does not depend on the batch (shard == full batch bit for bit), and the DEFAULT plan of the same model is untouched by it
    (bit for bit what a model that never saw the option gives)."""
    import ctypes
    model, sd = build(cfg, latency=True, reuse_buffers=False)
    x = recipes.images(79, shape)
    ref = ref_cpu.forward(sd, x, sigmoid=False)
    enc = model.get_encoded_outputs(x.cuda())
    for name, r in ref.items():
        torch.testing.assert_close(enc[name].cpu(), r, rtol=TOL, atol=TOL)
    plan = model._engine.plan_for(x.cuda(), sigmoid=False)
    lib = plan.lib
    wino = [L for L in plan.launches if L.fn is lib.cnl_conv3x3_winograd_f32]
    variants = [lib.cnl_conv3x3_winograd_variant(ctypes.byref(L.args)) for L in wino]
    # every eligible layer is on the 4-row x 32-cout items; what stays on winograd9: launches behind a folded upsample (the neck stages and first
    # head blocks of the simple neck) and the block that carries a folded out_conv (cnl_conv_params.fuse_w: winograd9's epilogue has it)
    assert sum(v == 11 for v in variants) >= 20, variants
    assert all(v == 11 or (v == 9 and ((L.args.flags & 4) or L.args.fuse_w)) for v, L in zip(variants, wino)), variants
    _, _, neck64, heads64 = ref_cpu.forward_float64(sd, x, sigmoid=False, return_intermediates="heads")
    nb, _, _, nc, nup = plan.neck_out
    neck = plan.tensor(nb)[..., :nc].permute(0, 3, 1, 2).cpu()
    if nup:
        neck = torch.nn.functional.interpolate(neck, scale_factor=2, mode="nearest")
    assert float((neck.double() - neck64).abs().max() / neck64.abs().max()) <= 5e-5
    # batch invariance inside the class: three copies of the batch give three times the same bits
    x3 = torch.cat([x, x, x], dim=0).cuda()
    enc3 = model.get_encoded_outputs(x3)
    n = shape[0]
    for name in enc:
        assert torch.equal(enc3[name][:n], enc[name]) and torch.equal(enc3[name][2 * n:], enc[name]), name
    # the default plan of the same weights: untouched by the option
    base, _ = build(cfg)
    want = base.get_encoded_outputs(x.cuda())
    model.set_kernel_options(latency=False)
    got = model.get_encoded_outputs(x.cuda())
    for name in want:
        assert torch.equal(got[name], want[name]), name


def test_in_place_weight_edit_is_noticed_without_refresh():
    """model.backbone.load_state_dict(...) / in-place edits after the first forward must not run on stale packed weights."""
    model, sd = build("resnet34_simple.yaml")
    x = recipes.images(3, (1, 3, 128, 128)).cuda()
    a = model.get_encoded_outputs(x)
    bsd = {k[len("backbone."):]: v * (1.1 if k.endswith("layer4.2.bn2.weight") else 1.0) for k, v in sd.items() if k.startswith("backbone.")}
    model.backbone.load_state_dict(bsd, strict=False)                 # the README's torchvision-weights route: a SUBMODULE load
    b = model.get_encoded_outputs(x)
    assert not torch.equal(a["heatmap"], b["heatmap"])
    with torch.no_grad():
        model.heads["box_2d"].out_conv.bias.add_(1.0)
    c = model.get_encoded_outputs(x)
    torch.testing.assert_close(c["box_2d"], b["box_2d"] + 1.0, rtol=0, atol=1e-5)
    assert torch.equal(c["heatmap"], b["heatmap"])
    # a REPLACED parameter (another tensor object: `module.bias = nn.Parameter(...)`, load_state_dict(assign=True)) is noticed too
    oc = model.heads["box_2d"].out_conv
    oc.bias = torch.nn.Parameter(oc.bias.detach().clone() + 2.0)
    d = model.get_encoded_outputs(x)
    torch.testing.assert_close(d["box_2d"], c["box_2d"] + 2.0, rtol=0, atol=1e-5)


def test_arena_reuse_shrinks_the_footprint_and_keeps_the_bytes():
    """Liveness-based buffer reuse: same outputs as with every intermediate kept, in a fraction of the memory."""
    x = recipes.images(8, (2, 3, 256, 256)).cuda()
    model_r, _ = build("resnet34_fpn.yaml")
    model_k, _ = build("resnet34_fpn.yaml", reuse_buffers=False)
    a, b = model_r.get_encoded_outputs(x), model_k.get_encoded_outputs(x)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    pr, pk = model_r._engine.plan_for(x, sigmoid=False), model_k._engine.plan_for(x, sigmoid=False)
    assert pk.arena_bytes == pk.bytes_without_reuse and pr.arena_bytes < 0.5 * pk.arena_bytes, (pr.arena_bytes, pk.arena_bytes)


@pytest.mark.parametrize("cfg,shape,k", [("resnet34_simple.yaml", (32, 3, 512, 512), 100),          # BASELINE C1
                                         ("resnet34_fpn.yaml", (64, 3, 512, 512), 100),             # C2 / one rank of C3
                                         ("tracking_resnet34_fpn.yaml", (32, 3, 608, 1088), 300)])  # one rank of C4
def test_full_size_properties(cfg, shape, k):
    """BASELINE.json's full per-GPU sizes, through size-independent properties (the CPU oracle needs minutes here):
    image i of the batch == the same image alone (bit-equal: kernels are batch-invariant, sub-batching included);
    the oracle agrees on a 2-image slice; decode output is sorted, in range, and consistent with the maps it was cut from."""
    model, sd = build(cfg)
    x = recipes.images(99, shape).cuda()
    out = model(x)
    N, _, H, W = shape
    h, w = H // 4, W // 4
    assert tuple(out[0].shape) == (N, model.num_classes, h, w) and tuple(out[1].shape) == (N, 4, h, w)
    for i in (0, N // 2 + 1, N - 1):
        single = model(x[i:i + 1])
        for a, b in zip(out, single):
            assert torch.equal(a[i:i + 1], b), i
    ref = ref_cpu.forward(sd, x[:2].cpu(), sigmoid=True)
    for name, o in zip(ref, out):
        torch.testing.assert_close(o[:2].cpu(), ref[name], rtol=TOL, atol=TOL)
    dets = (model.gather_tracking2d if model.task == "tracking" else model.gather_detection2d)(out, num_detections=k)
    s, l, b = dets["scores"], dets["labels"], dets["bboxes"]
    assert tuple(s.shape) == (N, k) and bool((s[:, :-1] >= s[:, 1:]).all())                       # sorted descending
    assert int(l.min()) >= 0 and int(l.max()) < model.num_classes and bool((b[..., 2:] >= b[..., :2]).all())
    full = cl.decode.decode(out[0], out[1], out[2] if model.task == "tracking" else None, k, 3, stride=model.output_stride)
    idx = full["indices"]
    assert int(idx.min()) >= 0 and int(idx.max()) < h * w
    assert all(len(torch.unique(idx[n])) == k for n in range(0, N, 7))                             # k distinct peaks per image
    heat_flat = out[0].flatten(2)                                                                  # score == heat[label, index]
    picked = heat_flat.gather(2, idx.unsqueeze(1).expand(-1, heat_flat.shape[1], -1)).gather(1, l.unsqueeze(1)).squeeze(1)
    assert torch.equal(picked, s)
    again = cl.decode.gather_boxes(out[1], idx, stride=model.output_stride)                        # boxes re-gathered at the indices
    assert torch.equal(again, b)
    if model.task == "tracking":
        assert torch.equal(cl.decode.gather_embeddings(out[2], idx), dets["embeddings"])


def test_large_batch_is_split_below_the_4gib_addressing_limit(monkeypatch):
    """Engine.max_batch: a batch whose fused head buffer would exceed 4 GiB runs as contiguous sub-batches with identical bytes."""
    model, _ = build("resnet34_fpn.yaml")
    x = recipes.images(5, (3, 3, 64, 96)).cuda()
    full = model.get_encoded_outputs(x)
    assert model._engine.max_batch(512, 512) >= 64 and 32 <= model._engine.max_batch(608, 1088) < 128
    monkeypatch.setattr(type(model._engine), "max_batch", lambda self, H, W: 2)
    split = model.get_encoded_outputs(x)
    for k in full:
        assert torch.equal(full[k], split[k])


def test_buffer_over_the_addressing_limit_retries_with_a_smaller_sub_batch_and_plans_are_bounded(monkeypatch):
    """A plan buffer max_batch() did not price (a neck option's wider tensor) must not be addressed past 4 GiB: Plan raises
    BufferTooLarge, the engine splits further and the bytes stay those of the one-batch run.  The plan cache is an LRU of max_plans."""
    from centernet_lightning_amd import engine as E
    model, _ = build("resnet34_simple.yaml")
    x = recipes.images(6, (4, 3, 64, 64)).cuda()
    full = model.get_encoded_outputs(x)
    eng = model._engine
    # the widest tensor of a 4-image plan at 64x64 (first head blocks per head under this limit): [4, 16, 16, 256] fp32 = 1 MiB; allow a little less
    monkeypatch.setattr(E, "ADDRESS_LIMIT", 4 * 16 * 16 * 256 * 4 - 1)
    monkeypatch.setattr(type(eng), "max_batch", lambda self, H, W: 64)           # the a-priori estimate sees nothing wrong
    eng.plans.clear()
    split = model.get_encoded_outputs(x)
    assert eng._sub_override[(64, 64)] == 2 and all(k[0] == 2 for k in eng.plans)
    for k in full:
        assert torch.equal(full[k], split[k])
    monkeypatch.setattr(E, "ADDRESS_LIMIT", 16)
    with pytest.raises(E.BufferTooLarge):
        model.get_encoded_outputs(x[:1])
    monkeypatch.undo()
    eng.invalidate()
    eng.max_plans = 2
    for n in (1, 2, 3, 1):
        model.get_encoded_outputs(x[:n])
    assert [k[0] for k in eng.plans] == [3, 1]                   # least recently used first; N = 2 was evicted


def test_preprocess_uint8_bit_exact_and_feeds_forward():
    """uint8 HWC -> normalised fp32 (cnl_normalize_u8_nhwc_f32) is bit-exact against the numpy restatement of A.Normalize, and
    the channels_last result goes through forward() like a contiguous NCHW tensor of the same values."""
    model, _ = build("resnet34_simple.yaml")
    g = torch.Generator().manual_seed(9)
    for shape in [(2, 64, 96, 3), (1, 33, 47, 3)]:
        u8 = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
        out = model.preprocess_uint8(u8.cuda())
        ref = decode_ref.normalize_u8(u8.numpy())
        assert tuple(out.shape) == (shape[0], 3, shape[1], shape[2])
        assert np.array_equal(out.permute(0, 2, 3, 1).cpu().numpy().view(np.uint32), ref.view(np.uint32))
    u8 = torch.randint(0, 256, (2, 64, 96, 3), generator=g, dtype=torch.uint8)
    x = model.preprocess_uint8(u8.cuda())
    a = model.get_encoded_outputs(x)
    b = model.get_encoded_outputs(x.contiguous())
    for k in a:
        assert torch.equal(a[k], b[k])


def test_resize_uint8_bit_exact_against_oracle():
    """cnl_resize_bilinear_u8 = cv2.resize(INTER_LINEAR) on uint8 frames (A.Resize, README.md:84): bit-exact against the numpy restatement
    of OpenCV's fixed-point rule, for down- and up-scaling, odd sizes and the identity; plus hand-checkable known answers."""
    model, _ = build("resnet34_simple.yaml")
    g = torch.Generator().manual_seed(31)
    for (shape, oh, ow) in [((2, 1080 // 4, 1920 // 4, 3), 128, 128), ((1, 33, 47, 3), 64, 96), ((2, 40, 24, 3), 40, 24),
                            ((1, 17, 29, 1), 5, 7), ((1, 64, 64, 3), 152, 272)]:
        u8 = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
        out = model.resize_uint8(u8.cuda(), oh, ow)
        ref = decode_ref.resize_bilinear_u8(u8.numpy(), oh, ow)
        assert tuple(out.shape) == (shape[0], oh, ow, shape[3])
        assert np.array_equal(out.cpu().numpy(), ref), (shape, oh, ow)
    row = torch.tensor([[[[0], [100]]]], dtype=torch.uint8).cuda()                    # 1 x 2 row -> width 4: 0, 25, 75, 100
    assert model.resize_uint8(row, 1, 4).flatten().tolist() == [0, 25, 75, 100]
    const = torch.full((1, 5, 6, 3), 137, dtype=torch.uint8).cuda()
    assert bool((model.resize_uint8(const, 11, 13) == 137).all())


@pytest.mark.parametrize("cfg", ["resnet34_simple.yaml", "resnet34_fpn.yaml"])
def test_forward_uint8_is_bit_identical_to_normalize_then_forward(cfg):
    """The uint8 stem (cnl_stem_conv7x7_u8: A.Normalize applied to the staged patch inside the kernel, no fp32 image in HBM) gives the
    bits of preprocess_uint8() + forward(): same arithmetic, same order.  Ragged sizes exercise the zero padding of the NORMALISED
    image (an out-of-image slot must stay 0, not (0 - mean) / std)."""
    model, _ = build(cfg)
    g = torch.Generator().manual_seed(17)
    for shape in [(2, 128, 160, 3), (1, 96, 224, 3)]:
        u8 = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8).cuda()
        a = model(model.preprocess_uint8(u8))
        b = model.forward_uint8(u8)
        for ta, tb in zip(a, b):
            assert torch.equal(ta, tb), shape
    frames = torch.randint(0, 256, (2, 270, 480, 3), generator=g, dtype=torch.uint8).cuda()      # "1080p / 4" frames -> A.Resize -> network
    a = model(model.preprocess_uint8(model.resize_uint8(frames, 128, 128)))
    b = model.forward_uint8(frames, resize=(128, 128))
    for ta, tb in zip(a, b):
        assert torch.equal(ta, tb)
    dets = model.gather_detection2d(b, num_detections=20)
    assert tuple(dets["bboxes"].shape) == (2, 20, 4)


def test_torchscript_export_traces_saves_and_replays_bit_equal(tmp_path):
    """tools/export.py:7-12 (to_torchscript(method="trace")): the forward traces to ONE node, centernet_gfx950::forward; the traced
    module replays bit-equal, survives torch.jit.save / load, and the saved file runs in a FRESH process that merely imports the
    package (the op rebuilds the model from the embedded config + the saved tensors)."""
    import subprocess
    import sys
    model, _ = build("resnet34_fpn.yaml")
    x = recipes.images(5, (2, 3, 128, 160)).cuda()
    ref = model(x)
    traced = cl.export_torchscript(model, save_path=str(tmp_path / "m.pt"), example_inputs=x)
    assert "centernet_gfx950::forward" in str(traced.graph)
    out = traced(x)
    assert len(out) == 2 and all(torch.equal(a, b) for a, b in zip(out, ref))
    x2 = recipes.images(6, (1, 3, 96, 96)).cuda()                      # another shape through the same traced module
    assert all(torch.equal(a, b) for a, b in zip(traced(x2), model(x2)))
    loaded = torch.jit.load(str(tmp_path / "m.pt"))
    assert all(torch.equal(a, b) for a, b in zip(loaded(x), ref))
    torch.save(x.cpu(), tmp_path / "x.pt")
    torch.save([t.cpu() for t in ref], tmp_path / "ref.pt")
    code = (
        "import sys, torch\n"
        f"sys.path.insert(0, {os.path.join(os.path.dirname(CONFIGS))!r})\n"
        "import centernet_lightning_amd\n"
        f"m = torch.jit.load({str(tmp_path / 'm.pt')!r})\n"
        f"x = torch.load({str(tmp_path / 'x.pt')!r}).cuda()\n"
        f"ref = torch.load({str(tmp_path / 'ref.pt')!r})\n"
        "out = m(x)\n"
        "assert all(torch.equal(a.cpu(), b) for a, b in zip(out, ref))\n"
        "print('fresh-process replay ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fresh-process replay ok" in r.stdout, r.stderr[-2000:]
    with pytest.raises(NotImplementedError):
        cl.export_onnx(model, "x.onnx")


def test_bench_contract_one_json_line():
    """bench.py's driver contract on a tiny workload: the LAST stdout line is the one JSON object, with the contract's keys, the `roofline`
    block measured in the run and `cpu_baseline` absent only because this invocation switches it off."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "2", "--height", "128",
                          "--width", "128", "--no-cpu-baseline", "--no-variants", "--no-accuracy"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    d = json.loads(lines[-1])
    assert sum(ln.lstrip().startswith("{") for ln in lines) == 1
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "decode", "decode_p50_ms", "decode_gpu_ms", "latency_ms_N1"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "images/s" and d["value"] > 0 and abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3e-3)) < 0.02 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["decode"]["gpu_ms"] > 0 and set(d["latency_ms_N1"]) >= {"default", "split_small"}


def test_check_range_guard_for_the_split_arithmetic():
    """KernelOptions(check_range=True) (VERDICT r5 #4c): every fp16-split launch's input is inspected before it runs — per image, the weakest populated 8 x 8 tile
    against the image maximum — and a SplitRangeError names the launch beyond 10^5; the statistic itself on crafted tensors; a normal forward passes with the
    guard on and gives the same bytes as without it; an activation map with a region 10^7 below its maximum (forced through a hook on the plan's first
    split launch) is refused instead of silently losing the path's 1e-4."""
    from centernet_lightning_amd import engine as E
    x = torch.ones(3, 16, 24, 8, device="cuda")
    x[1, :8, :8] = 1e-6                                  # one tile a million times below the rest
    x[2] = 0                                             # an all-zero image: exact, ratio 1
    x[2, 0, 0, 0] = 5.0
    r = E.split_range_ratio(x).cpu()
    assert float(r[0]) == 1.0 and abs(float(r[1]) - 1e6) / 1e6 < 1e-5 and float(r[2]) == 1.0
    model, sd = build("resnet34_simple.yaml")
    xi = recipes.images(99, (2, 3, 128, 128)).cuda()
    plain = [t.clone() for t in model(xi)]
    model.set_kernel_options(check_range=True)
    guarded = model(xi)
    assert all(torch.equal(a, b) for a, b in zip(plain, guarded))
    plan = model._engine.plan_for(xi, sigmoid=True)
    L = next(l for l in plan.launches if isinstance(l.args, E.ConvParams) and l.args.x_absmax and l.fn is plan.lib.cnl_conv3x3_winograd_f32)
    prev = plan._check_split_range

    def check(launch):
        if launch is L:
            xs = plan.input_view(L.args)
            xs[0, : L.args.H_in // 2] *= 1e-7              # the upper half of image 0's activations: 10^7 below the rest
        return prev(launch)
    plan._check_split_range = check
    with pytest.raises(E.SplitRangeError, match="below the image's maximum"):
        model(xi)
    plan._check_split_range = prev


def test_odd_width_input_runs_the_2d_winograd_kernels_end_to_end():
    """since round 5 no launch of the default C1 / C2 / C4 plans takes the 2-D F(2x2,3x3) split kernels (csrc/winograd5.hip / winograd6.hip) — they
    remain what AUTO takes on maps of ODD width with long channel loops, where the row kernels' two-pixel tiles do not apply.  Inputs are multiples of 32
    (reference docs/implementation.md:52), so such maps appear in layer4 of frames whose width is an odd multiple of 32: a 160 x 224 frame gives 5 x 7 maps at
    512 channels.  The plan must route them there, and forward + decode must match the CPU oracle like every other shape."""
    model, sd = build("resnet34_simple.yaml")
    x = recipes.images(77, (2, 3, 160, 224))
    ref = ref_cpu.forward(sd, x, sigmoid=True)
    xd = x.cuda()
    heat, box = model(xd)
    assert tuple(heat.shape) == tuple(ref["heatmap"].shape)
    torch.testing.assert_close(heat.cpu(), ref["heatmap"], rtol=TOL, atol=TOL)
    torch.testing.assert_close(box.cpu(), ref["box_2d"], rtol=TOL, atol=TOL)
    plan = model._engine.plan_for(xd, sigmoid=True)
    lib = plan.lib
    variants = [lib.cnl_conv3x3_winograd_variant(ctypes.byref(L.args)) for L in plan.launches if L.fn is lib.cnl_conv3x3_winograd_f32]
    assert any(v in (5, 6) for v in variants), variants
    k = 40
    dets = model.gather_detection2d((heat, box), num_detections=k)
    idx = cl.decode.decode(heat, box, None, k, 3, stride=model.output_stride)["indices"]
    compare_detections(dets, idx, ref["heatmap"].numpy(), ref["box_2d"].numpy(), k)
