"""CPU: host logic of the product package — config surface, model contract (re-stated from the reference's
tests/test_models.py, test_necks.py, test_backbones.py), C-ABI library load / exported symbols / argument
validation (no compute without a GPU), and the N>1 collate protocol over gloo (world_size 2)."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

import centernet_lightning_amd as cl
from centernet_lightning_amd import _lib
from centernet_lightning_amd.config import load_config, model_section

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------- C ABI
def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "centernet_gfx950.h")).read()
    declared = set(re.findall(r"\b(cnl_[a-z0-9_]+)\s*\(", header))
    declared -= {"cnl_conv_params", "cnl_decode_params"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/centernet_gfx950.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    assert lib.cnl_version() == _lib.ABI_VERSION == 13
    # ... and nothing else: the library is built with -fvisibility=hidden, its dynamic symbol table is the header (three C++ helpers used to leak)
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    if os.path.exists(nm):
        out = subprocess.run([nm, "-D", "--defined-only", _lib.lib_path()], capture_output=True, text=True, check=True).stdout
        defined = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in ("T", "W")}      # functions (the data symbols left are hipcc's kernel handles)
        extra = {n for n in defined - declared if not n.startswith(("__hip_", "_init", "_fini", "__bss", "_edata", "_end"))}
        assert not extra, sorted(extra)[:10]
    # the binding's parameter structs have the size the library was compiled with (cnl_sizeof_params: conv, decode, deconv)
    import ctypes as _ct
    assert [lib.cnl_sizeof_params(i) for i in range(4)] == [_ct.sizeof(_lib.ConvParams), _ct.sizeof(_lib.DecodeParams), _ct.sizeof(_lib.DeconvParams), 0]


def test_abi_error_convention_without_gpu():
    lib = _lib.load()
    assert lib.cnl_conv2d_nhwc_f32(None, None) == _lib.CNL_E_BAD_ARG
    assert "null" in _lib.last_error()
    p = _lib.ConvParams()
    p.x = p.w = p.bias = p.y = 0x1000
    p.N, p.H_in, p.W_in, p.Cin, p.Cout = 1, 8, 8, 3, 64
    p.KH = p.KW = 3
    p.stride, p.pad, p.ldx, p.ldy = 1, 1, 4, 64
    assert lib.cnl_conv2d_nhwc_f32(ctypes.byref(p), None) == _lib.CNL_E_UNSUPPORTED      # Cin % 32 != 0
    assert "multiple of 32" in _lib.last_error()
    with pytest.raises(ValueError):
        _lib.check(_lib.CNL_E_UNSUPPORTED, "x")
    d = _lib.DecodeParams()
    assert lib.cnl_decode_f32(ctypes.byref(d), None) == _lib.CNL_E_BAD_ARG
    assert lib.cnl_decode_workspace_bytes(2, 128, 128) >= 2 * 128 * 128 * 8
    ho, wo = ctypes.c_int32(), ctypes.c_int32()
    p.Cin = 64
    p.stride = 2
    assert lib.cnl_conv2d_out_hw(ctypes.byref(p), ctypes.byref(ho), ctypes.byref(wo)) == 0 and (ho.value, wo.value) == (4, 4)


def test_winograd_dispatcher_is_host_code_and_keeps_the_arithmetic_class_whatever_the_batch():
    """cnl_conv3x3_winograd_kernel / _variant are pure host functions of the parameters (no GPU needed): the arithmetic CLASS of a layer never
    depends on N; within the row-Winograd class a launch of at most 128 of winograd9's work items takes the bit-identical 4-row x 32-cout
    items (variant 11) — the rule the GPU tests pin bit for bit (tests/test_gpu_conv.py)."""
    lib = _lib.load()

    def ask(fn, N, Cin, H, W, Cout, algo=_lib.CNL_ALGO_AUTO, flags=0):
        p = _lib.ConvParams()
        p.N, p.H_in, p.W_in, p.Cin, p.Cout, p.KH, p.KW, p.stride, p.pad, p.ldx, p.ldy, p.flags, p.algo = N, H, W, Cin, Cout, 3, 3, 1, 1, Cin, Cout, flags, algo
        p.y = 1 << 20
        return fn(ctypes.byref(p))

    for shape in ((256, 128, 128, 256), (64, 128, 128, 64), (128, 64, 64, 128), (256, 32, 32, 256), (512, 16, 16, 512), (512, 19, 34, 512)):
        assert len({ask(lib.cnl_conv3x3_winograd_kernel, N, *shape) for N in (1, 2, 5, 32, 64)}) == 1, shape
    # winograd9 items: N x ceil(H / 8) x ceil(W / 64) x Cout / 64 (two images side by side on 32-pixel maps)
    assert [ask(lib.cnl_conv3x3_winograd_variant, N, 256, 128, 128, 256) for N in (1, 2, 32)] == [11, 9, 9]          # 128, 256, 4096 items
    assert [ask(lib.cnl_conv3x3_winograd_variant, N, 64, 128, 128, 64) for N in (1, 4, 5, 32)] == [11, 11, 9, 9]    # 32 per image
    assert [ask(lib.cnl_conv3x3_winograd_variant, N, 256, 32, 32, 256) for N in (1, 16, 17, 32)] == [11, 11, 9, 9]  # 16 per image pair
    assert ask(lib.cnl_conv3x3_winograd_variant, 1, 64, 64, 64, 512, flags=_lib.CNL_UPSAMPLE_IN) == 9               # not behind a folded upsample
    assert ask(lib.cnl_conv3x3_winograd_variant, 1, 64, 128, 128, 64, algo=_lib.CNL_ALGO_F32) == 2


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("CENTERNET_GFX950_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryError):
        _lib.load()
    monkeypatch.undo()
    _lib.load()


# ----------------------------------------------------------------------------- config surface
def test_configs_build(configs_dir):
    m = cl.build_centernet(os.path.join(configs_dir, "resnet34_simple.yaml"))
    assert type(m.neck).__name__ == "SimpleNeck" and m.num_classes == 80 and m.task == "detection"
    m = cl.build_centernet(os.path.join(configs_dir, "resnet34_fpn.yaml"))
    assert type(m.neck).__name__ == "FPNNeck"
    m = cl.build_centernet(os.path.join(configs_dir, "tracking_resnet34_fpn.yaml"))      # __base__ inheritance
    assert m.task == "tracking" and list(m.heads.keys()) == ["heatmap", "box_2d", "reid"] and m.num_classes == 2
    assert m.heads["reid"].depth == 1 and m.heads["reid"].out_channels == 64             # fairmot.py:20
    assert m.heads["heatmap"].out_conv.bias.data.eq(-2.19).all() and m.heads["box_2d"].out_conv.bias.data.eq(10).all()


def test_neck_params_nesting_and_ignored_sections():
    cfg = {"model": {"task": "detection", "backbone": {"name": "resnet34", "pretrained": True, "input_channels": 3},
                     "neck": {"name": "simple", "params": {"upsample_channels": [128, 64, 32], "upsample_type": "nearest",
                                                           "conv_type": "normal", "skip_kernel": 3}},
                     "output_heads": {"heatmap": {"num_classes": 20, "loss_function": "cornernet_focal"}, "box_2d": {}},
                     "optimizer": {"name": "SGD"}},
           "data": {"train": {}}, "trainer": {"gpus": 2}}
    m = cl.build_centernet(cfg)
    assert m.neck.out_channels == 32 and m.num_classes == 20
    assert model_section(cfg)["neck"]["upsample_channels"] == [128, 64, 32]


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference tree only in the build container")
@pytest.mark.parametrize("name", ["base_resnet34.yaml", "base_resnet34_fpn.yaml", "base_tracking_resnet34_fpn.yaml"])
def test_reference_yaml_files_build(name):
    m = cl.build_centernet(os.path.join("/root/reference/configs", name))
    assert m.output_stride == 4


def test_out_of_scope_options_raise():
    base = {"backbone": {"name": "resnet34"}, "neck": {"name": "fpn"}, "output_heads": {"heatmap": {"num_classes": 3}, "box_2d": {}}}
    for bad in ({"backbone": {"name": "mobilenet_v2"}}, {"neck": {"name": "nas_fpn"}}, {"neck": {"name": "fpn", "conv_type": "dilated"}},
                {"neck": {"name": "simple", "upsample_type": "conv_transpose", "deconv_kernel": 7}}):
        cfg = dict(base, **bad)
        with pytest.raises(ValueError):
            cl.CenterNet(cfg["backbone"], cfg["neck"], cfg["output_heads"], "detection")


# ----------------------------------------------------------------------------- model contract
@pytest.mark.parametrize("neck", ["simple", "fpn", "ida", "bifpn"])
def test_model_attributes_contract(neck):
    m = cl.CenterNet({"name": "resnet34"}, {"name": neck}, {"heatmap": {"num_classes": 20}, "box_2d": {}}, "detection")
    assert isinstance(m.output_stride, int) and m.output_stride == 4 and m.stride == 4       # tests/test_models.py:64
    assert m.task == "detection" and m.num_classes == 20
    assert m.backbone.out_channels == [64, 64, 128, 256, 512] and m.backbone.output_stride == 32
    assert m.neck.out_channels == 64 and m.neck.upsample_stride == 8                         # tests/test_necks.py:27-28
    assert not m.training
    with pytest.raises(RuntimeError):
        m.train()


def test_state_dict_uses_torchvision_and_generic_head_key_names():
    m = cl.build_centernet({"model": {"backbone": {"name": "resnet34"}, "neck": {"name": "fpn"},
                                      "output_heads": {"heatmap": {"num_classes": 80}, "box_2d": {}}}})
    keys = set(m.state_dict())
    for k in ("backbone.conv1.weight", "backbone.bn1.running_var", "backbone.layer1.0.conv1.weight", "backbone.layer2.0.downsample.0.weight",
              "backbone.layer2.0.downsample.1.running_mean", "backbone.layer4.2.bn2.bias", "neck.top_conv.bias",
              "neck.fuse.1.project.1.weight", "neck.fuse.0.output_conv.0.weight", "neck.fuse.2.output_conv.1.running_var",
              "heads.heatmap.block_1.conv.weight", "heads.heatmap.block_3.bn.weight", "heads.box_2d.out_conv.bias"):
        assert k in keys, k
    assert "neck.fuse.0.project.1.weight" not in keys                 # 256 -> 256: no projection (layers.py:152)
    n_backbone = sum(v.numel() for k, v in m.state_dict().items() if k.startswith("backbone.") and v.dim() == 4)
    assert abs(n_backbone - 21.26e6) < 0.1e6                          # ResNet-34 conv params (docs/experiments.md:24-27)
    n_heads = sum(p.numel() for p in m.heads.parameters())
    conv = lambda cin, cout: 9 * cin * cout + 2 * cout                # 3x3 no-bias conv + BN affine (meta.py:24-26)
    expect = 2 * (conv(64, 256) + 2 * conv(256, 256)) + (256 * 80 + 80) + (256 * 4 + 4)
    assert n_heads == expect                                          # with a 256-ch neck this is the 3.6 M of docs/experiments.md:27


def test_cpu_input_raises_no_fallback():
    m = cl.build_centernet({"model": {"backbone": {"name": "resnet34"}, "neck": {"name": "simple"},
                                      "output_heads": {"heatmap": {"num_classes": 4}, "box_2d": {}}}})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.gather_detection2d(torch.rand(1, 4, 16, 16), torch.rand(1, 4, 16, 16))
    with pytest.raises(RuntimeError):
        cl.pack_detections({"bboxes": torch.zeros(1, 2, 4), "scores": torch.zeros(1, 2), "labels": torch.zeros(1, 2, dtype=torch.int64)})


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "centernet-lightning_amd", "centernet_lightning_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+(oracle|decode_ref|ref_cpu|recipes)\b", src, re.M), fn


# ----------------------------------------------------------------------------- N>1 protocol (gloo, world_size 2)
def _collate_worker(rank, world, port, ret):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import decode_ref
    import recipes
    from centernet_lightning_amd import all_gather_records, shard_range
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        n_total, k, E = 6, 10, 4
        h, b, r = [t.numpy() for t in recipes.decode_inputs(21, (n_total, 3, 16, 16), E)]
        lo, hi = shard_range(n_total, rank, world)
        o = decode_ref.decode_detections(h[lo:hi], b[lo:hi], k, reid=r[lo:hi])
        rec = torch.from_numpy(decode_ref.pack_detections(o["boxes"], o["scores"], o["labels"], o["embeddings"]))
        full = all_gather_records(rec).numpy()
        g = decode_ref.decode_detections(h, b, k, reid=r)
        u = decode_ref.unpack_detections(full)
        ok = all(np.array_equal(u[key], g[key]) for key in ("boxes", "scores", "labels", "embeddings"))
        ret[rank] = bool(ok and full.shape == (n_total, k, 6 + E))
    finally:
        dist.destroy_process_group()


def _collator_worker(rank, world, port, ret):
    """The pipelined Collator (persistent slots, result one step behind submit) over gloo with CPU records."""
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
    from centernet_lightning_amd import Collator
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        c = Collator(depth=2)
        ok, pending, want = True, None, None
        for step in range(5):
            rec = torch.full((3, 4, 6), float(100 * step + rank))
            h = c.submit_records(rec)
            if pending is not None:
                got = c.result_records(pending)
                ok = ok and got.shape == (3 * world, 4, 6) and all(bool((got[3 * r:3 * r + 3] == want + r).all()) for r in range(world))
            pending, want = h, float(100 * step)
        got = c.result_records(pending)
        ok = ok and all(bool((got[3 * r:3 * r + 3] == want + r).all()) for r in range(world))
        ok = ok and len(c._slots) == 1 and len(next(iter(c._slots.values()))) == 2      # persistent: two slots, allocated once
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_pipelined_collator_gloo_world2():
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_collator_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get(0) is True and ret.get(1) is True


def test_collate_protocol_gloo_world2():
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_collate_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get(0) is True and ret.get(1) is True


def _world8_worker(rank, world, port, ret):
    """Rank order, slot reuse and the bench's own N > 1 plumbing (environment -> process group, max over ranks, whole-job throughput) at the
    north star's world size, on CPU over gloo: no 8-GPU node was ever available to this build."""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                       "CNL_BENCH_BACKEND": "gloo"})
    import bench
    from centernet_lightning_amd import Collator, shard_range
    try:
        r, w, lr = bench.setup_distributed(world)                      # what `bench.py --gpus 8` runs first on every rank
        ok = (r, w, lr) == (rank, world, rank) and dist.get_world_size() == world
        worst = bench.max_over_ranks(1.0 + 0.25 * rank, "cpu")         # the timed region ends with the slowest rank
        ok = ok and worst == 1.0 + 0.25 * (world - 1) and bench.job_throughput(32, world, 10, worst) == 32 * world * 10 / worst
        lo, hi = shard_range(512, rank, world)                         # C3: 512 images over 8 ranks
        ok = ok and (lo, hi) == (64 * rank, 64 * rank + 64)
        # `bench.py --gpus 8 --steps K --warmup W` (the driver's command): the default workload is C1 and the run ALSO schedules the two
        # configurations BASELINE.json defines on 8 GPUs — C3 (FPN, 64 per GPU) and C4 (tracking, 32 per GPU at 608 x 1088, k = 100)
        a = bench.parse_args(["--gpus", str(world), "--steps", "5", "--warmup", "2"])
        ok = ok and (a.gpus, a.config, a.batch, a.height, a.width, a.no_also) == (world, "simple", 32, 512, 512, False)
        also = bench.also_workloads(w, a.config, a.batch, a.height, a.width)
        ok = ok and [(s_["config"], s_["batch"], s_["height"], s_["width"], s_["k"]) for s_ in also] == [("fpn", 64, 512, 512, 100), ("tracking", 32, 608, 1088, 100)]
        ok = ok and also[0]["name"].startswith("C3 (512 images") and also[1]["name"].startswith("C4 (256 images")
        ok = ok and [s_["name"][:2] for s_ in bench.also_workloads(1, "simple", 32, 512, 512)] == ["C2", "C4"]
        ok = ok and bench.also_workloads(w, "fpn", 64, 512, 512) == []          # only the default (C1) run appends them
        c = Collator(depth=2)
        pending, want = None, None
        for step in range(3):                                          # three pipelined steps: result one step behind submit
            rec = torch.full((2, 5, 6), float(1000 * step + rank))
            h = c.submit_records(rec)
            if pending is not None:
                got = c.result_records(pending)
                ok = ok and got.shape == (2 * world, 5, 6) and all(bool((got[2 * q:2 * q + 2] == want + q).all()) for q in range(world))
            pending, want = h, float(1000 * step)
        got = c.result_records(pending)
        ok = ok and all(bool((got[2 * q:2 * q + 2] == want + q).all()) for q in range(world))
        ok = ok and len(c._slots) == 1 and len(next(iter(c._slots.values()))) == 2      # two persistent slots, reused
        # a handle is valid until its slot is re-claimed (depth - 1 further submits): a stale one raises instead of returning a later batch
        h0 = c.submit_records(torch.zeros(2, 5, 6)); c.submit_records(torch.ones(2, 5, 6)); c.submit_records(torch.ones(2, 5, 6))
        try:
            c.result_records(h0); ok = False
        except RuntimeError:
            pass
        # one slot: strictly sequential use works, a result collected behind a later submit raises (the same generation check)
        c1 = Collator(depth=1)
        g1 = c1.result_records(c1.submit_records(torch.full((2, 5, 6), float(rank))))
        ok = ok and all(bool((g1[2 * q:2 * q + 2] == q).all()) for q in range(world))
        h1 = c1.submit_records(torch.zeros(2, 5, 6)); c1.submit_records(torch.ones(2, 5, 6))
        try:
            c1.result_records(h1); ok = False
        except RuntimeError:
            pass
        ret[rank] = bool(ok)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_pipelined_collator_gloo_world8():
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_world8_worker, args=(8, port, ret), nprocs=8, join=True)
    assert all(ret.get(r) is True for r in range(8)), dict(ret)


def _plain_bench(extra_env, *argv):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"CNL_BENCH_BACKEND": "gloo", "CNL_BENCH_STUB": "1", "CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""}, **extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2 --steps 2 --warmup 1` as a PLAIN subprocess (no torchrun, no WORLD_SIZE: the form of the driver's N = 1 command):
    the script starts its two ranks itself (bench.self_launch -> torch.distributed.run on 127.0.0.1), they rendezvous over gloo, run the timed
    region with the pipelined Collator (records gathered in rank order, checked by every rank), and rank 0 prints the ONE JSON line.  No GPU
    here: CNL_BENCH_STUB=1 stands in for the model leg only (reference collective: eval/coco.py:10-18)."""
    r = _plain_bench({}, "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 64 and line["data"].startswith("STUB")
    assert line["value"] > 0 and abs(line["value"] - 64 * 2 / (line["ms_per_step"] * 2e-3)) / line["value"] < 0.01
    # the line proves which collective ran: backend, the world size the process group reports, one entry per rank — and says so in `parallelism`
    coll = line["collective"]
    assert coll["backend"] == "gloo" and coll["world_size"] == 2 and [d["rank"] for d in coll["devices"]] == [0, 1] and coll["rccl_version"] is None
    assert "gloo" in line["config"]["parallelism"] and "FUNCTIONAL" in line["config"]["parallelism"] and "RCCL all-gather of detections over" not in line["config"]["parallelism"]


def test_bench_self_launch_reports_a_failed_rank():
    r = _plain_bench({"CNL_BENCH_STUB_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_collate_is_noop_at_world_size_one():
    dets = {"bboxes": torch.zeros(1, 2, 4), "scores": torch.zeros(1, 2), "labels": torch.zeros(1, 2, dtype=torch.int64)}
    assert cl.collate_detections(dets) is dets                       # eval/coco.py:11-13 behaviour
    assert cl.shard_range(512, 3, 8) == (192, 256)
    with pytest.raises(ValueError):
        cl.shard_range(10, 0, 4)


def test_inline_asm_kernels_keep_valu_to_mfma_distance():
    """gfx950 needs two wait states between a VALU write of a VGPR and an MFMA reading it; the compiler does not look inside inline
    asm, so the kernels that split operands with asm VALU instructions are checked on their device assembly
    (tools/mfma_hazard_audit.py).  The checker itself is pinned on a two-line listing with and without the distance."""
    import shutil
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import mfma_hazard_audit as audit
    bad = "v_cvt_pkrtz_f16_f32 v5, v5, v14\nds_read2_b64 v[22:25], v9 offset1:1\nv_mfma_f32_32x32x16_f16 v[50:65], v[2:5], v[6:9], v[50:65]\n"
    assert audit.audit_asm(bad)[1] and not audit.audit_asm(bad.replace("ds_read2", "s_nop 0\nds_read2"))[1]
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    for name, (total, violations) in audit.audit_files().items():
        assert total > 50 and not violations, (name, violations[:3])


def test_absmax_arrays_are_one_cache_line_per_image():
    """ABI v10: image n's maximum sits at element n * cnl_absmax_stride() of an x_absmax / y_absmax array — 32 floats = one 128-byte line per
    image (device-scope atomics serialise per line: DESIGN.md 3.4); the host helpers pack / unpack that layout."""
    import torch
    lib = _lib.load()
    assert lib.cnl_absmax_stride() == 32 == _lib.absmax_stride()
    buf = _lib.absmax_pack(torch.tensor([1.0, 2.5, 0.0]))
    assert buf.numel() == 3 * 32 and float(buf[32]) == 2.5 and float(buf.sum()) == 3.5
    assert torch.equal(_lib.absmax_values(buf), torch.tensor([1.0, 2.5, 0.0]))
    assert _lib.absmax_buffer(4, device="cpu").numel() == 128


@pytest.mark.parametrize("src,kernels", [("winograd9.hip", 10), ("winograd10.hip", 8), ("winograd13.hip", 5)])
def test_winograd9_compiles_without_register_spills(src, kernels):
    """csrc/winograd9.hip sits at the edge of the register file (256 accumulator + 256 vector registers per lane): a spill inside its
    chunk loop comes back as a scratch load with a vmcnt(0) — a wait for every load in flight — and harmless-looking edits of the
    epilogue have produced 40-110 of them.  The device code of both variants (with / without residual) and of the
    weight transform must compile with ZERO spilled vector registers under the Makefile's flags.  csrc/winograd10.hip (two workgroups per CU: 256
    registers per wave, accumulators included) likewise — and it must keep its two waves per SIMD; csrc/winograd13.hip (192 accumulators + 254-256 vector registers) likewise."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "centernet-lightning_amd", "csrc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"), "-mllvm", "-pragma-unroll-threshold=4000000",
           "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", os.path.join(csrc, src), "-o", os.devnull]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    spills = [int(l.split("VGPRs Spill:")[1].split()[0]) for l in r.stderr.splitlines() if "VGPRs Spill:" in l]
    assert len(spills) == kernels and all(v == 0 for v in spills), spills
    if src == "winograd10.hip":
        occ = [int(l.split("Occupancy [waves/SIMD]:")[1].split()[0]) for l in r.stderr.splitlines() if "Occupancy [waves/SIMD]:" in l]
        assert occ == [2] * kernels, occ


def test_makefile_hands_winograd9_spill_count_to_the_dispatcher():
    """the product build records winograd9.hip's resource usage and generates build/w9_usage.h; winograd.hip reads
    CNL_W9_VGPR_SPILLS from it as a compile-time constant and sends winograd9's layers to the bit-identical winograd10.hip when it is not zero
    (a compiler that can only build the kernel with scratch traffic inside its chunk loop).  The shipped library was built with zero."""
    csrc = os.path.join(ROOT, "centernet-lightning_amd", "csrc")
    usage = os.path.join(csrc, "build", "w9_usage.h")
    if not os.path.exists(usage):
        pytest.skip("no build directory (library built elsewhere)")
    text = open(usage).read()
    assert "#define CNL_W9_VGPR_SPILLS 0" in text and "#define CNL_W9_KERNELS_CHECKED 8" in text
    src = open(os.path.join(csrc, "winograd.hip")).read()
    assert 'include "build/w9_usage.h"' in src and "CNL_W9_VGPR_SPILLS > 0" in src
