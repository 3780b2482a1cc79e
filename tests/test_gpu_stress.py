"""GPU: seeded randomised sweeps over shapes the hand-picked cases do not enumerate — ragged sizes, Cout tails, every flag
combination — for the conv kernels (direct MFMA, Winograd, transposed conv phases) against plain PyTorch CPU fp32.
Integer-valued operands make every product and partial sum exact in fp32, so any indexing / tiling / tail mistake is a hard
mismatch (torch.equal), independent of summation order."""
import ctypes
import random

import pytest
import torch
import torch.nn.functional as F

from test_gpu_conv import ref_conv, run_conv, run_winograd
from centernet_lightning_amd._lib import CNL_RELU, CNL_SIGMOID, CNL_UPSAMPLE_IN, CNL_UPSAMPLE_OUT_ADD

pytestmark = pytest.mark.gpu


def _ints(shape, lo, hi, g):
    return torch.randint(lo, hi + 1, shape, generator=g).float()


@pytest.mark.parametrize("seed", range(6))
def test_direct_conv_random_shapes_exact(seed):
    rnd = random.Random(1000 + seed)
    g = torch.Generator().manual_seed(seed)
    for _ in range(12):
        N = rnd.randint(1, 3)
        Cin = rnd.choice([32, 64, 96, 160])
        Cout = rnd.choice([1, 3, 4, 31, 32, 33, 64, 65, 80, 96, 97, 128, 130, 200])
        k, stride = rnd.choice([(1, 1), (1, 2), (3, 1), (3, 2)])
        H, W = rnd.randint(1, 23), rnd.randint(1, 23)
        mode = rnd.choice(["plain", "relu", "res", "up_in", "up_out"])
        flags, res = 0, None
        x = _ints((N, Cin, H, W), -2, 2, g)
        w = _ints((Cout, Cin, k, k), -2, 2, g)
        b = _ints((Cout,), -4, 4, g)
        if mode == "relu":
            flags = CNL_RELU
        elif mode == "up_in":
            flags = CNL_UPSAMPLE_IN | CNL_RELU
        ho = ((H * (2 if mode == "up_in" else 1)) + 2 * ((k - 1) // 2) - k) // stride + 1
        wo = ((W * (2 if mode == "up_in" else 1)) + 2 * ((k - 1) // 2) - k) // stride + 1
        if mode == "res":
            flags = CNL_RELU
            res = _ints((N, Cout, ho, wo), -9, 9, g)
        if mode == "up_out":
            if k != 1 or stride != 1:
                continue
            flags = CNL_UPSAMPLE_OUT_ADD | CNL_RELU
            res = _ints((N, Cout, 2 * ho, 2 * wo), -9, 9, g)
        out = run_conv(x, w, b, stride, flags, res)
        ref = ref_conv(x, w, b, stride, flags, res)
        assert torch.equal(out, ref), (seed, N, Cin, Cout, k, stride, H, W, mode, float((out - ref).abs().max()))


@pytest.mark.parametrize("seed", range(6))
def test_winograd_random_shapes_exact(seed):
    rnd = random.Random(2000 + seed)
    g = torch.Generator().manual_seed(100 + seed)
    for _ in range(10):
        N = rnd.randint(1, 3)
        Cin = rnd.choice([8, 16, 24, 64, 72])
        Cout = rnd.choice([1, 5, 32, 63, 64, 65, 100, 128, 129])
        H, W = rnd.randint(1, 37), rnd.randint(1, 37)
        mode = rnd.choice(["plain", "relu", "res", "up_in"])
        x = _ints((N, Cin, H, W), -3, 3, g)
        w = _ints((Cout, Cin, 3, 3), -2, 2, g) * 4              # multiples of 4: G g G^T stays integral
        b = _ints((Cout,), -5, 5, g)
        flags, res = 0, None
        if mode in ("relu", "res"):
            flags = CNL_RELU
        if mode == "up_in":
            flags = CNL_UPSAMPLE_IN
        up = 2 if mode == "up_in" else 1
        if mode == "res":
            res = _ints((N, Cout, H, W), -9, 9, g)
        out = run_winograd(x, w, b, flags, res)
        ref = ref_conv(x, w, b, 1, flags, res)
        assert tuple(out.shape) == (N, Cout, H * up, W * up)
        assert torch.equal(out, ref), (seed, N, Cin, Cout, H, W, mode, float((out - ref).abs().max()))


@pytest.mark.parametrize("seed", range(3))
def test_deconv_random_shapes_exact(seed):
    from centernet_lightning_amd import _lib, engine, params as P
    lib = _lib.load()
    rnd = random.Random(3000 + seed)
    g = torch.Generator().manual_seed(200 + seed)
    for _ in range(8):
        K = rnd.choice([2, 3, 4])
        C = rnd.choice([32, 64, 96])
        N, H, W = rnd.randint(1, 2), rnd.randint(1, 13), rnd.randint(1, 13)
        mod = P.DeconvBn(C, K, init_bilinear=False).eval()
        with torch.no_grad():
            mod.deconv.weight.copy_(_ints(tuple(mod.deconv.weight.shape), -2, 2, g))
            mod.bn.weight.fill_(2.0); mod.bn.bias.copy_(_ints((C,), -3, 3, g))
            mod.bn.running_mean.zero_(); mod.bn.running_var.fill_(1.0); mod.bn.eps = 3.0        # gamma / sqrt(1 + 3) = 1: BN folds to + bias, exactly
        x = _ints((N, C, H, W), -2, 2, g)
        with torch.no_grad():
            ref = F.relu(mod.bn(mod.deconv(x)))
        layer = engine._DeconvLayer(mod, torch.device("cuda:0"))
        xd = x.permute(0, 2, 3, 1).contiguous().cuda()
        y = torch.full((N, 2 * H, 2 * W, C), float("nan"), device="cuda")
        p = _lib.DeconvParams()
        p.x, p.w, p.bias, p.y = xd.data_ptr(), layer.w.data_ptr(), layer.b.data_ptr(), y.data_ptr()
        p.N, p.H_in, p.W_in, p.Cin, p.Cout, p.K = N, H, W, C, C, K
        p.ldx, p.ldy, p.flags = C, C, CNL_RELU
        _lib.check(lib.cnl_deconv2x_nhwc_f32(ctypes.byref(p), None))
        assert torch.equal(y.cpu().permute(0, 3, 1, 2), ref), (seed, K, C, N, H, W)
