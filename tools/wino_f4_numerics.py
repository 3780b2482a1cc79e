"""CPU numerics experiment (no GPU): error of Winograd F(4x4,3x3) vs F(2x2,3x3) against a float64 direct convolution on a head-block-like
layer (Cin = 256, post-ReLU activations, Kaiming x BN-scale weights), with the arithmetic of the fp16-split kernels emulated:
transforms in fp32, V and U rounded to 22 significant bits (hi + lo fp16 pieces), products exact, accumulation in fp32."""
import sys
import numpy as np

rng = np.random.default_rng(0)


def round_bits(a, bits=22):
    """round fp32 array to `bits` significant bits (emulates x*S = hi + lo with 11 + 11 bits)"""
    a = a.astype(np.float32)
    m, e = np.frexp(a.astype(np.float64))
    return (np.round(m * (1 << bits)) / (1 << bits) * np.exp2(e)).astype(np.float32)


def direct64(x, w):
    H, W, C = x.shape
    K = w.shape[0]
    xp = np.zeros((H + 2, W + 2, C)); xp[1:-1, 1:-1] = x
    y = np.zeros((H, W, K))
    for ky in range(3):
        for kx in range(3):
            y += xp[ky:ky + H, kx:kx + W].reshape(-1, C).astype(np.float64) @ w[:, ky, kx, :].astype(np.float64).T.reshape(C, K) .reshape(C, K) if False else \
                 (xp[ky:ky + H, kx:kx + W].reshape(-1, C) @ w[:, ky, kx, :].astype(np.float64).T).reshape(H, W, K)
    return y


def wino(x, w, m, pts=None, split=True):
    """F(m x m, 3x3), fp32 emulation.  Returns y [H,W,K] fp32."""
    if m == 2:
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    else:
        if pts == "std":
            BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
            G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
            AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
        else:
            # points 0, 1, -1, 1/2, -1/2(?)...: build by Cook-Toom for a point list
            BT, G, AT = cook_toom(pts)
    a = m + 2
    H, W, C = x.shape
    K = w.shape[0]
    # weights: transform in float64, store fp32 (then split -> 22 bits)
    U = np.einsum("ia,kabc,jb->ijkc", G, w.astype(np.float64), G).astype(np.float32)       # [a,a,K,C]
    if split:
        U = round_bits(U)
    th, tw = H // m, W // m
    xp = np.zeros((H + 2, W + 2, C), np.float32); xp[1:-1, 1:-1] = x
    y = np.zeros((H, W, K), np.float32)
    BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)
    # tiles: d [th,tw,a,a,C]
    d = np.stack([np.stack([xp[i:i + m * th:m, j:j + m * tw:m] for j in range(a)], 2) for i in range(a)], 2)   # [th,tw,a,a,C]
    # fp32 transform, sequential fp32 adds: emulate with two fp32 matmuls (each row is a short sum: rounding per partial sum differs slightly but same order of magnitude)
    t = np.einsum("ia,yxabc->yxibc", BT32, d).astype(np.float32)
    V = np.einsum("jb,yxibc->yxijc", BT32, t).astype(np.float32)
    if split:
        V = round_bits(V)
    # products exact, fp32 accumulation over C: use float32 matmul with chunks of 16 (fp32 accumulate across chunks)
    M = np.zeros((th, tw, a, a, K), np.float32)
    for c0 in range(0, C, 16):
        part = np.einsum("yxijc,ijkc->yxijk", V[..., c0:c0 + 16].astype(np.float64), U[..., c0:c0 + 16].astype(np.float64))
        M = (M.astype(np.float64) + part).astype(np.float32)          # one fp32 rounding per 16-channel MFMA (hardware accumulates in fp32 inside too, slightly pessimistic-free)
    s = np.einsum("ia,yxabk->yxibk", AT32, M).astype(np.float32)
    Y = np.einsum("jb,yxibk->yxijk", AT32, s).astype(np.float32)       # [th,tw,m,m,K]
    return Y.transpose(0, 2, 1, 3, 4).reshape(H, W, K)


def cook_toom(points):
    """F(4,3) matrices from 5 finite interpolation points + infinity (Lavin's construction via Vandermonde)."""
    import numpy.polynomial.polynomial as P
    n = 6
    pts = list(points)
    assert len(pts) == 5
    # AT [4 x 6]: rows i = p^i, last column = infinity (only for i = 3)
    AT = np.zeros((4, n)); 
    for j, p in enumerate(pts):
        for i in range(4):
            AT[i, j] = p ** i
    AT[3, 5] = 1
    # G [6 x 3]: row j = [1, p, p^2] / prod_{k != j}(p_j - p_k); last row = [0,0,1]
    G = np.zeros((n, 3))
    for j, p in enumerate(pts):
        den = np.prod([p - q for k, q in enumerate(pts) if k != j])
        G[j] = np.array([1, p, p * p]) / den
    G[5] = [0, 0, 1]
    # BT [6 x 6]: row j (finite) = coefficients of prod_{k != j}(x - p_k) (degree 4) padded; last row = coefficients of prod_k (x - p_k) (degree 5)
    BT = np.zeros((n, n))
    for j in range(5):
        c = np.array([1.0])
        for k, q in enumerate(pts):
            if k != j:
                c = P.polymul(c, [-q, 1.0])
        BT[j, :5] = c
    c = np.array([1.0])
    for q in pts:
        c = P.polymul(c, [-q, 1.0])
    BT[5, :6] = c
    return BT, G, AT


def main():
    H = W = 32; C = 256; K = 32
    # post-ReLU activations like a head block input: relu(N(0,1)) with ~50% zeros, a few large values
    x = np.maximum(rng.standard_normal((H, W, C)), 0).astype(np.float32) * rng.uniform(0.3, 3.0, C).astype(np.float32)
    w = (rng.standard_normal((K, 3, 3, C)) * np.sqrt(2.0 / (9 * C)) * rng.uniform(0.7, 1.3, (K, 1, 1, 1))).astype(np.float32)
    ref = direct64(x, w)
    scale = np.abs(ref).max()
    # sanity of cook_toom: compare against float64 Winograd
    for name, m, pts in [("F(2x2)", 2, None), ("F(4x4) std pts 0,+-1,+-2", 4, "std"), ("F(4x4) pts 0,1,-1,1/2,-1/2", 4, (0, 1, -1, .5, -.5)),
                         ("F(4x4) pts 0,1,-1,1/2,-2", 4, (0, 1, -1, .5, -2)), ("F(4x4) pts 0,1,-1,2,-1/2", 4, (0, 1, -1, 2, -.5)),
                         ("F(4x4) pts 0,1/2,-1/2,1,-1 (reordered)", 4, (0, .5, -.5, 1, -1)), ("F(4x4) pts 0,+-1/2,+-3/2", 4, (0, .5, -.5, 1.5, -1.5)), ("F(4x4) pts 0,+-3/4,+-3/2?", 4, (0, .75, -.75, 1.5, -1.5))]:
        for split in (False, True):
            y = wino(x, w, m, pts, split)
            e = np.abs(y - ref)
            print(f"{name:44s} split={int(split)}  max|err|/max|ref| = {e.max() / scale:.3e}   rms/max = {np.sqrt((e**2).mean()) / scale:.3e}")
    # direct fp32 accumulate for scale
    yd = np.zeros((H, W, K), np.float32)
    xp = np.zeros((H + 2, W + 2, C), np.float32); xp[1:-1, 1:-1] = x
    for ky in range(3):
        for kx in range(3):
            for c0 in range(0, C, 16):
                yd = (yd.astype(np.float64) + (xp[ky:ky + H, kx:kx + W, c0:c0 + 16].reshape(-1, 16).astype(np.float64) @ w[:, ky, kx, c0:c0 + 16].astype(np.float64).T).reshape(H, W, K)).astype(np.float32)
    e = np.abs(yd - ref)
    print(f"{'direct, fp32 accumulate per 16-chunk':44s}          max|err|/max|ref| = {e.max() / scale:.3e}   rms/max = {np.sqrt((e**2).mean()) / scale:.3e}")


main()
