"""CPU numerics experiment (no GPU): error of 1-D row-Winograd F(2,3) (winograd9 / winograd10) vs F(4,3) with the kernel rows folded into the
reduction, against a float64 direct convolution on a head-block-like layer — the arithmetic of the fp16-split kernels emulated: transforms in
fp32, V and U rounded to 22 significant bits (hi + lo fp16 pieces), products exact, fp32 accumulation per 16-channel MFMA.  (VERDICT r3 #1b.)"""
import numpy as np
import numpy.polynomial.polynomial as P

rng = np.random.default_rng(0)


def round_bits(a, bits=22):
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
            y += (xp[ky:ky + H, kx:kx + W].reshape(-1, C) @ w[:, ky, kx, :].astype(np.float64).T).reshape(H, W, K)
    return y


def cook_toom(points, m):
    """F(m,3) matrices from m + 1 finite interpolation points + infinity."""
    n = m + 2
    pts = list(points)
    assert len(pts) == n - 1
    AT = np.zeros((m, n))
    for j, p in enumerate(pts):
        for i in range(m):
            AT[i, j] = p ** i
    AT[m - 1, n - 1] = 1
    G = np.zeros((n, 3))
    for j, p in enumerate(pts):
        den = np.prod([p - q for k, q in enumerate(pts) if k != j])
        G[j] = np.array([1, p, p * p]) / den
    G[n - 1] = [0, 0, 1]
    BT = np.zeros((n, n))
    for j in range(n - 1):
        c = np.array([1.0])
        for k, q in enumerate(pts):
            if k != j:
                c = P.polymul(c, [-q, 1.0])
        BT[j, :n - 1] = c
    c = np.array([1.0])
    for q in pts:
        c = P.polymul(c, [-q, 1.0])
    BT[n - 1, :n] = c
    return BT, G, AT


def rescale(BT, G, AT):
    """Move row scales between BT and G so that max |BT row| == 1-ish powers of two (the kernel's V scale is a power of two anyway)."""
    return BT, G, AT


def wino_row(x, w, m, pts, split=True):
    BT, G, AT = cook_toom(pts, m)
    a = m + 2
    H, W, C = x.shape
    K = w.shape[0]
    U = np.einsum("ia,kyac->iykc", G, w.astype(np.float64)).astype(np.float32)           # [a, ky, K, C]
    if split:
        U = round_bits(U)
    tw = W // m
    xp = np.zeros((H + 2, W + 2, C), np.float32); xp[1:-1, 1:-1] = x
    BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)
    d = np.stack([xp[:, j:j + m * tw:m] for j in range(a)], 2)                          # [H+2, tw, a, C]
    V = np.einsum("ia,rxac->rxic", BT32.astype(np.float64), d.astype(np.float64)).astype(np.float32)       # one fp32 rounding per element (fma chains keep more)
    if split:
        V = round_bits(V)
    M = np.zeros((H, tw, a, K), np.float32)
    for c0 in range(0, C, 16):
        for ky in range(3):
            part = np.einsum("rxic,ikc->rxik", V[ky:ky + H, :, :, c0:c0 + 16].astype(np.float64), U[:, ky, :, c0:c0 + 16].astype(np.float64))
            M = (M.astype(np.float64) + part).astype(np.float32)
    Y = np.einsum("ja,rxak->rxjk", AT32.astype(np.float64), M.astype(np.float64)).astype(np.float32)        # [H, tw, m, K]
    return Y.reshape(H, W, K)


def main():
    H, W, C, K = 16, 64, 256, 32
    x = np.maximum(rng.standard_normal((H, W, C)), 0).astype(np.float32) * rng.uniform(0.3, 3.0, C).astype(np.float32)
    w = (rng.standard_normal((K, 3, 3, C)) * np.sqrt(2.0 / (9 * C)) * rng.uniform(0.7, 1.3, (K, 1, 1, 1))).astype(np.float32)
    ref = direct64(x, w)
    scale = np.abs(ref).max()
    cases = [("F(2,3) pts 0,1,-1", 2, (0, 1, -1)), ("F(4,3) pts 0,1,-1,2,-2 (textbook)", 4, (0, 1, -1, 2, -2)), ("F(4,3) pts 0,1,-1,1/2,-1/2", 4, (0, 1, -1, .5, -.5)),
             ("F(4,3) pts 0,1,-1,1/2,-2", 4, (0, 1, -1, .5, -2)), ("F(4,3) pts 0,1,-1,2,-1/2", 4, (0, 1, -1, 2, -.5)), ("F(4,3) pts 0,+-1/2,+-3/2", 4, (0, .5, -.5, 1.5, -1.5)),
             ("F(4,3) pts 0,+-3/4,+-3/2", 4, (0, .75, -.75, 1.5, -1.5)), ("F(4,3) pts 0,+-1/2,+-1", 4, (0, .5, -.5, 1, -1)), ("F(3,3) pts 0,1,-1,2", 3, (0, 1, -1, 2)),
             ("F(3,3) pts 0,1,-1,1/2", 3, (0, 1, -1, .5)), ("F(3,3) pts 0,1,-1,-1/2", 3, (0, 1, -1, -.5))]
    for name, m, pts in cases:
        if W % m:
            xx, ww, rr = x[:, :W // m * m], w, ref[:, :W // m * m]
        else:
            xx, ww, rr = x, w, ref
        if W % m:
            rr = direct64(xx, ww)
        for split in (False, True):
            y = wino_row(xx, ww, m, pts, split)
            e = np.abs(y - rr)
            print(f"{name:40s} split={int(split)}  max|err|/max|ref| = {e.max() / scale:.3e}   rms/max = {np.sqrt((e ** 2).mean()) / scale:.3e}")
    yd = np.zeros((H, W, K), np.float32)
    xp = np.zeros((H + 2, W + 2, C), np.float32); xp[1:-1, 1:-1] = x
    for c0 in range(0, C, 16):
        for ky in range(3):
            for kx in range(3):
                yd = (yd.astype(np.float64) + (xp[ky:ky + H, kx:kx + W, c0:c0 + 16].reshape(-1, 16).astype(np.float64) @ w[:, ky, kx, c0:c0 + 16].astype(np.float64).T).reshape(H, W, K)).astype(np.float32)
    e = np.abs(yd - ref)
    print(f"{'direct, fp32 accumulate per 16-chunk':40s}          max|err|/max|ref| = {e.max() / scale:.3e}   rms/max = {np.sqrt((e ** 2).mean()) / scale:.3e}")


main()
