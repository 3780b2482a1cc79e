#!/usr/bin/env python
"""CPU emulation of the split arithmetic (scaled two-way fp16 split of both operands, three cross terms, fp32 accumulation in 16-channel steps) for 1-D Winograd
F(2,3) and F(4,3) along x with the kernel rows in the reduction — how round 6 chose winograd13.hip's interpolation points and found where its error comes from.
    python tools/wino_f43_numerics.py sets     error against float64 for several point sets (3 seeds, 256 -> 64 channels, 8 x 64 map, post-ReLU inputs)
    python tools/wino_f43_numerics.py abl      ablation: exact V transform / float64 accumulation / float64 output transform / no split
Needs sympy (Cook-Toom matrices in exact arithmetic)."""
import sys
import numpy as np, torch
import sympy as sp
torch.manual_seed(0); np.random.seed(0)
def split16(v, S):
    vs = (v*S).astype(np.float32)
    hi = vs.astype(np.float16)
    r = (vs - hi.astype(np.float32)).astype(np.float32)
    # RZ16 of r
    lo = r.astype(np.float16)
    lo32 = lo.astype(np.float32)
    adj = np.abs(lo32) > np.abs(r)
    lo = np.where(adj, np.nextafter(lo, np.float16(0)), lo)
    return hi, lo
def pow2scale(m, top):
    # m*S in [2^(top-1), 2^top)
    e = np.floor(np.log2(m)) + 1
    return 2.0 ** (top - e)
def winograd_mats(points):
    # Cook-Toom F(m, r) with given finite points + infinity; returns AT (m x n), G (n x r), BT (n x n), in float64
    import sympy as sp
    pts = [sp.Rational(p) for p in points]
    n = len(pts) + 1; r = 3; m = n - r + 1
    # Following the standard construction (wincnn)
    x = sp.symbols('x')
    def At(a, m, n): return sp.Matrix(m, n, lambda i, j: a[j]**i if j < n-1 else (1 if i == m-1 else 0))
    a = pts
    def Tfn(a, n): return sp.Matrix(n, n, lambda i, j: 1 if i==j else 0)
    # use wincnn formulas
    def A_(a, m, n): return sp.Matrix(m, n, lambda i, j: (a[j]**i if j < n-1 else (1 if i == m-1 else 0)))
    f = lambda i: sp.prod([ (a[i]-a[k]) for k in range(n-1) if k != i])
    Fd = [f(i) for i in range(n-1)]
    AT = A_(a, m, n)
    Gm = sp.Matrix(n, r, lambda i, j: (a[i]**j / Fd[i]) if i < n-1 else (1 if j == r-1 else 0))
    # B^T: from polynomial products
    Mx = sp.prod([(x - a[i]) for i in range(n-1)])
    BT = sp.zeros(n, n)
    for i in range(n-1):
        poly = sp.Poly(sp.expand(Mx / (x - a[i])), x)  # cancel
        poly = sp.Poly(sp.cancel(Mx/(x-a[i])), x)
        co = poly.all_coeffs()[::-1]
        for j, c in enumerate(co): BT[i, j] = c
    co = sp.Poly(sp.expand(Mx), x).all_coeffs()[::-1]
    for j, c in enumerate(co): BT[n-1, j] = c
    return np.array(AT.tolist(), dtype=np.float64), np.array(Gm.tolist(), dtype=np.float64), np.array(BT.tolist(), dtype=np.float64)

def check(AT, G, BT):
    g = np.random.randn(3); d = np.random.randn(BT.shape[0])
    m = AT.shape[0]
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i+k]*g[k] for k in range(3)) for i in range(m)])
    return np.abs(y-ref).max()

def run(Cin=256, Cout=64, H=8, W=64, trials=1, relu=True, wscale='he'):
    x = torch.randn(1, Cin, H, W, dtype=torch.float64)
    if relu: x = torch.relu(x)
    w = torch.randn(Cout, Cin, 3, 3, dtype=torch.float64) * (2.0/(9*Cin))**0.5
    ref = torch.nn.functional.conv2d(x, w, padding=1)
    f32 = torch.nn.functional.conv2d(x.float(), w.float(), padding=1).double()
    mx = ref.abs().max().item()
    print("fp32 direct err/max: %.3e" % ((f32-ref).abs().max().item()/mx))
    xn = x[0].numpy(); wn = w.numpy()
    xp = np.pad(xn, ((0,0),(1,1),(1,5)))
    res = {}
    for name, pts, mout in [("F23", [0,1,-1], 2), ("F43 std", [0,1,-1,2,-2], 4), ("F43 half", [0,1,-1,sp_half,-sp_half], 4), ("F43 mixed", [0,-1,1,sp_half,-2],4)]:
        AT, G, BT = winograd_mats(pts)
        assert check(AT,G,BT) < 1e-9, name
        n = BT.shape[0]
        # row scaling to make BT integer-ish? keep as is (fp32 transform)
        U = np.einsum('pk,oiyk->pyoi', G, wn).astype(np.float32)       # [p][ky][co][ci] computed in f64 then rounded to f32
        mU = np.abs(U).max(axis=(0,1,3))
        Su = pow2scale(mU, 13)[None,None,:,None]
        Uh, Ul = split16(U, Su.astype(np.float32))
        vmax = np.abs(BT).sum(1).max() * np.abs(xn).max()
        Sv = np.float32(pow2scale(vmax, 14))
        nt = W // mout
        out = np.zeros((Cout, H, W))
        # V[p][row][tile][ci]
        V = np.zeros((n, H+2, nt, Cin), dtype=np.float32)
        for t in range(nt):
            d = xp[:, :, t*mout:t*mout+n].astype(np.float32)   # [ci][row][n]
            # fp32 transform with sequential fma-ish accumulate
            v = np.zeros((n, Cin, H+2), dtype=np.float32)
            for p in range(n):
                acc = np.zeros((Cin, H+2), dtype=np.float32)
                for k in range(n):
                    if BT[p,k] != 0: acc = (acc + np.float32(BT[p,k]) * d[:,:,k]).astype(np.float32)
                v[p] = acc
            V[:, :, t, :] = v.transpose(0,2,1)
        Vh, Vl = split16(V, Sv)
        Uh32, Ul32, Vh32, Vl32 = [a.astype(np.float32) for a in (Uh, Ul, Vh, Vl)]
        Y = np.zeros((n, H, nt, Cout), dtype=np.float32)
        for p in range(n):
            for y in range(H):
                acc = np.zeros((nt, Cout), dtype=np.float32)
                for ky in range(3):
                    r = y + ky
                    for c0 in range(0, Cin, 16):
                        sl = slice(c0, c0+16)
                        acc += Vh32[p, r][:, sl] @ Ul32[p, ky][:, sl].T
                        acc += Vl32[p, r][:, sl] @ Uh32[p, ky][:, sl].T
                        acc += Vh32[p, r][:, sl] @ Uh32[p, ky][:, sl].T
                Y[p, y] = acc
        # output transform in fp32
        isc = (1.0/(Su[0,0,:,0]*Sv)).astype(np.float32)
        o = np.zeros((mout, H, nt, Cout), dtype=np.float32)
        for i in range(mout):
            acc = np.zeros((H, nt, Cout), dtype=np.float32)
            for p in range(n):
                if AT[i,p] != 0: acc = (acc + np.float32(AT[i,p]) * Y[p]).astype(np.float32)
            o[i] = acc * isc
        outw = o.transpose(3,1,2,0).reshape(Cout, H, W)
        err = np.abs(outw - ref[0].numpy())
        print("%-10s err/max: %.3e  rms/max %.3e   |BT|rowsum max %.1f" % (name, err.max()/mx, np.sqrt((err**2).mean())/mx, np.abs(BT).sum(1).max()))

H=8; W=64; Cin=256; Cout=64
def run(pts, mout, seed, vexact=False, nosplit=False, acc64=False, out64=False, name=""):
    torch.manual_seed(seed)
    x = torch.relu(torch.randn(1, Cin, H, W, dtype=torch.float64))
    w = torch.randn(Cout, Cin, 3, 3, dtype=torch.float64) * (2.0/(9*Cin))**0.5
    ref = torch.nn.functional.conv2d(x, w, padding=1)[0].numpy(); mx = np.abs(ref).max()
    xn = x[0].numpy(); wn = w.numpy()
    xp = np.pad(xn, ((0,0),(1,1),(1,5)))
    AT, G, BT = winograd_mats(pts); n = BT.shape[0]
    U = np.einsum('pk,oiyk->pyoi', G, wn).astype(np.float32)
    mU = np.abs(U).max(axis=(0,1,3)); Su = pow2scale(mU, 13)[None,None,:,None].astype(np.float32)
    vmax = np.abs(BT).sum(1).max() * np.abs(xn).max(); Sv = np.float32(pow2scale(vmax, 14))
    nt = W // mout
    V = np.zeros((n, H+2, nt, Cin), dtype=np.float64 if vexact else np.float32)
    for t in range(nt):
        d = xp[:, :, t*mout:t*mout+n].astype(np.float32)
        for p in range(n):
            if vexact: acc = sum(BT[p,k]*d[:,:,k].astype(np.float64) for k in range(n))
            else:
                acc = np.zeros((Cin, H+2), dtype=np.float32)
                for k in range(n):
                    if BT[p,k] != 0: acc = (acc + np.float32(BT[p,k]) * d[:,:,k]).astype(np.float32)
            V[p, :, t, :] = acc.T
    V = V.astype(np.float32)
    if nosplit:
        terms = [((V*Sv).astype(np.float64), (U*Su).astype(np.float64))]
    else:
        Vh, Vl = split16(V, Sv); Uh, Ul = split16(U, Su)
        f = np.float64 if acc64 else np.float32
        terms = [(Vh.astype(f), Ul.astype(f)), (Vl.astype(f), Uh.astype(f)), (Vh.astype(f), Uh.astype(f))]
    Y = np.zeros((n, H, nt, Cout), dtype=np.float64 if (acc64 or nosplit) else np.float32)
    for p in range(n):
        for y in range(H):
            acc = np.zeros((nt, Cout), dtype=Y.dtype)
            for ky in range(3):
                for c0 in range(0, Cin, 16):
                    sl = slice(c0, c0+16)
                    for (a, b) in terms: acc += a[p, y+ky][:, sl] @ b[p, ky][:, sl].T
            Y[p, y] = acc
    isc = (1.0/(Su[0,0,:,0]*Sv))
    fo = np.float64 if out64 else np.float32
    o = np.zeros((mout, H, nt, Cout), dtype=fo)
    for i in range(mout):
        acc = np.zeros((H, nt, Cout), dtype=fo)
        for p in range(n):
            if AT[i,p] != 0: acc = (acc + fo(AT[i,p]) * Y[p].astype(fo)).astype(fo)
        o[i] = acc * isc.astype(fo)
    outw = o.transpose(3,1,2,0).reshape(Cout, H, W)
    err = np.abs(outw - ref)
    return err.max()/mx, np.sqrt((err**2).mean())/mx
h = sp.Rational(1,2)
sets = {"F23": ([0,1,-1],2), "std": ([0,1,-1,2,-2],4), "mixed": ([0,-1,1,h,-2],4), "half": ([0,1,-1,h,-h],4),
        "m2": ([0,1,-1,2,-h],4), "m3": ([0,1,-1,sp.Rational(3,2),-sp.Rational(3,2)],4), "m4":([0,1,-1,sp.Rational(1,2),-sp.Rational(3,2)],4),
        "m5": ([0,h,-h,sp.Rational(3,2),-sp.Rational(3,2)],4), "m6": ([0,h,-1,2,-2],4)}
if sys.argv[1] == "sets":
    for k,(pts,m) in sets.items():
        r = [run(pts,m,s) for s in range(3)]
        print("%-6s max %.3e rms %.3e" % (k, np.mean([a for a,b in r]), np.mean([b for a,b in r])))
else:
    for k in ["F23","std","mixed"]:
        pts,m = sets[k]
        for kw in [dict(), dict(vexact=True), dict(acc64=True), dict(out64=True), dict(nosplit=True), dict(vexact=True,acc64=True,out64=True)]:
            print(k, kw, "max %.3e rms %.3e" % run(pts,m,0,**kw))
