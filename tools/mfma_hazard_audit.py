#!/usr/bin/env python
"""Static check of the kernels that split operands with inline-asm VALU instructions (winograd5, 6, 8.hip): gfx950 needs two wait
states between a VALU write of a VGPR and an MFMA that reads it as SrcA / SrcB.  The compiler keeps that distance for
instructions it knows, but it does not look inside inline asm — tests/test_gpu_conv.py's stem kernel lost 1.6 % of its outputs to
exactly this before its split was rewritten in plain C.  Usage: python tools/mfma_hazard_audit.py [file.hip ...]  (exit 1 on a
violation); tests/test_host.py runs it on every build."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "centernet-lightning_amd", "csrc")
ASM_KERNELS = ["winograd5.hip", "winograd6.hip", "winograd9.hip", "winograd10.hip", "winograd13.hip"]      # (tools/experiments/winograd3,4,7.hip: python tools/mfma_hazard_audit.py <path>)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-mllvm",
         "-pragma-unroll-threshold=4000000", "-fno-slp-vectorize", "-S", "--cuda-device-only"]      # = csrc/Makefile's for these files


def _regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit_asm(text):
    """Returns (number of MFMAs, list of violations) for one device assembly listing."""
    hist, bad, total = [], [], 0
    for ln in text.splitlines():
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        args = t[len(op):].split(",")
        if op.startswith("v_mfma"):
            total += 1
            src = _regs(args[1]) | _regs(args[2])
            ws = 0
            for w, d, txt in reversed(hist[-4:]):
                if ws >= 2:
                    break
                if d & src:
                    bad.append(f"{txt}  ->  {t}  ({ws} wait states)")
                    break
                ws += w
        w = int(args[0]) + 1 if op == "s_nop" else 1
        d = _regs(args[0]) if op.startswith("v_") and not op.startswith("v_cmp") else set()
        hist.append((w, d, t))
    return total, bad


def audit_files(names=ASM_KERNELS, hipcc="/opt/rocm/bin/hipcc"):
    out = {}
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for n in names:
            s = os.path.join(d, os.path.basename(n) + ".s")
            src = n if os.path.isabs(n) else os.path.join(CSRC, n)
            procs.append((n, s, subprocess.Popen([hipcc] + FLAGS + [src, "-o", s], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
        for n, s, p in procs:
            err = p.communicate()[1]
            if p.returncode != 0:
                raise RuntimeError(f"hipcc failed on {n}: {err.decode()[-500:]}")
            out[n] = audit_asm(open(s).read())
    return out


if __name__ == "__main__":
    res = audit_files(sys.argv[1:] or ASM_KERNELS)
    rc = 0
    for n, (total, bad) in res.items():
        print(f"{n}: {total} MFMAs, {len(bad)} violations")
        for b in bad:
            print("   ", b)
        rc |= bool(bad)
    sys.exit(rc)
