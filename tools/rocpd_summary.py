#!/usr/bin/env python
"""Summaries of rocprofv3's rocpd sqlite output (ROCm 7: the default output format), as committed under profiles/.

    python tools/rocpd_summary.py stats <dir-with-*_results.db>              -> per-kernel calls / total / average (us) / %
    python tools/rocpd_summary.py pmc <fetch-dir> <write-dir>                 -> per-kernel mean FETCH_SIZE / WRITE_SIZE per launch
    python tools/rocpd_summary.py json <stats-dir> <fetch-dir> <write-dir> <out.json> [key=value ...]
                                                                              -> the same numbers as ONE json (what bench.py quotes as
                                                                                 roofline.traffic / decode.kernels_profiled: a quoted figure
                                                                                 is then byte-equal to a field of a committed profiles/ file)
"""
import json
import glob
import os
import sqlite3
import sys


def _db(root):
    paths = glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)
    if not paths:
        raise SystemExit(f"no *_results.db under {root}")
    return sqlite3.connect(paths[0])


def _short(name):
    return name.split("(")[0].replace("void ", "")[:70]


def stats(root):
    c = _db(root)
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage"')
    for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print(f'"{_short(name)}",{calls},{total * 1e3:.0f},{avg * 1e3:.0f},{pct:.2f}')


def as_json(sroot, froot, wroot, out, meta):
    """{"meta": {...}, "kernels": {short name: {calls, avg_us, total_us, pct, fetch_KiB, write_KiB, hbm_MB_per_launch}}};
    hbm_MB_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / 1e6 — gfx950's FETCH_SIZE reports half of a 16 B/lane stream
    (MI355X_MICROARCH.md, HBM section)."""
    kern = {}
    for name, calls, total, avg, pct in _db(sroot).execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        kern[_short(name)] = {"calls": calls, "total_us": round(total, 1), "avg_us": round(avg, 2), "pct": round(pct, 2)}
    for root in (froot, wroot):
        for name, ctr, n, mean in _db(root).execute(
                "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
            k = kern.setdefault(_short(name), {})
            k[{"FETCH_SIZE": "fetch_KiB", "WRITE_SIZE": "write_KiB"}.get(ctr, ctr)] = round(mean, 1)
            k["pmc_launches"] = n
    for k in kern.values():
        if "fetch_KiB" in k and "write_KiB" in k:
            k["hbm_MB_per_launch"] = round((2 * k["fetch_KiB"] + k["write_KiB"]) * 1024 / 1e6, 1)
    json.dump({"meta": meta, "kernels": kern}, open(out, "w"), indent=1, sort_keys=True)
    print(f"wrote {out}: {len(kern)} kernels")


def pmc(froot, wroot):
    acc = {}
    for root in (froot, wroot):
        for name, ctr, n, mean in _db(root).execute(
                "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
            acc.setdefault(_short(name), {})[ctr] = (n, mean)
    print(f"{'kernel':70s} {'launches':>8s} {'FETCH_SIZE KiB':>15s} {'WRITE_SIZE KiB':>15s} {'HBM MB/launch (FETCH x2 + WRITE)':>34s}")
    for k, v in sorted(acc.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", (0, 0))[1] * kv[1].get("FETCH_SIZE", (0, 0))[0])):
        f, w = v.get("FETCH_SIZE", (0, 0.0)), v.get("WRITE_SIZE", (0, 0.0))
        print(f"{k:70s} {max(f[0], w[0]):8d} {f[1]:15.1f} {w[1]:15.1f} {(2 * f[1] + w[1]) * 1024 / 1e6:34.1f}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "json":
    as_json(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], dict(kv.split("=", 1) for kv in sys.argv[6:]))
    sys.exit(0)
if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
