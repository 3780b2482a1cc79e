#!/usr/bin/env python
"""Summaries of rocprofv3's rocpd sqlite output (ROCm 7: the default output format), as committed under profiles/.

    python tools/rocpd_summary.py stats <dir-with-*_results.db>              -> per-kernel calls / total / average (us) / %
    python tools/rocpd_summary.py pmc <fetch-dir> <write-dir>                 -> per-kernel mean FETCH_SIZE / WRITE_SIZE per launch
"""
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


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
