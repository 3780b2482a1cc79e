#!/usr/bin/env python
"""Summarise rocprofv3 --pmc output (csv or rocpd sqlite) per kernel: mean counter value per dispatch."""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def from_csv(root):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    root = sys.argv[1]
    acc = from_csv(root)
    for kern, ctrs in sorted(acc.items()):
        n = max(len(v) for v in ctrs.values())
        print(f"{kern}  (dispatches {n})")
        for c, v in sorted(ctrs.items()):
            print(f"    {c:32s} mean {sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main()
