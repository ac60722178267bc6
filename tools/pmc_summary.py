#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per (kernel, counter)."""
import csv
import glob
import sys
from collections import defaultdict


def main(root):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1
    for k in acc:
        print("==", k)
        for c in sorted(acc[k]):
            print("  %-26s total %.4g   per-dispatch %.4g  (%d dispatches)" % (c, acc[k][c], acc[k][c] / calls[k][c], calls[k][c]))


if __name__ == "__main__":
    main(sys.argv[1])
