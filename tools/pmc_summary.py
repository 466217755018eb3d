#!/usr/bin/env python3
"""Summarise rocprofv3 CSV outputs (kernel_stats / counter_collection) per kernel.

    python tools/pmc_summary.py gpurun_out/prof_kt gpurun_out/prof_pmc1 ... > profiles/rNN_summary.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name[-70:]


def main(dirs):
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, "*kernel_stats.csv"))):
            print("# %s" % f)
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    print("%-72s calls=%-5s total_ns=%-12s avg_ns=%-12s pct=%s" % (short(row["Name"]), row["Calls"], row["TotalDurationNs"], row["AverageNs"], row["Percentage"]))
        for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
            print("# %s" % f)
            agg = defaultdict(lambda: [0, 0.0])
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = (short(row["Kernel_Name"]), row["Counter_Name"])
                    agg[k][0] += 1
                    agg[k][1] += float(row["Counter_Value"])
            for (kn, cn), (n, v) in sorted(agg.items()):
                print("%-72s %-24s dispatches=%-4d avg_per_dispatch=%.1f" % (kn, cn, n, v / n))


if __name__ == "__main__":
    main(sys.argv[1:])
