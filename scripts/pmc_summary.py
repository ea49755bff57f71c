#!/usr/bin/env python3
"""Mean per-launch value of every counter in rocprofv3 counter_collection.csv files found under
the given directories, grouped by kernel name."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(list)
    for d in sys.argv[1:]:
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    key = (row["Dispatch_Id"], row["Counter_Name"])
                    per_dispatch[key] += float(row["Counter_Value"])
                    names[row["Dispatch_Id"]] = row["Kernel_Name"]
            for (disp, cn), v in per_dispatch.items():
                acc[(names[disp][:70], cn)].append(v)
    for (k, cn), vs in sorted(acc.items()):
        if "at::native" in k or "elementwise" in k:
            continue
        print("pmc %-70s %-24s n=%3d mean=%.6g" % (k, cn, len(vs), sum(vs) / len(vs)))


if __name__ == "__main__":
    main()
