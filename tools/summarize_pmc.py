#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel name from a counter_collection csv.
usage: python tools/summarize_pmc.py <dir-or-csv> [...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    files = []
    for a in sys.argv[1:]:
        files += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(a) else [a]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            short = name.replace("void ", "").replace("smo::(anonymous namespace)::", "")
            short = short[:short.index(">(") + 1] if ">(" in short else short.split("(")[0]
            c = acc[short][row["Counter_Name"]]
            c[0] += float(row["Counter_Value"]); c[1] += 1
    for k in sorted(acc):
        print(k)
        for cn, (tot, n) in sorted(acc[k].items()):
            print("    %-28s avg %.6g over %d dispatches" % (cn, tot / n, n))


if __name__ == "__main__":
    main()
