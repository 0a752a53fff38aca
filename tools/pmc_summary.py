"""Aggregate rocprofv3 counter_collection CSVs per kernel family: mean counter value per dispatch."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def fam(name):
    m = re.search(r"k_lvc_layer<(\d+)", name)
    if m:
        return "lvc_layer_h" + m.group(1)
    m = re.search(r"::(k_\w+)", name)
    return m.group(1) if m else name[:40]


acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(os.path.join(out, "*counter_collection.csv"))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = fam(row["Kernel_Name"])
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
dur = defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(os.path.join(out, "p1_kernel_trace.csv"))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = fam(row["Kernel_Name"])
            dur[k][0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); dur[k][1] += 1
for k in sorted(acc, key=lambda k: -dur[k][0]):
    d = dur[k][0] / max(dur[k][1], 1)
    print(f"{k:22s} n={dur[k][1]:4d} avg_us={d/1e3:9.1f} " + " ".join(f"{c}={v[0]/v[1]:.4g}" for c, v in sorted(acc[k].items())))
