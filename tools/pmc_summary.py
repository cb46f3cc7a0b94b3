#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel name (optionally per grid)."""
import csv, sys, glob, collections
rows = []
for f in sys.argv[1:]:
    for path in glob.glob(f, recursive=True):
        rows += list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r.get("Kernel_Name", "")
    if "kemr" not in name:
        continue
    key = (name.split("(")[0].replace("void kemr::", "")[:48], r.get("Grid_Size", ""))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(agg.items()):
    print(key, {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}, "n=", len(next(iter(cs.values()))))
