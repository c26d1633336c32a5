"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv files under the given dirs) per
kernel: sum of each counter over dispatches, and the dispatch count."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for d in sys.argv[1:]:
    for f in Path(d).rglob("*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k].add((str(f), row["Dispatch_Id"]))
out = {}
for k in tot:
    n_files = len({f for f, _ in disp[k]})
    out[k] = {"dispatches": len(disp[k]) // max(n_files, 1), **tot[k]}
print(json.dumps(out, indent=1))
