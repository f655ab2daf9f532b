"""Summarise a rocprofv3 counter_collection.csv: mean counter value per dispatch, per kernel."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r.get("Kernel_Name", "?").split("(")[0]
        if len(k) > 60:
            k = k[:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "trc_" not in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s mean %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
