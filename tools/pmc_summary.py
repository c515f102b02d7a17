"""Summarise a rocprofv3 --pmc run (csv output): average counter value per kernel launch."""
import csv, sys, collections, glob
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
filt = sys.argv[2:] or None
for k, v in d.items():
    if filt and not any(x in k for x in filt):
        continue
    print(k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())}, "launches", len(next(iter(v.values()))))
