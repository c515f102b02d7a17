"""Per-dispatch counter values, in dispatch order, of one kernel from a rocprofv3 --pmc csv directory.
Usage: pmc_sequence.py <dir> <kernel substring> [every]"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
want = sys.argv[2]
every = int(sys.argv[3]) if len(sys.argv) > 3 else 1
d = collections.OrderedDict()
for r in rows:
    if want not in r["Kernel_Name"]:
        continue
    d.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({k for v in d.values() for k in v})
print("dispatch", *names)
for n, (k, v) in enumerate(sorted(d.items())):
    if n % every == 0:
        print(n, *[int(v.get(c, 0)) for c in names])
