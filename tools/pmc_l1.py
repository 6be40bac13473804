import re
"""development: per-dispatch counters of one frame's kernels from a tools/pmc.sh pass directory"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_" not in r["Kernel_Name"] or re.search(r"<\w+, true,", r["Kernel_Name"]):
        continue
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("::")[-1][:12])
    d.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
keys = sorted(d)
last_primary = [k for k in keys if "k_primary" in k[1]][-1]
for k in keys:
    if k[0] >= last_primary[0] and k[0] < last_primary[0] + 4:
        print(k, {a: "%.4g" % b for a, b in d[k].items()})
