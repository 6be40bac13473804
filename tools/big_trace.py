#!/usr/bin/env python3
"""development: ms per kind of kernel in the LAST frame of a rocprofv3 --kernel-trace run (tools/big_round.sh)."""
import collections, csv, glob, re, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
rows = sorted((r for r in rows if "k_" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_hits" in r["Kernel_Name"] or ("k_primary<" in r["Kernel_Name"] and not re.search(r"k_primary<\w+, true,", r["Kernel_Name"]))]
fr = rows[idx[-1]:]
agg = collections.OrderedDict()
for r in fr:
    m = re.search(r"(k_[a-z_]+)(<[^>]*>)?", r["Kernel_Name"])
    k = m.group(1) + (m.group(2) or "")
    a = agg.setdefault(k, [0, 0.0, []])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a[2].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
tot = 0.0
for k, (n, ms, each) in agg.items():
    tot += ms
    print(f"{k:28s} x{n:<3d} {ms:8.3f} ms   " + " ".join(f"{e:.2f}" for e in each[:17]))
print(f"sum {tot:.2f} ms; frame span {(int(fr[-1]['End_Timestamp']) - int(fr[0]['Start_Timestamp'])) / 1e6:.2f} ms")
