#!/usr/bin/env python3
"""development: per-kernel table of the LAST frame of tools/profile_big_r03.sh's runs (kernel trace + PMC passes).
usage: pmc_big.py <tag> <cfg>     reads gpurun_out/prof_<tag>_<cfg> and gpurun_out/pmc_<tag>_<cfg>_<n>"""
import collections, csv, glob, re, sys

tag, cfg = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r"(rt_\w+::)?(k_[a-z_]+)(<[^>]*>)?", name)
    return (m.group(2) + (m.group(3) or "")) if m else name[:40]


def last_frame(rows, key):
    """rows sorted by dispatch order; the last frame starts at the last plain (not instrumented) k_primary"""
    rows = [r for r in rows if "k_" in r["Kernel_Name"]]
    idx = [i for i, r in enumerate(rows) if "k_hits" in r["Kernel_Name"] or
           ("k_primary<" in r["Kernel_Name"] and not re.search(r"k_primary<\w+, true,", r["Kernel_Name"]))]
    return rows[idx[-1]:] if idx else []


trace = list(csv.DictReader(open(glob.glob(f"gpurun_out/prof_{tag}_{cfg}/**/*kernel_trace.csv", recursive=True)[0])))
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
frame = last_frame(trace, "Start_Timestamp")
labels, dur = [], []
seen = collections.Counter()
for r in frame:
    s = short(r["Kernel_Name"])
    seen[s] += 1
    labels.append(f"{s}#{seen[s]}")
    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
vg = [r.get("VGPR_Count", "?") for r in frame]
grid = [int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // 256 for r in frame]

ctr = [collections.defaultdict(float) for _ in frame]
for d in sorted(glob.glob(f"gpurun_out/pmc_{tag}_{cfg}_*/")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    per = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        k = int(r["Dispatch_Id"])
        per.setdefault(k, {"Kernel_Name": r["Kernel_Name"], "c": collections.defaultdict(float)})
        per[k]["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    rows = [dict(Kernel_Name=v["Kernel_Name"], c=v["c"]) for _, v in sorted(per.items())]
    fr = last_frame(rows, None)
    if len(fr) != len(frame):
        print(f"# {d}: {len(fr)} kernels in the last frame against {len(frame)} in the trace: matched by position as far as they go")
    for i, r in enumerate(fr[:len(frame)]):
        ctr[i].update(r["c"])

print(f"# {cfg}, exact build, last frame, one frame at a time (tools/profile_big_r03.sh {tag}); us = kernel-trace duration;")
print("# busy = 2 x VALU insts / (1024 SIMDs x elapsed cycles); lanes = active lanes per VALU inst / 64; w/simd = resident waves per SIMD;")
print("# wait = share of wave cycles parked in s_waitcnt; stall = share in issue stalls; L2hit = TCC hit / (hit + miss); L1miss = TCP->TCC reads / TCP accesses;")
print("# MBrd / MBwr = HBM bytes (2 x FETCH_SIZE, WRITE_SIZE); issue = (2 x VALU + SALU + SMEM + LDS + branch + VMEM insts) / (1024 SIMDs x elapsed cycles)")
print(f"{'kernel':34s} {'us':>9s} {'blocks':>7s} {'vgpr':>4s} {'waves':>8s} {'valu_M':>8s} {'busy':>5s} {'lanes':>5s} {'w/simd':>6s} {'wait':>5s} {'stall':>5s} {'vmemrd_M':>8s} {'L2hit':>5s} {'L1miss':>6s} {'MBrd':>7s} {'MBwr':>7s} {'salu_M':>8s} {'brnch_M':>8s} {'lds_M':>7s} {'issue':>5s}")
tot = 0.0
for i, lab in enumerate(labels):
    c = ctr[i]
    cyc = c.get("SQ_BUSY_CYCLES", 0) / 32
    f = lambda x: f"{x:5.2f}" if x is not None else "    -"
    busy = c["SQ_INSTS_VALU"] * 2 / (1024 * cyc) if cyc else None
    lanes = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_INSTS_VALU"] * 64) if c.get("SQ_INSTS_VALU") else None
    wps = c["SQ_WAVE_CYCLES"] * 4 / (1024 * cyc) if cyc else None
    wait = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None
    stall = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None
    l2 = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) else None
    l1 = c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"] if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") else None
    tot += dur[i]
    # the SIMDs' issue slots in use by the model of DESIGN (a vector instruction two cycles, every other kind one)
    issue = (2 * c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"] + c["SQ_INSTS_SMEM"] + c["SQ_INSTS_LDS"] + c["SQ_INSTS_BRANCH"]
             + c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]) / (1024 * cyc) if cyc else None
    print(f"{lab:34s} {dur[i]:9.1f} {grid[i]:7d} {vg[i]:>4s} {c.get('SQ_WAVES', 0):8.0f} {c.get('SQ_INSTS_VALU', 0) / 1e6:8.2f} {f(busy)} {f(lanes)} {f(wps):>6s} {f(wait)} {f(stall)} "
          f"{c.get('SQ_INSTS_VMEM_RD', 0) / 1e6:8.2f} {f(l2)} {f(l1):>6s} {c.get('FETCH_SIZE', 0) * 2 / 1024:7.1f} {c.get('WRITE_SIZE', 0) / 1024:7.1f} {c.get('SQ_INSTS_SALU', 0) / 1e6:8.2f} {c.get('SQ_INSTS_BRANCH', 0) / 1e6:8.2f} {c.get('SQ_INSTS_LDS', 0) / 1e6:7.2f} {f(issue)}")
print(f"# sum of kernel durations {tot / 1e3:.2f} ms")
