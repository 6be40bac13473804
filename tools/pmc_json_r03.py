"""Counter CSVs of tools/final_profile_r03.sh -> the JSON kept as profiles/r03_pmc.json (read by bench.py: the fields
are reported only while the kernels' sources hash to `kernels_sha`).  Per build ("c2:fast", "c2:exact") and kernel:
per-launch averages over the launches of each pass."""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernels_sha  # noqa: E402
tag = sys.argv[1]
out = {"_comment": "rocprofv3 --pmc, one run per pass (tools/final_profile_r03.sh), bench.py --steps 10 --tail-stream off at "
                   "config 2 (builtin scene 1920x1080, 4x4 samples, depth 8), per-launch averages. SQ_* summed over the chip; "
                   "SQ_BUSY_CYCLES / 32 = elapsed shader cycles; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles. "
                   "FETCH_SIZE / WRITE_SIZE in KiB; hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950 tallies "
                   "128-byte read requests at 64 bytes (MI355X_MICROARCH.md, HBM), writes are counted exactly. "
                   "fp_lane_flops_per_launch = (2 x FMA + MUL + ADD + TRANS instructions, F32 and F64) x 64 lanes x the share of "
                   "lanes active per vector instruction.",
       "kernels_sha": kernels_sha()}
for num in ("fast", "exact"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_{tag}_{num}_*/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        grid = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            m = re.search(r"(rt_\w+::k_[a-z_]+)(<[^>]*>)?", k)
            if not m:
                continue
            name = m.group(1) + (m.group(2) or "")
            if re.search(r"k_(primary|bounce)<\w+, true", name):  # the instrumented pass outside the timed region
                continue
            per[(name, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            grid[(name, r["Dispatch_Id"])] = int(r["Grid_Size"])
        # k_primary: whole frames only (a context's first frames size its queues and are rendered in batches)
        full = max((g for (n, _), g in grid.items() if "k_primary" in n), default=0)
        # ... and of those the last third of the run: sized, planned frames (what tools/hbm_by_kernel.sh takes as well)
        ids = sorted({int(dsp) for (n, dsp), g in grid.items() if "k_primary" in n and g == full})
        keep = set(ids[len(ids) * 2 // 3:])
        for (name, disp, ctr), v in per.items():
            if "k_primary" in name and int(disp) not in keep:
                continue
            agg[name][ctr].append(v)
    build = {}
    for name, ctrs in agg.items():
        short = re.sub(r"rt_\w+::", "", name).split("<")[0]
        d = {c: sum(v) / len(v) for c, v in ctrs.items()}
        if short == "k_bounce":  # level 1 is the largest launch
            d = {c: max(v) for c, v in ctrs.items()}
            short = "k_bounce_level1"
        build.setdefault(short, {}).update({k: round(v, 1) for k, v in d.items()})
        build[short]["kernel"] = name
    kp = build.get("k_primary", {})
    if kp:
        cyc = kp.get("SQ_BUSY_CYCLES", 0) / 32
        lanes = kp.get("SQ_THREAD_CYCLES_VALU", 0) / (kp["SQ_INSTS_VALU"] * 64) if kp.get("SQ_INSTS_VALU") else 0.0
        fp = sum(w * kp.get(f"SQ_INSTS_VALU_{op}_{ty}", 0) for op, w in (("FMA", 2), ("MUL", 1), ("ADD", 1), ("TRANS", 1)) for ty in ("F32", "F64"))
        build["hbm_bytes_per_launch"] = int((2 * kp.get("FETCH_SIZE", 0) + kp.get("WRITE_SIZE", 0)) * 1024)
        build["valu_lane_ops_per_launch"] = kp.get("SQ_INSTS_VALU", 0) * 64
        build["valu_active_lane_ops_per_launch"] = kp.get("SQ_THREAD_CYCLES_VALU", 0)
        build["active_lanes_per_valu_inst"] = round(lanes, 4)
        build["fp_lane_flops_per_launch"] = fp * 64 * lanes
        build["valu_busy"] = round(kp.get("SQ_INSTS_VALU", 0) * 2 / (1024 * cyc), 4) if cyc else None
        build["valu_busy_definition"] = "vector instructions x 2 issue cycles / (1024 SIMDs x elapsed cycles)"
        build["lds_conflict_rate"] = (kp.get("SQ_LDS_BANK_CONFLICT", 0) / kp["SQ_LDS_IDX_ACTIVE"]) if kp.get("SQ_LDS_IDX_ACTIVE") else None
        build["waves_per_simd"] = round(kp.get("SQ_WAVE_CYCLES", 0) * 4 / (1024 * cyc), 2) if cyc else None
    out[f"c2:{num}"] = build
print(json.dumps(out, indent=1))
