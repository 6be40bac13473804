"""development: mean duration of k_primary over the TIMED steps of a rocprofv3 --kernel-trace run of the default
bench.py (the --stats summary averages every launch of the process: set-up frames without a launch plan, warm-up,
the timed steps, and the frames rendered one at a time afterwards).  python tools/timed_region_stats.py <dir> <steps> <warmup> [frames one at a time afterwards]"""
import csv, glob, sys, statistics as st
d, steps, warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
alone = int(sys.argv[4]) if len(sys.argv) > 4 else 40  # frames rendered one at a time after the timed region (bench.py: 10 + 30)
import os
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)  # (the bench itself, not its one-shot child processes)
rows = sorted((r for r in csv.DictReader(open(f)) if "rt_fast::k_primary<false, false, true, 1" in r["Kernel_Name"]),
              key=lambda r: int(r["Start_Timestamp"]))
full = max(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) for r in rows)
whole = [r for r in rows if int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) == full]  # (a context's first frames are rendered in batches)
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in whole]
timed, after, before = dur[-(steps + alone):-alone], dur[-alone:], dur[:-(steps + alone)]
print(f"{len(rows)} launches of the scene-shape k_primary (tolerance build) in the trace, {len(whole)} of them whole frames")
print(f"set-up + warm-up ({len(before)} whole frames): mean {st.mean(before):.1f} us (the first whole frame of each context has no launch plan)")
print(f"timed steps ({len(timed)}): mean {st.mean(timed):.1f} us, median {st.median(timed):.1f}, min {min(timed):.1f}, max {max(timed):.1f}")
print(f"afterwards, one frame at a time ({len(after)}): mean {st.mean(after):.1f} us")
alld = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(f"all launches: mean {st.mean(alld):.1f} us (what --stats reports)")
