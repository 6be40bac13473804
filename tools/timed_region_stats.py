"""development: mean duration of k_primary over the TIMED steps of a rocprofv3 --kernel-trace run of the default
bench.py (the --stats summary averages every launch of the process: set-up frames without a launch plan, warm-up,
the timed steps, and the frames rendered one at a time afterwards).  python tools/timed_region_stats.py <dir> <steps> <warmup> [inflight]"""
import csv, glob, sys, statistics as st
d, steps, warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
inflight = int(sys.argv[4]) if len(sys.argv) > 4 else 3
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted((r for r in csv.DictReader(open(f)) if "rt_fast::k_primary<false, false, true, 1" in r["Kernel_Name"]),
              key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
a = 2 * inflight + warm
timed, after = dur[a:a + steps], dur[a + steps:]
print(f"{len(dur)} launches of the scene-shape k_primary (tolerance build) in the trace")
print(f"set-up + warm-up (first {a}): mean {st.mean(dur[:a]):.1f} us (the first frame of each context has no launch plan)")
print(f"timed steps ({len(timed)}): mean {st.mean(timed):.1f} us, median {st.median(timed):.1f}, min {min(timed):.1f}, max {max(timed):.1f}")
if after:
    print(f"afterwards, one frame at a time ({len(after)}): mean {st.mean(after):.1f} us")
print(f"all launches: mean {st.mean(dur):.1f} us (what --stats reports)")
