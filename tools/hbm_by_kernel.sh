#!/bin/bash
# development: HBM bytes per kernel launch (FETCH_SIZE / WRITE_SIZE, one rocprofv3 run each) of a short one-stream bench run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/hbm_$pass -- python3 bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-one-shot --no-generic --tail-stream off > gpurun_out/hbm_$pass.log 2>&1 || { tail -5 gpurun_out/hbm_$pass.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, re, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/hbm_{ctr}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(float); name = {}; grid = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr: continue
        per[r["Dispatch_Id"]] += float(r["Counter_Value"]); name[r["Dispatch_Id"]] = r["Kernel_Name"]; grid[r["Dispatch_Id"]] = r.get("Grid_Size", "")
    ids = sorted(per, key=int)
    ids = ids[len(ids) * 2 // 3:]  # the last third of the run: planned, sized frames
    for i in ids:
        m = re.search(r"(k_[a-z_]+)(<[^>]*>)?", name[i]); k = m.group(1) if m else name[i][:20]
        if k in ("k_bounce", "k_fold", "k_primary"): k += " grid " + grid[i]  # (k_primary: whole frames and the batches of a context's sizing frames)
        tot[k][ctr].append(per[i])
for k, v in sorted(tot.items()):
    f = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1); w = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
    print(f"{k:32s} launches {len(v['FETCH_SIZE']):3d}  read {2 * f * 1024 / 1e6:7.1f} MB (2 x FETCH_SIZE)  written {w * 1024 / 1e6:7.1f} MB")
PY
