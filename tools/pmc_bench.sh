#!/bin/bash
# usage: tools/pmc_bench.sh <tag> "<bench args>" <counters...>   (run on the GPU box from the repo root)
tag=$1; shift; bargs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-shot $bargs > gpurun_out/pmc_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_$tag.log; exit 1; }
python3 - <<PY
import csv,glob,collections,re
f=glob.glob("gpurun_out/pmc_$tag/**/*counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); seen=set()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "k_" not in k: continue
    m=re.search(r"((?:rt_\w+::)?k_[a-z_]+)(<[^>]*>)?", k); k=(m.group(1)+(m.group(2) or "")) if m else k[:40]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); seen.add((k,r["Dispatch_Id"]))
calls=collections.Counter(k for k,_ in seen)
for k,v in sorted(agg.items()): print(k, "calls",calls[k], {a:"%.4g"%(b/calls[k]) for a,b in v.items()})
PY
