#!/bin/bash
# usage: tools/pmc_any.sh <tag> "<python script and args>" <counters...>   (on the GPU box, from the repo root)
tag=$1; shift; cmd=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 $cmd > gpurun_out/pmc_$tag.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections,re
f=glob.glob("gpurun_out/pmc_$tag/**/*counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); seen=set()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "k_" not in k: continue
    m=re.search(r"(k_[a-z]+)(<[^>]*>)?", k); k=(m.group(1)+(m.group(2) or "")) if m else k[:40]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); seen.add((k,r["Dispatch_Id"]))
calls=collections.Counter(k for k,_ in seen)
for k,v in agg.items(): print(k, "calls",calls[k], "TOTALS", {a:"%.4g"%b for a,b in v.items()})
PY
