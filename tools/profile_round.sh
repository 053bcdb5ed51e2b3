#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + HBM traffic counters for the bench workload.
# Usage: tools/profile_round.sh r01   -> writes gpurun_out/profiles_<tag>/...
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the exact bench command (N=1)
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.log 2>&1
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limit), kernel-trace only
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o bench --output-format csv -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_$C.log 2>&1
done
python - <<PY
import csv, glob, json, collections
out = {}
for f in glob.glob("$OUT/trace/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    out["kernel_stats"] = rows[:8]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/pmc_%s/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    out[c] = {k: {"n": len(v), "mean": sum(v) / len(v)} for k, v in acc.items()}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
