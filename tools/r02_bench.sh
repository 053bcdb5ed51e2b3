#!/bin/bash
# round 2: full default bench (what the driver runs) + the rand64 workload
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_bench
mkdir -p $OUT
cd $R
timeout 420 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "rc=$?" >> $OUT/bench_default.err
timeout 300 python bench.py --workload rand64 --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_rand64.json 2> $OUT/bench_rand64.err
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
for k in ("metric","value","ms_per_step","roofline","cpu_baseline","through_module","expansions_per_s"): print(k, json.dumps(d.get(k))[:700])
for s_ in d.get("secondary",[]): print(json.dumps(s_)[:400])
e=d.get("extra",{})
for k,v in e.items(): print(k, json.dumps(v)[:300])
d=json.load(open("$OUT/bench_rand64.json")); print("rand64", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
tail -3 $OUT/bench_default.err
