#!/bin/bash
# round 3: the driver's default bench command (N = 1) + the other forward workloads + both training configurations
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_bench
mkdir -p $OUT
cd $R
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "rc=$?" >> $OUT/bench_default.err
for c in maze warcraft; do
  timeout 300 python bench.py --mode train --config $c --steps 40 --warmup 5 > $OUT/train_$c.json 2> $OUT/train_$c.err
done
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
for k in ("metric","value","ms_per_step","roofline","cpu_baseline","through_module","expansions_per_s"): print(k, json.dumps(d.get(k))[:1500])
for s_ in d.get("secondary",[]): print(json.dumps(s_)[:400])
for k,v in d.get("extra",{}).items(): print(k, json.dumps(v)[:600])
for c in ("maze","warcraft"):
    j=json.load(open("$OUT/train_%s.json" % c)); print("train", c, j["ms_per_step"], j["value"], j.get("torch_encoder_on_this_gpu"), j.get("cpu_baseline"))
PY
tail -3 $OUT/bench_default.err
