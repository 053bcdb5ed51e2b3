#!/bin/bash
# round 2, GPU run 2: parity of the two-maps-per-wavefront forward kernel, A/B/C bench (duo / single compact / round-1 layout)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_run2
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --deselect "tests/test_gpu_parity.py::test_backward_matches_reference_autograd[grad_rand100_eval_g050]" --deselect "tests/test_gpu_parity.py::test_backward_matches_reference_autograd[grad_rand96_train_T005]" 2>&1 | tail -15 > $OUT/pytest.log
for V in 0 4 1; do
  NASTAR_FORWARD_FLAGS=$V timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $OUT/bench_f$V.json 2> $OUT/bench_f$V.err
done
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  NASTAR_FORWARD_FLAGS=$V timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc_sq_f$V -o bench --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_sq_f$V.log 2>&1
done
python - <<PY
import csv, glob, json, collections
out={}
for V in ("0","1"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/pmc_sq_f%s/**/*counter_collection.csv"%V, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out["flags"+V]={k:{c:sum(v)/len(v) for c,v in d.items()} for k,d in acc.items() if "forward" in k}
json.dump(out, open("$OUT/sq_summary.json","w"), indent=1)
print(json.dumps(out, indent=1))
PY
for V in 0 4 1; do python -c "
import json
d=json.load(open('$OUT/bench_f$V.json')); print('flags',$V, d['value'], d['ms_per_step'], d['roofline']['launch_ms_avg'], d['roofline']['frac'])
"; done
cat $OUT/pytest.log
