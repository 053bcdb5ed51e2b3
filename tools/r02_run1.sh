#!/bin/bash
# round 2, GPU run 1: parity of the compact forward kernel + new grad goldens, A/B bench (compact vs round-1 layout), SQ counters
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_run1
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_backward_matches_reference_autograd 2>&1 | tail -15 > $OUT/pytest_fwd.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "test_backward_matches_reference_autograd" 2>&1 | tail -40 > $OUT/pytest_bwd.log
python -c "
import sys; sys.path.insert(0,'neural-astar_amd')
from neural_astar import _native
import ctypes
lib=_native.load()
for (h,w) in [(32,32),(64,64),(16,16),(12,12),(100,100),(128,128)]:
    b=ctypes.c_int(0); n=lib.nastar_debug_occupancy(h,w,ctypes.byref(b)); print(h,w,'wg/cu',n,'lds',b.value)
" > $OUT/occupancy.log 2>&1
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $OUT/bench_compact.json 2> $OUT/bench_compact.err
NASTAR_FORWARD_FLAGS=1 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $OUT/bench_legacy.json 2> $OUT/bench_legacy.err
cd /tmp && export TMPDIR=/tmp
for V in compact legacy; do
  F=0; [ $V = legacy ] && F=1
  NASTAR_FORWARD_FLAGS=$F timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_$V -o bench --output-format csv -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/trace_$V.log 2>&1
  NASTAR_FORWARD_FLAGS=$F timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc_sq_$V -o bench --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_sq_$V.log 2>&1
done
python - <<PY
import csv, glob, json, collections
out={}
for V in ("compact","legacy"):
    o={}
    for f in glob.glob("$OUT/trace_%s/**/*kernel_stats.csv"%V, recursive=True):
        o["kernel_stats"]=list(csv.DictReader(open(f)))[:4]
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/pmc_sq_%s/**/*counter_collection.csv"%V, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    o["sq"]={k:{c:sum(v)/len(v) for c,v in d.items()} for k,d in acc.items() if "forward" in k}
    out[V]=o
json.dump(out, open("$OUT/summary.json","w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
