#!/bin/bash
# per-kernel timing of the HIP encoder (run on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_encoder
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o enc --output-format csv -- python $R/tools/run_hip_encoder.py > $OUT/log.txt 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/enc_kernel_stats.csv")))
for r in rows[:9]:
    print(r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
PY
grep "HIP encoder" $OUT/log.txt; rm -f $OUT/enc_kernel_trace.csv
