#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_encoder
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_UNALIGNED_STALL"; do
  T=$(echo $C | cut -c1-16 | tr " " _)
  rocprofv3 --pmc $C --kernel-trace -d $OUT -o p_$T --output-format csv -- python $R/tools/run_hip_encoder.py 1024 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "_kernel<128" in k:
            acc[k[:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in d.items(): print("   ", c, sum(v) / len(v))
PY
rm -f $OUT/*kernel_trace.csv
