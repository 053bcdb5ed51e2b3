#!/bin/bash
# kernel-time breakdown of the encoder training step (rocprofv3 --kernel-trace --stats); usage: tools/prof_train.sh B precision tag
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-4096}; P=${2:-f16}; TAG=${3:-train}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python $R/tools/train_step_profile.py $B $P 3 > $OUT/log.txt 2>&1
F=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$F" $OUT/kernel_stats.csv 2>/dev/null
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step:", tot/3e6)
for r in rows[:22]:
    print(f'{float(r["TotalDurationNs"])/3e6:9.3f} ms/step  {int(r["Calls"])//3:4d} calls/step  {r["Name"][:110]}')
print("launches per step:", sum(int(r["Calls"]) for r in rows)/3)
PY
