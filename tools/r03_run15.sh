#!/bin/bash
# round 3, GPU call: f16x3 CNN encoder: last layer fused into the 128->256 kernel (split products chained in the epilogue) + coalesced first layer
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "encoder or f16x3 or shipped or checkpoint" 2>&1 | tail -3
python - <<'P'
import sys, json
sys.path[:0]=["/root/repo","/root/repo/neural-astar_amd"]
import torch, bench, os
dev=torch.device("cuda:0")
pr=bench.make_problem("maze32",4096,1234)
r=bench.neural_astar_f16x3_ms(pr, dev); print("fused", json.dumps(r))
os.environ["NASTAR_ENCODER_FLAGS"]="8"
r=bench.neural_astar_f16x3_ms(pr, dev); print("unfused last layer (flags 8)", json.dumps(r))
P
cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r03/prof_f16x3 -o t --output-format csv -- python /root/repo/tools/run_hip_encoder_f16x3.py 4096 > /dev/null 2>&1
python - <<'P'
import csv, glob
for f in glob.glob("/root/repo/gpurun_out/r03/prof_f16x3/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:8]: print(r["Name"][:100], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
P
