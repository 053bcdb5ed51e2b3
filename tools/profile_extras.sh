#!/bin/bash
# Run on the GPU box: evidence for the numbers DESIGN.md quotes outside the headline kernel.
#   1. matrix-pipe / LDS counters of the encoder kernels (separate --pmc passes, kernel-trace only)
#   2. kernel stats of the training step (forward in training mode + backward) and of the device data loader
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_extras
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for C in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  T=$(echo $C | cut -d" " -f1)
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$T -o enc --output-format csv -- python $R/tools/run_hip_encoder.py 4096 > /dev/null 2>&1
done
cat > /tmp/train_step.py <<PY
import sys
sys.path[:0] = ["$R/neural-astar_amd", "$R"]
import torch
from neural_astar import ops
from neural_astar.utils import synthetic as syn
dev = torch.device("cuda:0")
pr = syn.maze_maps(4096, 32, seed=1234)
m, s, g = (torch.from_numpy(x[:, 0]).to(dev).contiguous() for x in pr)
traj = ((torch.rand_like(m) < 0.2).float() * m).contiguous()
cost = torch.from_numpy(syn.random_costs(4096, 32, 32, seed=3)[:, 0]).to(dev).requires_grad_(True)
for _ in range(20):
    cost.grad = None
    ops.astar_l1_loss(cost, s, g, m, traj, 0.5, 256)[0].backward()
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d $OUT/train -o train --output-format csv -- python /tmp/train_step.py > /dev/null 2>&1
python - <<PY
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("$OUT/pmc_*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "nastar" in k:
            acc[k.split("(")[0][-70:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in d.items()})
json.dump(out, open("$OUT/encoder_pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
rows = list(csv.DictReader(open(glob.glob("$OUT/train/*kernel_stats.csv")[0])))
for r in rows[:6]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -f $OUT/*/*kernel_trace.csv $OUT/*/*agent_info.csv
