#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats of the LARGE-MAP kernel (csrc/nastar_search_hybrid.hip.h) on the probe's inputs --
# the per-launch durations behind DESIGN.md section 4.4's "ns per step of the longest search" (tools/probe_large.py times with HIP events).
# Usage: tools/profile_large.sh r06 -> gpurun_out/profiles_<tag>/large_map_kernel_stats.csv (copy into profiles/<tag>/): MinNs = the 256x256
# launches, MaxNs = the 512x512 ones
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/large_probe.py <<PY
import sys
sys.path[:0] = ["$R/neural-astar_amd", "$R"]
import torch
from neural_astar import ops
from neural_astar.utils import synthetic as syn
dev = torch.device("cuda:0")
for H, B in ((256, 256), (512, 256)):
    pr = syn.random_obstacle_maps(B, H, H, 0.2, seed=7)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    for _ in range(6):
        out = ops.search_nograd(m, s, g, m, 0.5, H * H)
    torch.cuda.synchronize()
    print(H, B, "longest search", int(out[2].max()), "steps; all maps", int(out[2].sum()), "steps")
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_large -o large --output-format csv -- python /tmp/large_probe.py > $OUT/large_under_rocprof.log 2>&1
for f in $(find $OUT/trace_large -name "*kernel_stats.csv"); do cp $f $OUT/large_map_kernel_stats.csv; done
rm -rf $OUT/trace_large
tail -3 $OUT/large_under_rocprof.log
# instruction counts of the search launch (PMC, own pass; kernel trace only): per launch, to be divided by the launch's total steps
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace -d $OUT/pmc_large -o large --output-format csv -- python /tmp/large_probe.py > $OUT/large_pmc.log 2>&1
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_large/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "forward_hybrid" in r["Kernel_Name"]:
            acc[(int(r["Grid_Size"]), r["Counter_Name"])]["v"].append(float(r["Counter_Value"]))
out = {}
for (grid, name), d in sorted(acc.items()):
    out.setdefault(str(grid), {})[name] = sum(d["v"]) / len(d["v"])
json.dump(out, open("$OUT/large_map_kernel_counters.json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
rm -rf $OUT/pmc_large
grep "longest search" $OUT/large_pmc.log
