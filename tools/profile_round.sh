#!/bin/bash
# Run on the GPU box (via gpurun): the evidence the bench line's roofline block is checked against.
#   1. rocprofv3 kernel trace + stats of the default bench command (N=1; hinted placement, and once more with --placement natural) -> kernel_stats
#   2. HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slot limit), kernel-trace only
#   3. SQ counters of the shipped forward kernel and, for comparison, of other streams / layouts (SQ_FLAGS, default "0 128 64":
#      general layout, unit-cost layout; the older instruction streams need NASTAR_LIB=.../libnastar_hip_dev.so since round 6)
#   4. kernel stats of the fused training step (forward with selection log + replay backward), 4096 maps, Tmax = 0.25
#   5. (round 6) per-kernel table of the ENCODERS: CNN f16x3 / bf16 inference and one f16x3 training step on 4096 maps of 32x32
# Usage: tools/profile_round.sh r03   -> writes gpurun_out/profiles_<tag>/ (copy the summaries into profiles/<tag>/)
set -u
TAG=${1:-r04}
SQ_FLAGS=${SQ_FLAGS:-0 64}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (--no-natural: the default bench line also times the natural-order placement beside the hinted one; a kernel average must not mix the two)
# (--no-prewarm: the pre-warm launches are bare dataset-placed launches and would dominate the kernel's call count)
B="python $R/bench.py --no-cpu-baseline --no-secondary --no-natural --no-prewarm"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- $B --steps 200 --warmup 20 > $OUT/bench_under_rocprof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_natural -o bench --output-format csv -- $B --placement natural --steps 200 --warmup 20 > $OUT/bench_natural_under_rocprof.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o bench --output-format csv -- $B --steps 20 --warmup 2 > $OUT/pmc_$C.log 2>&1
done
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"
for V in $SQ_FLAGS; do
  NASTAR_FORWARD_FLAGS=$V timeout 300 rocprofv3 --pmc $SQ --kernel-trace -d $OUT/sq_f$V -o bench --output-format csv -- $B --steps 10 --warmup 2 > $OUT/sq_f$V.log 2>&1
  NASTAR_FORWARD_FLAGS=$V timeout 300 rocprofv3 --pmc $SQ2 --kernel-trace -d $OUT/sq2_f$V -o bench --output-format csv -- $B --steps 10 --warmup 2 > $OUT/sq2_f$V.log 2>&1
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
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/train -o train --output-format csv -- python /tmp/train_step.py > $OUT/train.log 2>&1
cat > /tmp/encoder_step.py <<PY
import sys
sys.path[:0] = ["$R/neural-astar_amd", "$R"]
import torch
from neural_astar.planner import NeuralAstar
from neural_astar.utils import synthetic as syn
dev = torch.device("cuda:0")
pr = syn.maze_maps(4096, 32, seed=1234)
m, s, g = (torch.from_numpy(x).to(dev).contiguous() for x in pr)
torch.manual_seed(0)
na = NeuralAstar(encoder_arch="CNN", encoder_depth=4).to(dev)
mode = sys.argv[1]
if mode == "train":
    na.train()
    na.encoder_backend = "hip_f16x3"
    for _ in range(4):
        for p in na.parameters():
            p.grad = None
        c = na.encode(m, s, g)
        (c * 1e-6).sum().backward()
else:
    na.eval()
    na.encoder_backend = "hip_" + mode
    with torch.no_grad():
        for _ in range(6):
            na.encode(m, s, g)
torch.cuda.synchronize()
PY
for M in f16x3 bf16 train; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/enc_$M -o enc --output-format csv -- python /tmp/encoder_step.py $M > $OUT/enc_$M.log 2>&1
done
python - <<PY
import csv, glob, json, collections
out = {}
for tag, pat in (("bench_kernel_stats", "$OUT/trace/**/*kernel_stats.csv"), ("bench_natural_order_kernel_stats", "$OUT/trace_natural/**/*kernel_stats.csv"),
                 ("train_step_kernel_stats", "$OUT/train/**/*kernel_stats.csv"), ("encoder_cnn_f16x3_infer_kernel_stats", "$OUT/enc_f16x3/**/*kernel_stats.csv"),
                 ("encoder_cnn_bf16_infer_kernel_stats", "$OUT/enc_bf16/**/*kernel_stats.csv"), ("encoder_cnn_f16x3_train_kernel_stats", "$OUT/enc_train/**/*kernel_stats.csv")):
    for f in glob.glob(pat, recursive=True):
        rows = list(csv.DictReader(open(f)))
        out[tag] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")} for r in rows[:14]]
        open("$OUT/%s.csv" % tag, "w").write(open(f).read())
def counters(pat):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in acc.items() if "nastar" in k}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    out[c] = counters("$OUT/pmc_%s/**/*counter_collection.csv" % c)
for V in "$SQ_FLAGS".split():
    d = counters("$OUT/sq_f%s/**/*counter_collection.csv" % V)
    d2 = counters("$OUT/sq2_f%s/**/*counter_collection.csv" % V)
    for k in d2:
        d.setdefault(k, {}).update(d2[k])
    out["sq_flags" + V] = d
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:5000])
PY
# the raw traces are hundreds of MB (gpurun copies back at most 64 MiB): keep the per-run kernel-stats CSVs + summary.json + the logs
for d in trace trace_natural train enc_f16x3 enc_bf16 enc_train; do rm -rf $OUT/$d; done
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/sq_f* $OUT/sq2_f* 2>/dev/null
find $OUT -maxdepth 1 -type d -name "sq*" -exec rm -rf {} + 2>/dev/null
du -sh $OUT
