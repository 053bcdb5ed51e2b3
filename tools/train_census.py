"""Timeline of ONE training step: every kernel launch in order with its duration and the idle gap before it (rocprofv3 kernel trace).
Usage (GPU box): python tools/train_census.py run maze|warcraft|unet [B]      -- the traced program (run under rocprofv3)
                 python tools/train_census.py parse <kernel_trace.csv> [steps] -- the census of the middle step"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, WARM = 8, 4


def run(cfg_name, B):
    import torch
    sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
    import bench  # noqa: F401
    import bench_extras
    from neural_astar.planner import NeuralAstar
    from neural_astar.utils import distributed as D
    dev = torch.device("cuda:0")
    kw = dict(encoder_input="m+", encoder_arch="Unet", encoder_depth=4, Tmax=0.25) if cfg_name == "unet" else bench_extras.TRAIN_CONFIGS[cfg_name]["kw"]
    batches = [bench_extras.train_batch("maze" if cfg_name == "unet" else cfg_name, B, 1234 + 1000 * k, dev) for k in range(2)]
    torch.manual_seed(1234)
    p = NeuralAstar(**kw).to(dev)
    p.encoder_backend = "hip_f16x3"
    t = D.DataParallelTrainer(p, lr=1e-3, coupling="local")
    for i in range(WARM):
        t.train_step(*batches[i % 2])
    torch.cuda.synchronize()
    marker = torch.zeros(1, device=dev)
    for i in range(STEPS):
        marker.add_(1.0).sin_()  # two elementwise launches that mark the step boundary in the trace (add_ + sin_ back to back)
        t.train_step(*batches[i % 2])
    torch.cuda.synchronize()


def parse(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # step boundaries: an elementwise add followed directly by a sin kernel
    marks = [i for i in range(len(rows) - 1) if "sin_kernel" in names[i + 1] and "CUDAFunctorOnSelf_add" in names[i]]
    marks = marks[-STEPS:]
    k = len(marks) // 2
    lo, hi = marks[k] + 2, marks[k + 1]
    step = rows[lo:hi]
    t0 = int(rows[marks[k] + 1]["End_Timestamp"])
    busy = 0
    agg = {}
    prev_end = t0
    print(f"step {k}: {len(step)} launches")
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = s - prev_end
        busy += e - s
        n = r["Kernel_Name"]
        short = n.split("(")[0].replace("void ", "")[:70]
        a = agg.setdefault(short, [0, 0, 0])
        a[0] += 1; a[1] += e - s; a[2] += max(gap, 0)
        prev_end = max(prev_end, e)
    wall = prev_end - t0
    print(f"wall {wall / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, idle gaps {(wall - busy) / 1e3:.1f} us")
    print(f"{'kernel':72s} {'n':>4s} {'busy us':>9s} {'gap-before us':>13s}")
    for short, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print(f"{short:72s} {a[0]:4d} {a[1] / 1e3:9.1f} {a[2] / 1e3:13.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 100)
    else:
        parse(sys.argv[2])
