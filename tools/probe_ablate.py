"""Dev probe (timing only): cost of the pieces of the compact forward step.  Each NASTAR_ABLATE value runs the 32x32 single-map
compact kernel for exactly 256 steps per map with one piece removed (results are garbage); 128 = nothing removed.
Usage (GPU box): python tools/probe_ablate.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path[:0] = [os.path.join(%r, "neural-astar_amd"), %r]
import numpy as np, torch
from neural_astar.utils import synthetic as syn
from neural_astar import ops
dev = torch.device("cuda:0")
mz = syn.maze_maps(64, 32, seed=1234)
one = syn.Problems(*(np.repeat(x[:1], 4096, 0) for x in mz))
out = []
for B in (1, 1024, 4096):
    m, s, g = (torch.from_numpy(x[:B, 0]).to(dev) for x in one)
    for _ in range(3):
        r = torch.ops.nastar.astar_forward(m, s, g, m, 0.5, 256, False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = torch.ops.nastar.astar_forward(m, s, g, m, 0.5, 256, False); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    out.append("B=%%d %%.1f us (%%.0f ns/step, iters %%d)" %% (B, min(ts) * 1e3, min(ts) * 1e6 / 256, int(r[2][0])))
print("  ".join(out))
''' % (ROOT, ROOT)
names = {128: "baseline (fixed 256 steps)", 129: "-h0", 130: "-chunk re-min DPP", 132: "-own-entry store", 136: "-atomic",
         144: "-g/pdir stores", 160: "-select DPP", 192: "-gc reads", 142: "-(remin,own,atomic)", 175: "-(h0,remin,own,atomic,selDPP)",
         255: "-everything"}
for a in (128, 129, 130, 132, 136, 144, 160, 192, 142, 175, 255):
    env = dict(os.environ, NASTAR_ABLATE=str(a), NASTAR_FORWARD_FLAGS="4")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"{a:4d} {names[a]:34s} {r.stdout.strip() or r.stderr.strip()[-300:]}")
