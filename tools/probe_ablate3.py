"""Dev probe (timing only, `make DEV=1` build): what each section of the round-3 step loop contributes to the step's critical path.
Each NASTAR_ABLATE = 300 + V runs the 32x32 kernel for exactly 256 steps per map, both exits disabled, with one section removed or
altered (nastar_search_asm3_abl.hip.h; results are garbage); V = 0 removes nothing.  Lone wavefront (B = 1), one per SIMD (B = 1024)
and the full chip (B = 4096 copies of the same map).
Usage (GPU box): python tools/probe_ablate3.py"""
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
    for _ in range(5):
        r = torch.ops.nastar.astar_forward(m, s, g, m, 0.5, 256, False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = torch.ops.nastar.astar_forward(m, s, g, m, 0.5, 256, False); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    out.append("B=%%d %%.1f us (%%.1f ns/step)" %% (B, min(ts) * 1e3, min(ts) * 1e6 / 256))
print("  ".join(out))
''' % (ROOT, ROOT)
names = {0: "baseline (256 steps, exits disabled)", 1: "- 4 row-level DPP stages", 2: "- whole reduction", 3: "- pick: cmp + ff1",
         4: "- heuristic (10 VALU)", 5: "- key arithmetic (9 VALU)", 6: "- relax: 2 stores + atomic", 7: "- relax: atomic only",
         8: "- close s* (2 stores + 2 exec)", 9: "- cell read + its wait", 10: "- cmin read-back", 11: "+ one taken branch",
         12: "- reduction wait states", 13: "- everything after the pick", 14: "skeleton (counter, read-back, loop)",
         15: "- relax: the 2 plain stores", 16: "close s* without exec switches", 17: "- address prefix", 18: "- both exit compares"}
for v in sorted(names):
    env = dict(os.environ, NASTAR_ABLATE=str(300 + v))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"{v:3d} {names[v]:40s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
