import os, sys, json, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import numpy as np, torch
from neural_astar.planner.differentiable_astar import DifferentiableAstar
z = np.load(sys.argv[1])
dev = torch.device("cuda:0")
c, s, g, m = (torch.from_numpy(np.ascontiguousarray(z[k])).to(dev) for k in ("cost", "start", "goal", "passable"))
gr, Tmax, train = float(z["g_ratio"]), float(z["Tmax"]), bool(z["train"])
for mode in (True, "deferred", False):
    for grad in (False, True):
        da = DifferentiableAstar(gr, Tmax, check_solvable=mode).to(dev).train(train)
        try:
            if grad:
                cg = c.clone().requires_grad_(True)
                out = da(cg, s, g, m)
                out.histories.sum().backward()
            else:
                with torch.no_grad():
                    out = da(c, s, g, m)
            da.raise_if_unsolvable()
            h = out.histories[:, 0].detach().cpu().numpy(); p = out.paths[:, 0].cpu().numpy()
            res = {"mode": str(mode), "grad": grad,
                   "hist_diff_per_map": [int((h[b] != z["histories"][b]).sum()) for b in range(h.shape[0])],
                   "path_diff_per_map": [int((p[b] != z["paths"][b]).sum()) for b in range(h.shape[0])],
                   "hist_vs_sm": [int((h[b] != z["sm_histories"][b]).sum()) for b in range(h.shape[0])],
                   "hist_sum": [int(h[b].sum()) for b in range(h.shape[0])], "ref_sum": [int(z["histories"][b].sum()) for b in range(h.shape[0])]}
        except Exception as e:
            res = {"mode": str(mode), "grad": grad, "error": f"{type(e).__name__}: {e}"[:200]}
        print(json.dumps(res), flush=True)
