#!/bin/bash
# round 3, GPU call: which of the round-3b step's changes pay on which workload.  NASTAR_FORWARD_FLAGS: 32 = plain round-3 loop,
# 0 = 3b (exit tests in the LDS shadow + VALU prefix + rotation row stage + 2 steps per loop), 64 = exit tests + 2 steps per loop only,
# 128 = exit tests + VALU prefix + 2 steps per loop.  Equality of all outputs against the plain loop first.
mkdir -p gpurun_out/r03
python - <<'P'
import os, sys, numpy as np, torch
sys.path[:0] = ["neural-astar_amd", "."]
import bench
dev = torch.device("cuda:0")
for wl in ("maze32", "rand32"):
    pr = bench.make_problem(wl, 4096, seed=1234)
    ref = None
    for f in (32, 0, 64, 128):
        os.environ["NASTAR_FORWARD_FLAGS"] = str(f)
        run = bench.Runner(pr, dev); run.step(); torch.cuda.synchronize()
        cur = [x.clone() for x in (run.hist, run.paths, run.iters, run.status)]
        if ref is None: ref = cur
        else: print(wl, "flags", f, "equal to plain:", all(torch.equal(a, b) for a, b in zip(ref, cur)))
P
for f in 32 0 64 128; do for w in maze32 rand32; do
  NASTAR_FORWARD_FLAGS=$f python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 10 --workload $w > gpurun_out/r03/v3b_${w}_f$f.json 2>> gpurun_out/r03/v3b.err
done; done
python - <<'P'
import json
for w in ("maze32","rand32"):
    for f in (32,0,64,128):
        j=json.load(open(f"gpurun_out/r03/v3b_{w}_f{f}.json")); print(w,f,round(j["value"]/1e6,2),"M maps/s", round(j["ms_per_step"]*1e3,1), round(j["roofline"]["launch_ms_median"]*1e3,1),"us median", round(j["roofline"]["frac"],4))
P
for f in 32 0 64 128; do echo flags $f; NASTAR_FORWARD_FLAGS=$f timeout 200 python tools/probe_latency.py 2>&1 | grep "fixture32\|maze32 B=256\|B=4096"; done
