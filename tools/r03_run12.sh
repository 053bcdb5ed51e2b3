#!/bin/bash
# round 3, GPU call: deferred verdict on a side stream: tests + through_module timing
python -m pytest tests/test_gpu_parity.py tests/test_data_path.py tests/test_trainstep_golden_gpu.py -m gpu -q -x -k "unsolvable or hipgraph or planner or fused or validation or trainstep or module" 2>&1 | tail -2
python - <<'P'
import sys
sys.path[:0]=["/root/repo","/root/repo/neural-astar_amd"]
import torch, bench
dev=torch.device("cuda:0")
pr=bench.make_problem("maze32",4096,1234)
print(bench.through_module_ms(pr, dev, reps=60))
P
