import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/neural-astar_amd"]
os.environ["NASTAR_BENCH_DEBUG"] = "1"
import torch, bench
mode = sys.argv[1]
dev = torch.device("cuda:0")
t00 = time.perf_counter()
if "sleep" in mode:
    time.sleep(11)
prs = [bench.make_problem("maze32", 4096, seed=1234 + 1000 * k) for k in range(3)]
print("problems", round(time.perf_counter() - t00, 1), flush=True)
run = bench.Runner(prs, dev, placement="hinted")
for rep in range(3):
    bench.timed_loop(run, 20, 5, 1, dev)
    if "noprewarm" not in mode:
        bench.prewarm(run, dev)
    bench.timed_loop(run, 20, 5, 1, dev)
    if "gc" in mode:
        import gc; gc.collect()
