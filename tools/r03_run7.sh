#!/bin/bash
# round 3, GPU call: two-stage statistics: kernel tests, encoder training step timings (r02: 2.24 / 14.7 / 45.6 ms f16x3), train bench
mkdir -p gpurun_out/r03
python -m pytest tests/test_encoder_train_gpu.py tests/test_trainstep_golden_gpu.py tests/test_unet_gpu.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r03/t7.log
tail -3 gpurun_out/r03/t7.log
python tools/probe_train.py 100,1024,4096 hip_f16x3,hip_f16 2>&1 | grep -v Warn | tail -8
cp gpurun_out/probe_train.json gpurun_out/r03/probe_train_twostage.json
for c in maze warcraft; do
  python bench.py --mode train --config $c --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r03/train2_$c.json 2> gpurun_out/r03/train2_$c.err || tail -5 gpurun_out/r03/train2_$c.err
done
python - <<'P'
import json
for c in ("maze","warcraft"):
    j=json.load(open(f"gpurun_out/r03/train2_{c}.json")); print(c, round(j["ms_per_step"],3), round(j["value"]))
P
cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r03/prof_train_maze -o t --output-format csv -- python /root/repo/bench.py --mode train --config maze --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python - <<'P'
import csv, glob
for f in glob.glob("/root/repo/gpurun_out/r03/prof_train_maze/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
    print("total kernel ms", tot/1e6, "launches", calls, "per step (23 steps):", tot/1e6/23, calls/23)
    for r in rows[:14]: print(r["Name"][:64], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
P
