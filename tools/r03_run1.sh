#!/bin/bash
# round 3, GPU call: reference-generated training-step goldens, sync-BN equivalence, the new training bench mode (incl. the RCCL path in a 1-rank group)
mkdir -p gpurun_out/r03
python -m pytest tests/test_trainstep_golden_gpu.py tests/test_distributed_training.py -m gpu -q -s 2>&1 | grep -v Warning > gpurun_out/r03/t1.log
tail -25 gpurun_out/r03/t1.log
for c in maze warcraft; do
  python bench.py --mode train --config $c --steps 40 --warmup 5 > gpurun_out/r03/train_$c.json 2> gpurun_out/r03/train_$c.err || tail -5 gpurun_out/r03/train_$c.err
  python bench.py --mode train --config $c --steps 40 --warmup 5 --force-collate --no-cpu-baseline > gpurun_out/r03/train_${c}_rccl1.json 2> gpurun_out/r03/train_${c}_rccl1.err || tail -5 gpurun_out/r03/train_${c}_rccl1.err
done
cut -c1-400 gpurun_out/r03/train_*.json
