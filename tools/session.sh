#!/bin/bash
# GPU sessions as they were run (round 4 on): ONE parametrised script instead of a file per call.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/session.sh <name>'
# Every session writes under gpurun_out/<round>/; what is kept as evidence is copied into profiles/<round>/ afterwards.
set -u
S=${1:?session name}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
# r04_a only: a build of commit 87d8995's tree with the development kernels of rounds 1-3 (make DEV=1 BUILD=build_dev OUT=../lib/libnastar_hip_dev.so);
# those kernels -- two maps per wavefront among them -- were measured there one last time and then deleted (NOTES.md)
DEVLIB=$R/neural-astar_amd/lib/libnastar_hip_dev.so
case "$S" in
r06_final)
  # round 6, final tree: the whole GPU suite, the driver's bench command, rocprofv3 kernel stats + PMC traffic + encoder tables, the boundary probe
  # (native and Python host lanes), the N > 1 branches of bench.py with 8 gloo ranks sharing the GPU and a 1-rank RCCL group
  O=gpurun_out/r06/final; mkdir -p $O
  python -m pytest tests -q -m gpu > $O/gpu_tests_final.log 2>&1; echo "suite rc=$?"; tail -6 $O/gpu_tests_final.log
  python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_command.json 2> $O/bench_n1_driver_command.err; echo "bench rc=$?"
  python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench_n1_steps200.json 2>/dev/null
  bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1; tail -c 400 $O/profile_round.log
  python tools/probe_boundary.py > $O/probe_boundary.jsonl 2> $O/probe_boundary.err
  NASTAR_FASTLANE=0 python tools/probe_boundary.py 2>/dev/null | head -1 > $O/probe_boundary_python_lane.jsonl
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --steps 10 --warmup 2 --dist-backend gloo --share-gpu --no-cpu-baseline --no-secondary > $O/world8_maze32_weak.json 2> $O/world8.err; echo "world8 rc=$?"
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --force-collate --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_force_collate_rccl1.json 2> $O/rccl1.err; echo "rccl1 rc=$?"
  python - <<'P'
import json
for f in ("bench_n1_driver_command", "bench_n1_steps200", "world8_maze32_weak", "bench_force_collate_rccl1"):
    try:
        j = json.load(open(f"gpurun_out/r06/final/{f}.json"))
        print(f, {k: j.get(k) for k in ("value", "ms_per_step", "value_natural_order", "value_bare_launch", "n_gpus", "extras_note")}, "frac", j["roofline"]["frac"], "launch_ms", j["roofline"]["launch_ms_avg"], j["config"].get("distributed"))
    except Exception as e:
        print(f, "ERR", e)
P
  ;;
r04_a)
  # Throughput regime of the search: what binds it, and what two maps per wavefront can and cannot buy.
  #  1. aggregate issue rates of one CU (tools/ubench/rate.hip)
  #  2. streams sweep + one-launch big batches, shipped kernel, all three workloads
  #  3. DEV build: compiler-generated single-map step (flags 8) vs two maps per wavefront (flags 4) vs asm3 (flags 0), same probe
  #  4. asm3 with HALF the resident maps per CU (NASTAR_LDS_PAD: 8 wavefronts per CU) = the contention level of a duo kernel
  O=gpurun_out/r04/a; mkdir -p $O
  (cd tools/ubench && ./rate) > $O/rate.txt 2>&1
  python tools/probe_streams.py --workloads maze32,rand32,rand64 --flags 0 --streams 1,2,3,4,6,8 --bigb 2,4,8 > $O/streams_product.jsonl 2> $O/streams_product.err
  NASTAR_LIB=$DEVLIB python tools/probe_streams.py --workloads maze32,rand32 --flags 0,8,4 --streams 1,2,4,6 --bigb 4 > $O/streams_dev.jsonl 2> $O/streams_dev.err
  for pad in 3600 9984; do   # 12 and 8 wavefronts per CU instead of 16
    NASTAR_LIB=$DEVLIB NASTAR_LDS_PAD=$pad python tools/probe_streams.py --workloads maze32,rand32 --flags 0 --streams 1,2,4,6,8 --bigb 4 > $O/streams_pad$pad.jsonl 2> $O/streams_pad$pad.err
  done
  tail -n +1 $O/rate.txt $O/*.jsonl; tail -n 3 $O/*.err
  ;;
r04_b)
  # auto backend + 3-step training-loop golden + ADVICE fixes: the new test verbosely, then the whole GPU suite
  O=gpurun_out/r04/b; mkdir -p $O
  python -m pytest tests/test_trainloop_golden_gpu.py -q -s -m gpu > $O/trainloop.log 2>&1; echo "trainloop rc=$?"
  grep -a "TRAINLOOP\|passed\|failed\|Error\|assert" $O/trainloop.log | cut -c1-1500 | tail -20
  python -m pytest tests -q -m gpu > $O/all_gpu_tests.log 2>&1; echo "suite rc=$?"
  tail -25 $O/all_gpu_tests.log
  ;;
r04_c)
  # round-4 stream (asm4) + unit-cost kernel: parity first, then serial and in-flight throughput per flags value
  #   flags 0 = asm4 (default), 128 = asm3 (round 3), 64 = unit-cost layout (cost == passable in the probe / bench: same tensor)
  O=gpurun_out/r04/c; mkdir -p $O
  python -m pytest tests/test_gpu_parity.py -q -m gpu -k "instruction_streams or unit_cost or golden or oracle or planner_modules or unsolvable" > $O/parity.log 2>&1; echo "parity rc=$?"
  tail -15 $O/parity.log
  python tools/probe_streams.py --workloads maze32,rand32,rand64 --flags 0,128,64 --streams 1,2,4,6,8 --bigb 4,8 > $O/streams.jsonl 2> $O/streams.err
  cat $O/streams.jsonl; tail -n 3 $O/streams.err
  for f in 0 128 64; do for w in maze32 rand32 rand64; do
    NASTAR_FORWARD_FLAGS=$f python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 10 --workload $w > $O/serial_${w}_f$f.json 2>> $O/serial.err
  done; done
  python - <<'P'
import json
for w in ("maze32","rand32","rand64"):
    for f in (0,128,64):
        try:
            j=json.load(open(f"gpurun_out/r04/c/serial_{w}_f{f}.json")); print(w,"flags",f,round(j["value"]/1e6,2),"M maps/s", round(j["ms_per_step"]*1e3,1),"us/step", round(j["roofline"]["launch_ms_avg"]*1e3,1),"us launch avg")
        except Exception as e: print(w,f,"ERR",e)
P
  ;;
r04_prof)
  # evidence for the bench line: kernel stats + FETCH/WRITE PMC + SQ counters (round-4 stream, round-3 stream, unit-cost layout),
  # lone-wavefront step latency, and the bench line itself (driver's command)
  O=gpurun_out/r04/prof; mkdir -p $O
  bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1; tail -c 3000 $O/profile_round.log
  python tools/probe_latency.py > $O/lat_asm4.txt 2>&1; NASTAR_FORWARD_FLAGS=128 python tools/probe_latency.py > $O/lat_asm3.txt 2>&1
  head -4 $O/lat_asm4.txt $O/lat_asm3.txt
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_command.json 2> $O/bench_n1_driver_command.err; echo "bench rc=$?"
  tail -n 5 $O/bench_n1_driver_command.err; head -c 1500 $O/bench_n1_driver_command.json
  ;;
r04_d)
  # (1) tests touched since r04_b; (2) the driver's three N = 8 shapes with EIGHT ranks sharing this GPU over gloo (code paths, ports,
  # divisibility, synthesis time, memory -- the timings mean nothing); (3) 1-rank RCCL training step, sync-BN on / off
  O=gpurun_out/r04/d; mkdir -p $O
  python -m pytest tests/test_trainloop_golden_gpu.py tests/test_encoder_train_gpu.py tests/test_distributed_training.py tests/test_unet_gpu.py tests/test_trainstep_golden_gpu.py -q -s -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"
  grep -a "TRAINLOOP\|GRADERR unet\|ambiguous-only\|passed\|failed\|^E  " $O/tests.log | cut -c1-900 | tail -40
  run8() { name=$1; shift; port=$((29600 + RANDOM % 300)); t0=$(date +%s)
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 8 --steps 20 --warmup 5 --dist-backend gloo --share-gpu "$@" > $O/world8_$name.json 2> $O/world8_$name.err
    echo "world8 $name rc=$? $(( $(date +%s) - t0 )) s: $(head -c 300 $O/world8_$name.json)"; }
  run8 maze32_weak
  run8 rand64_contiguous --workload rand64 --global-batch 32768
  run8 rand64_interleaved --workload rand64 --global-batch 32768 --shard interleaved
  run8 train_warcraft --mode train --config warcraft
  run8 train_maze --mode train --config maze
  for c in maze warcraft; do
    python bench.py --mode train --config $c --no-cpu-baseline > $O/train_${c}.json 2> $O/train_${c}.err
    python bench.py --mode train --config $c --no-cpu-baseline --force-collate > $O/train_${c}_rccl1.json 2> $O/train_${c}_rccl1.err
    python - <<P
import json
a=json.load(open("$O/train_${c}.json")); b=json.load(open("$O/train_${c}_rccl1.json"))
print("$c", "single", round(a["ms_per_step"],3), "ms; 1-rank RCCL sync_bn on", round(b["ms_per_step"],3), "off", b.get("sync_bn"))
P
  done
  ;;
r04_e)
  # after the legacy (DEV) kernels and entry points were removed: the whole GPU suite, smoke, the training-loop golden verbosely
  O=gpurun_out/r04/e; mkdir -p $O
  python -m pytest tests -q -m gpu > $O/all_gpu_tests.log 2>&1; echo "suite rc=$?"; tail -8 $O/all_gpu_tests.log
  python -m pytest tests/test_trainloop_golden_gpu.py -q -s -m gpu > $O/trainloop.log 2>&1; echo "trainloop rc=$?"
  grep -a "TRAINLOOP\|passed\|failed\|^E  " $O/trainloop.log | cut -c1-1200 | tail -12
  python __graft_entry__.py smoke 2>&1 | tail -2
  ;;
r04_f)
  # per-map dive switch: parity, then serial timing with / without (flags 32); training-loop golden diagnostics; bench contract
  O=gpurun_out/r04/f; mkdir -p $O
  python -m pytest tests/test_gpu_parity.py -q -m gpu -k "instruction_streams or unit_cost or golden" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -4 $O/parity.log
  python -m pytest tests/test_trainloop_golden_gpu.py tests/test_bench_contract.py -q -s -m gpu > $O/trainloop.log 2>&1; echo "trainloop rc=$?"
  grep -a "TRAINLOOP\|passed\|failed" $O/trainloop.log | cut -c1-2500 | tail -8
  for rep in 1 2; do for f in 0 32 64 96; do for w in maze32 rand32; do
    NASTAR_FORWARD_FLAGS=$f python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 10 --workload $w > $O/serial_${w}_f${f}_$rep.json 2>> $O/serial.err
  done; done; done
  python - <<'P'
import json
for w in ("maze32","rand32"):
    for f in (0,32,64,96):
        for rep in (1,2):
            try:
                j=json.load(open(f"gpurun_out/r04/f/serial_{w}_f{f}_{rep}.json")); print(w,"flags",f,"rep",rep,round(j["value"]/1e6,2),"M maps/s", round(j["ms_per_step"]*1e3,1),"us/step", round(j["roofline"]["launch_ms_avg"]*1e3,1),"us launch avg")
            except Exception as e: print(w,f,"ERR",e)
P
  ;;
r04_final)
  # the final build: whole GPU suite, smoke, profile round, the driver's bench command
  O=gpurun_out/r04/final; mkdir -p $O
  python -m pytest tests -q -m gpu > $O/all_gpu_tests.log 2>&1; echo "suite rc=$?"; tail -4 $O/all_gpu_tests.log
  python __graft_entry__.py smoke 2>&1 | tail -1
  bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_command.json 2> $O/bench_n1_driver_command.err; echo "bench rc=$?"
  tail -n 3 $O/bench_n1_driver_command.err; head -c 600 $O/bench_n1_driver_command.json
  ;;
r04_tp)
  # throughput regime under rocprofv3: ONE launch of 32768 maps per step (kernel duration -> roofline fraction by the bench line's own
  # definition), general and unit-cost kernels, all workloads; FETCH / WRITE traffic of the rand64 unit-cost launch
  O=$R/gpurun_out/r04/tp; mkdir -p $O
  cd /tmp && export TMPDIR=/tmp
  for f in 0 64; do
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_f$f -o tp --output-format csv -- python $R/tools/probe_streams.py --workloads maze32,rand32,rand64 --flags $f --streams 1 --bigb 8 --steps 80 > $O/probe_f$f.jsonl 2> $O/probe_f$f.err
  done
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_$C -o tp --output-format csv -- python $R/tools/probe_streams.py --workloads rand64 --flags 64 --streams 1 --bigb 8 --steps 40 > $O/pmc_$C.jsonl 2> $O/pmc_$C.err
  done
  python - <<P
import csv, glob, collections
for f in (0, 64):
    for path in glob.glob("$O/trace_f%d/**/*kernel_stats.csv" % f, recursive=True):
        for r in list(csv.DictReader(open(path)))[:4]:
            print("flags", f, r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(path)):
            acc[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: (len(v), sum(v) / len(v)) for c, v in d.items()})
P
  ;;
r04_h)
  # NASTAR_FLAG_PERSISTENT (256; existed at commit 201811d only -- measured, dropped): parity on batches larger than the resident capacity,
  # then one 32768-map launch and 4096-map launches in flight with / without, general (0 / 256) and unit-cost (64 / 320) kernels
  O=gpurun_out/r04/h; mkdir -p $O
  python -m pytest tests/test_gpu_parity.py -q -m gpu -k "persistent or packed" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
  python tools/probe_streams.py --workloads maze32,rand32,rand64 --flags 0,256,64,320 --streams 1,4 --bigb 2,8 --steps 160 > $O/streams.jsonl 2> $O/streams.err
  cat $O/streams.jsonl; tail -n 2 $O/streams.err
  ;;
r04_enc)
  # encoder training / inference census: per-kernel time of the f16x3 4096-map and 100-map encoder steps
  O=gpurun_out/r04/enc${2:-}; mkdir -p $O
  export TMPDIR=/tmp
  python tools/probe_train.py 100,4096 hip_f16x3 > $O/probe_train.txt 2>&1; tail -3 $O/probe_train.txt
  for B in 4096 100; do
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_b$B -o t --output-format csv -- python tools/probe_train.py $B hip_f16x3 > $O/trace_b$B.txt 2>&1
    python - <<P
import csv, glob
for path in glob.glob("$O/trace_b$B/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("B=$B total kernel ms over 6 steps", round(tot / 1e6, 2))
    for r in rows[:26]:
        print("%-100s %4s %9.1f us avg %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
P
  done
  ;;
r04_spec)
  # (flags 256 / 512 and csrc/nastar_search_spec.hip.h existed at commit 38d47f3 only -- measured, dropped: NOTES.md "Round 4")
  # two selections per step (256) and a selection that leaves the critical path (512), compiler-generated: stream equality on whole
  # batches, then the serial launch and batches in flight against the shipped stream (0) and hipcc's single-selection step (8)
  O=gpurun_out/r04/spec${2:-}; mkdir -p $O
  python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "instruction_streams" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -5 $O/parity.log
  for f in 0 8 256 512; do for w in maze32 rand32; do
    NASTAR_FORWARD_FLAGS=$f python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 10 --workload $w > $O/serial_${w}_f$f.json 2>> $O/serial.err
  done; done
  python - <<P
import json
for w in ("maze32","rand32"):
    for f in (0,8,256,512):
        try:
            j=json.load(open("$O/serial_%s_f%d.json" % (w, f))); print(w,"flags",f,round(j["value"]/1e6,2),"M maps/s", round(j["ms_per_step"]*1e3,1),"us/step", round(j["roofline"]["launch_ms_avg"]*1e3,1),"us launch avg")
        except Exception as e: print(w,f,"ERR",e)
P
  python tools/probe_streams.py --workloads maze32,rand32 --flags 0,512 --streams 1,4 --bigb 4 > $O/streams.jsonl 2> $O/streams.err
  cat $O/streams.jsonl; tail -n 2 $O/streams.err
  ;;
r04_order)
  # placement by known step counts (nastar_forward_ordered): us per 4096-map launch for several orders, general and unit-cost kernels
  O=gpurun_out/r04/order${2:-}; mkdir -p $O
  python tools/probe_order.py maze32,rand32,rand64 0 > $O/order_f0.jsonl 2> $O/order_f0.err; cat $O/order_f0.jsonl; tail -n 2 $O/order_f0.err
  ;;
r04_order2)
  # placement, wired: parity tests, the lone-wavefront step at working clocks, the bench line (hinted headline + natural order beside it)
  O=gpurun_out/r04/order2${2:-}; mkdir -p $O
  python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "placement" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -5 $O/parity.log
  python tools/probe_latency.py > $O/lat.txt 2>&1; tail -4 $O/lat.txt
  python bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
  python - <<P
import json
j = json.load(open("$O/bench_driver_command.json"))
print({k: j[k] for k in ("value", "ms_per_step")}, j["roofline"]["launch_ms_avg"], j["roofline"]["frac"], j["natural_order"])
for s in j["secondary"]: print(s["workload"], s["launch_ms_avg"], s["hbm_frac"], s.get("launch_ms_avg_natural_order"), s.get("hbm_frac_natural_order"))
print(j["issue_model"])
P
  ;;
r04_final2)
  # the record after the placement work: whole GPU suite, lone-wavefront step at working clocks, training pair with placements,
  # the driver's bench command, the profile round (kernel stats hinted + natural, PMC, SQ, training step)
  O=gpurun_out/r04/final2; mkdir -p $O
  python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_tests.log
  python __graft_entry__.py smoke 2>&1 | tail -1
  python tools/probe_latency.py > $O/lat_working_clocks.txt 2>&1; tail -8 $O/lat_working_clocks.txt
  python tools/probe_order_train.py > $O/order_train_pair.json 2> $O/order_train_pair.err; cat $O/order_train_pair.json
  python tools/probe_order.py maze32,rand32,rand64 0 > $O/order_placements_probe.jsonl 2>/dev/null
  python bench.py --steps 20 --warmup 5 > $O/bench_n1_driver_command.json 2> $O/bench.err; echo "bench rc=$?"
  bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1; tail -c 600 $O/profile_round.log
  ;;
r05_a)
  # 0.5.0 boundary: new tests first (pinned status summary, checked orders, fast path, data-set placement, InFlightPlanner), then the
  # bench line (driver's command shape, short) and a clean first-visit kernel profile, then the whole GPU suite
  O=gpurun_out/r05/a; mkdir -p $O
  python -m pytest tests/test_boundary_gpu.py -q -x -m gpu > $O/boundary.log 2>&1; echo "boundary rc=$?"
  tail -30 $O/boundary.log
  python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $O/bench_short.json 2> $O/bench_short.err; echo "bench rc=$?"
  tail -5 $O/bench_short.err
  python - <<'P'
import json
j=json.load(open("gpurun_out/r05/a/bench_short.json"))
print("value", round(j["value"]/1e6,2), "M maps/s  ms/step", round(j["ms_per_step"],4), "natural", j["value_natural_order"] and round(j["value_natural_order"]/1e6,2), "hinted", j["value_hinted"] and round(j["value_hinted"]/1e6,2))
print("roofline", {k: j["roofline"][k] for k in ("frac","frac_28B_per_cell","launch_ms_avg","launch_ms_median","launch_ms_min")})
print("first visits:", j["config"]["every_timed_step_is_a_first_visit"], j["config"]["first_visit_note"])
P
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o bench --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --no-natural --no-prewarm --steps 200 --warmup 20 > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err)
  find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/first_visit_kernel_stats.csv \;
  head -5 $O/first_visit_kernel_stats.csv
  python -m pytest tests -q -m gpu -x > $O/all_gpu_tests.log 2>&1; echo "suite rc=$?"
  tail -15 $O/all_gpu_tests.log
  ;;
r05_b)
  # boundary tests again, the whole GPU suite (no -x), then the driver's bench command in full (through_module, in-flight through the API, secondaries)
  O=gpurun_out/r05/b; mkdir -p $O
  python -m pytest tests/test_boundary_gpu.py -q -m gpu > $O/boundary.log 2>&1; echo "boundary rc=$?"; tail -5 $O/boundary.log
  python -m pytest tests -q -m gpu > $O/all_gpu_tests.log 2>&1; echo "suite rc=$?"; tail -8 $O/all_gpu_tests.log
  python bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
  python - <<'P'
import json
j=json.load(open("gpurun_out/r05/b/bench_driver_command.json"))
print("value", round(j["value"]/1e6,2), "M maps/s  ms/step", round(j["ms_per_step"],4), "natural", j["value_natural_order"] and round(j["value_natural_order"]/1e6,2), "hinted", j["value_hinted"] and round(j["value_hinted"]/1e6,2), "in flight", j.get("value_in_flight"))
print("roofline", {k: j["roofline"][k] for k in ("frac","frac_28B_per_cell","launch_ms_avg","launch_ms_median","launch_ms_min")})
print("through_module", j.get("through_module"))
print("in_flight", json.dumps(j.get("in_flight_through_api"), indent=0))
print("cpu_baseline", {k: v for k, v in j.get("cpu_baseline", {}).items() if k in ("value","unit","cores","kind","sample")})
P
  ;;
r05_c)
  # host-time breakdown of the boundary (forward() vs raw launch; InFlightPlanner vs raw round-robin), then the tests that failed in r05_b
  O=gpurun_out/r05/c; mkdir -p $O
  python tools/probe_boundary.py > $O/probe_boundary.jsonl 2> $O/probe_boundary.err; echo "probe rc=$?"; tail -3 $O/probe_boundary.err; cat $O/probe_boundary.jsonl
  python -m pytest tests/test_boundary_gpu.py tests/test_tie_class.py -q -m gpu > $O/boundary.log 2>&1; echo "boundary rc=$?"; tail -5 $O/boundary.log
  python -m pytest tests/test_gpu_parity.py -q -m gpu -k "unit_cost or placement or unsolvable or hipgraph or planner_modules" > $O/parity_subset.log 2>&1; echo "parity rc=$?"; tail -5 $O/parity_subset.log
  ;;
r05_d)
  # completion flag + lean boundary: probe again; large maps: parity (oracle at 256^2 / 512^2, hybrid vs round-4 kernel), ns per step; then the suite
  O=gpurun_out/r05/d; mkdir -p $O
  python -m pytest tests/test_large_maps_gpu.py tests/test_boundary_gpu.py tests/test_tie_class.py -q -m gpu > $O/new_tests.log 2>&1; echo "new tests rc=$?"; tail -8 $O/new_tests.log
  python tools/probe_large.py > $O/probe_large.jsonl 2> $O/probe_large.err; echo "large rc=$?"; tail -2 $O/probe_large.err; cat $O/probe_large.jsonl
  python tools/probe_boundary.py > $O/probe_boundary.jsonl 2> $O/probe_boundary.err; echo "probe rc=$?"; tail -3 $O/probe_boundary.err; cat $O/probe_boundary.jsonl
  python -m pytest tests -q -m gpu > $O/all_gpu_tests.log 2>&1; echo "suite rc=$?"; tail -8 $O/all_gpu_tests.log
  ;;
r05_e)
  # after the fixes: whole GPU suite, large-map probe (fill / search / store launches), 8 gloo ranks sharing the GPU (every world > 1 branch of bench.py)
  O=gpurun_out/r05/e; mkdir -p $O
  python -m pytest tests -q -m gpu > $O/all_gpu_tests.log 2>&1; echo "suite rc=$?"; tail -12 $O/all_gpu_tests.log | cut -c1-300
  python tools/probe_large.py > $O/probe_large.jsonl 2> $O/probe_large.err; echo "large rc=$?"; cat $O/probe_large.jsonl
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 8 --steps 10 --warmup 3 --dist-backend gloo --share-gpu --no-secondary --no-cpu-baseline > $O/world8_maze32_weak.json 2> $O/world8_maze32_weak.err; echo "world8 weak rc=$?"; tail -2 $O/world8_maze32_weak.err; cut -c1-600 $O/world8_maze32_weak.json
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 8 --steps 10 --warmup 3 --dist-backend gloo --share-gpu --no-secondary --no-cpu-baseline --workload rand64 --global-batch 32768 --shard interleaved > $O/world8_rand64_interleaved.json 2> $O/world8_rand64_interleaved.err; echo "world8 rand64 rc=$?"; tail -2 $O/world8_rand64_interleaved.err; cut -c1-600 $O/world8_rand64_interleaved.json
  ;;
r05_prof)
  # the record of round 5 on the final tree: GPU suite + smoke, the driver's bench command, a clean FIRST-VISIT kernel profile (every launch of the
  # run is a dataset-placed search of a never-searched batch), natural-order profile, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes),
  # large-map and boundary probes
  O=gpurun_out/r05/prof; mkdir -p $O
  python -m pytest tests -q -m gpu > $O/gpu_tests_final.log 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_tests_final.log | cut -c1-200
  python __graft_entry__.py smoke 2>&1 | tail -1
  python bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
  B="python $R/bench.py --no-cpu-baseline --no-secondary --no-natural --no-prewarm"
  (cd /tmp && export TMPDIR=/tmp
   timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o bench --output-format csv -- $B --steps 200 --warmup 20 > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err
   timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_natural -o bench --output-format csv -- $B --placement natural --steps 200 --warmup 20 > $R/$O/bench_natural_under_rocprof.json 2> $R/$O/bench_natural_under_rocprof.err
   for C in FETCH_SIZE WRITE_SIZE; do
     timeout 300 rocprofv3 --pmc $C --kernel-trace -d $R/$O/pmc_$C -o bench --output-format csv -- $B --steps 20 --warmup 2 > $R/$O/pmc_$C.log 2>&1
   done)
  find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/bench_first_visit_kernel_stats.csv \;
  find $O/trace_natural -name "*kernel_stats.csv" -exec cp {} $O/bench_natural_order_kernel_stats.csv \;
  python - <<'P'
import csv, glob, json, collections
O = "gpurun_out/r05/prof"
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "nastar_forward_compact" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[c] = {k: {"n": len(v), "mean": sum(v) / len(v)} for k, v in acc.items()}
json.dump(out, open(f"{O}/pmc_summary.json", "w"), indent=1)
print(json.dumps(out))
for name in ("bench_first_visit_kernel_stats.csv", "bench_natural_order_kernel_stats.csv"):
    try:
        print(name, open(f"{O}/{name}").read().splitlines()[1][:200])
    except Exception as e:
        print(name, "ERR", e)
j = json.load(open(f"{O}/bench_driver_command.json"))
print("value", round(j["value"] / 1e6, 2), "M maps/s  ms/step", round(j["ms_per_step"], 4), "natural", round(j["value_natural_order"] / 1e6, 2), "hinted", round(j["value_hinted"] / 1e6, 2), "in flight", j.get("value_in_flight"))
print("roofline", {k: j["roofline"][k] for k in ("frac", "frac_28B_per_cell", "launch_ms_avg")}, "through_module", j.get("through_module"))
for r in j.get("in_flight_through_api", []):
    print(r["workload"], r["kernel"], {k: round(v / 1e6, 1) for k, v in r["streams_sweep_maps_per_s"].items()}, r["equal_to_sequential_forward"])
P
  python tools/probe_large.py > $O/probe_large.jsonl 2> $O/probe_large.err; echo "large rc=$?"; grep '"B": 256' $O/probe_large.jsonl | cut -c1-260
  python tools/probe_boundary.py > $O/probe_boundary.jsonl 2> $O/probe_boundary.err; echo "probe rc=$?"; head -1 $O/probe_boundary.jsonl | cut -c1-1200
  ;;
r05_f)
  # the shipped hybrid kernel (VGPR reductions + FMA division): parity and step time once more
  O=gpurun_out/r05/f; mkdir -p $O
  python -m pytest tests/test_large_maps_gpu.py tests/test_boundary_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
  python tools/probe_large.py > $O/probe_large.jsonl 2> $O/probe_large.err; echo "large rc=$?"; grep -v '"B": 1,' $O/probe_large.jsonl | cut -c1-260
  ;;
r05_g)
  # final tree: whole GPU suite (incl. the status-verdict stress test), the collated path in a 1-rank RCCL group, both training benches
  O=gpurun_out/r05/g; mkdir -p $O
  python -m pytest tests -q -m gpu > $O/gpu_tests_final.log 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_tests_final.log | cut -c1-200
  python bench.py --force-collate --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_force_collate.json 2> $O/bench_force_collate.err; echo "collate rc=$?"; tail -2 $O/bench_force_collate.err; cut -c1-300 $O/bench_force_collate.json
  for c in maze warcraft; do
    python bench.py --mode train --config $c --steps 20 --warmup 5 --no-cpu-baseline > $O/train_$c.json 2> $O/train_$c.err; echo "train $c rc=$?"; cut -c1-260 $O/train_$c.json
  done
  ;;
r05_fuzz)
  # randomised parity sweep of the search against the oracle (sizes around every kernel boundary, costs, budgets, logs, placements)
  O=gpurun_out/r05/fuzz; mkdir -p $O
  for seed in 20260926 7 8; do
    python tools/fuzz_parity.py $seed ${2:-1500} 0.2 > $O/fuzz_parity_$seed.jsonl 2> $O/fuzz_parity_$seed.err; echo "fuzz $seed rc=$?"; tail -1 $O/fuzz_parity_$seed.err; tail -6 $O/fuzz_parity_$seed.jsonl
  done
  python -m pytest tests/test_fuzz_parity_gpu.py -q -m gpu 2>&1 | tail -2
  ;;
r05_final)
  # the final tree: GPU suite, smoke, the driver's bench command
  O=gpurun_out/r05/final; mkdir -p $O
  python -m pytest tests -q -m gpu > $O/gpu_tests_final.log 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_tests_final.log | cut -c1-200
  python __graft_entry__.py smoke 2>&1 | tail -1
  python bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
  python - <<'P'
import json
j = json.load(open("gpurun_out/r05/final/bench_driver_command.json"))
print("value", round(j["value"] / 1e6, 2), "M maps/s  ms/step", round(j["ms_per_step"], 4), "natural", round(j["value_natural_order"] / 1e6, 2), "hinted", round(j["value_hinted"] / 1e6, 2), "in flight", j.get("value_in_flight"))
print("roofline", {k: j["roofline"][k] for k in ("frac", "frac_28B_per_cell", "launch_ms_avg")}, "through_module", {k: round(v, 4) for k, v in j.get("through_module", {}).items() if isinstance(v, float)})
for r in j.get("in_flight_through_api", []):
    print(r["workload"], r["kernel"], {k: round(v / 1e6, 1) for k, v in r["streams_sweep_maps_per_s"].items()}, r["equal_to_sequential_forward"])
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
P
  ;;
r05_h)
  # foreign-stream lifetimes of InFlightPlanner + forward() against the literal batch loop incl. signed costs
  O=gpurun_out/r05/h; mkdir -p $O
  python -m pytest tests/test_boundary_gpu.py -q -x -k "in_flight" > $O/in_flight.log 2>&1; echo "in_flight rc=$?"; tail -15 $O/in_flight.log | cut -c1-220
  python - > $O/module_signed.jsonl 2>&1 <<'P'
import json, sys
sys.path.insert(0, "tools")
import fuzz_parity
for seed in (9, 31, 77):
    n, rr, bad = fuzz_parity.run_module(seed=seed, N=150, verbose=True)
    print(json.dumps({"seed": seed, "module_vs_literal_batch_loop_cases": n, "batches_in_the_coupled_class": rr, "module_failures": len(bad)}))
P
  echo "module rc=$?"; tail -12 $O/module_signed.jsonl | cut -c1-220
  ;;
r05_i)
  # hybrid large-map kernel: A/B of the memory-path / scalar-control-flow variants, parity under the most aggressive ones
  O=gpurun_out/r05/i; mkdir -p $O
  timeout 200 python tools/probe_large.py variants > $O/probe_large_variants.jsonl 2> $O/probe.err; echo "probe rc=$?"; tail -2 $O/probe.err
  python - <<'P'
import json
for l in open("gpurun_out/r05/i/probe_large_variants.jsonl"):
    r = json.loads(l)
    print(r["H"], r["B"], r["cost"], "flags", r["flags"], "ms", round(r["launch_ms"], 3), "ns/step", round(r["ns_per_step_of_longest"]), "equal", r["equal_to_default"])
P
  for F in 22528 14336; do
    NASTAR_FORWARD_FLAGS=$F timeout 400 python -m pytest tests/test_large_maps_gpu.py tests/test_fuzz_parity_gpu.py -q -x -k "not gradients and not module" > $O/parity_flags_$F.log 2>&1; echo "parity flags=$F rc=$?"; tail -3 $O/parity_flags_$F.log | cut -c1-200
  done
  ;;
r05_final2)
  # the final tree after the hybrid kernel's L1 path became the default: GPU suite, smoke, the driver's bench command, then the hybrid A/B
  O=gpurun_out/r05/final2; mkdir -p $O
  timeout 240 python -m pytest tests -q -m gpu > $O/gpu_tests_final.log 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_tests_final.log | cut -c1-200
  timeout 60 python __graft_entry__.py smoke 2>&1 | tail -1
  timeout 110 python bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
  python - <<'P'
import json
try:
    j = json.load(open("gpurun_out/r05/final2/bench_driver_command.json"))
    print("value", round(j["value"] / 1e6, 2), "M maps/s  ms/step", round(j["ms_per_step"], 4), "natural", round(j["value_natural_order"] / 1e6, 2), "hinted", round(j["value_hinted"] / 1e6, 2))
    print("roofline", {k: j["roofline"][k] for k in ("frac", "frac_28B_per_cell", "launch_ms_avg")}, "through_module", {k: round(v, 4) for k, v in j.get("through_module", {}).items() if isinstance(v, float)})
except Exception as e:
    print("bench line unreadable", e)
P
  timeout 90 python tools/probe_large.py variants > $O/probe_large_variants.jsonl 2> $O/probe.err; echo "probe rc=$?"
  python - <<'P'
import json
for l in open("gpurun_out/r05/final2/probe_large_variants.jsonl"):
    r = json.loads(l)
    print(r["H"], r["B"], r["cost"], "flags", r["flags"], "ms", round(r["launch_ms"], 3), "ns/step", round(r["ns_per_step_of_longest"]), "equal", r["equal_to_default"])
P
  ;;
*)
  echo "unknown session $S"; exit 2;;
esac
