#!/usr/bin/env python3
"""bench_extras.py -- every SECONDARY experiment of the bench line, in a process of its own.

`bench.py` computes the line's core (headline, roofline, cpu_baseline) and then runs this file as a CHILD process with a time limit
(bench.run_extras): a crash, a hang or an out-of-memory here cannot take the one line that matters with it.  This
process prints one JSON object per FINISHED section on stdout (the parent merges them):

  through_module                 VanillaAstar.forward() per call: same-call verdict / deferred / unchecked, with and without the loader's hint
  hinted                         recurring batches placed by their previous visit (planner.Placement; the round-4 headline)
  pipelined_with_predictor       never-searched batches, the next batch's placement predicted beside the current search
  secondary                      the other workloads (rand32, rand64; training budget; g_ratio 0.8) as single launches
  in_flight_through_api          parallel.InFlightPlanner: batches in flight behind the planner API
  throughput_regime              several batches in flight through the C ABI, general + unit-cost kernels, all workloads
  extra                          NeuralAstar encoders (CNN bf16 / f16x3, U-Net), encoder training step, search fwd + bwd, fused L1 step, data path
  reference_torch_on_this_gpu    the reference's own differentiable_astar.py run through PyTorch-ROCm on this GPU

`python bench.py --mode train ...` (BASELINE config 5: one full NeuralAstar training step per bench step) is `train_main` below.
Dev tools import the helpers from here (tools/probe_*.py)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "neural-astar_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench as core  # noqa: E402
from bench import (B_PER_GPU, FIXED_US, FLAG_UNIT_COST, G_RATIO, H, HBM_PEAK_GBS, LONE_STEP_NS, N_ROTATE, PLACEMENT, PREWARM_S, REF_STAGED, W, FreshBatches,  # noqa: E402,F401
                   Runner, _device_distances, _log, kernel_launch_ms, make_problem, oracle_check, pipe_model, prewarm, timed_loop)


def fresh_batches_pipelined(run, steps, warmup, dev):
    """Batches that have NEVER been searched, in a pipeline: while batch i is searched, a side stream computes the placement of batch
    i + 1 from its maps alone (nastar_placement_predict: length of the shortest route by a bit-parallel wave + counting sort; 17-44 us of
    small launches that fit into the search launch's idle tail).  Nothing measured on an earlier visit of a batch is used.  Returns
    seconds for `steps` steps, or None when the map size has no predictor."""
    lib = run.lib
    if run.H != run.W or run.W not in (32, 64):
        return None
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    nset = len(run.sets)
    orders = [torch.empty((run.B,), dtype=torch.int32, device=dev) for _ in range(nset)]
    wss = [torch.empty((run.B,), dtype=torch.int32, device=dev) for _ in range(nset)]
    ready = [torch.cuda.Event() for _ in range(nset)]
    done = [torch.cuda.Event() for _ in range(nset)]

    def predict(k):
        z = run.sets[k]
        side.wait_event(done[k])  # the previous search of this set no longer reads its order buffer
        rc = lib.nastar_placement_predict(z["m"].data_ptr(), z["s"].data_ptr(), z["g"].data_ptr(), run.B, run.H, run.W, orders[k].data_ptr(),
                                          wss[k].data_ptr(), run.B * 4, side.cuda_stream)
        run._check(rc, "nastar_placement_predict")
        ready[k].record(side)

    def search(k):
        z = run.sets[k]
        main.wait_event(ready[k])
        rc = lib.nastar_forward_ordered(z["m"].data_ptr(), z["s"].data_ptr(), z["g"].data_ptr(), z["m"].data_ptr(), run.B, run.H, run.W,
                                        run.g_ratio, run.max_iters, z["hist"].data_ptr(), z["paths"].data_ptr(), None, z["iters"].data_ptr(),
                                        z["status"].data_ptr(), None, None, 0, run.flags, orders[k].data_ptr(), None, main.cuda_stream)
        run._check(rc, "nastar_forward_ordered")
        done[k].record(main)

    for k in range(nset):
        done[k].record(main)
    predict(0)
    i = 0
    for phase, n in (("warm", warmup), ("timed", steps)):
        if phase == "timed":
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
        for _ in range(n):
            k = i % nset
            predict((i + 1) % nset)  # ... of the NEXT batch, beside this batch's search
            search(k)
            i += 1
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0


def multi_stream_throughput(pr, steps, dev, nstreams, flags=None, runs=None):
    runs = runs if runs is not None else [Runner(pr, dev, flags=flags, placement="natural") for _ in range(nstreams)]
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    for i in range(2 * nstreams):
        with torch.cuda.stream(streams[i % nstreams]):
            runs[i % nstreams].step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % nstreams]):
            runs[i % nstreams].step()
    torch.cuda.synchronize(dev)
    return runs[0].B * steps / (time.perf_counter() - t0)


def throughput_regime(dev, steps, workloads=("maze32", "rand32", "rand64"), ks=(2, 3, 4, 6, 8, 12)):
    """What the search sustains when the GPU always has a next batch (a planning service, an evaluation sweep, config 4's 32768-map
    batch): (a) the bench's 4096-map launches issued round-robin over k HIP streams, each stream with its own input AND output buffers
    -- the tail of one batch (its longest search) overlaps the bulk of the next; (b) ONE launch over 32768 maps (8 distinct-memory copies
    of the batch) on one stream: there `hbm_frac` is the roofline fraction in the bench line's own definition (algorithmic bytes of the
    launch / the launch's duration).  Per workload for the general kernel and for the unit-cost LDS layout (NASTAR_FLAG_UNIT_COST:
    cost and passable are one binary tensor, i.e. VanillaAstar; 29 instead of 16 resident 32x32 maps per CU).  Not the headline."""
    from neural_astar.utils import synthetic as syn
    out = []
    for w in workloads:
        pr = make_problem(w, B_PER_GPU, seed=1234)
        for label, flags in (("general", 0), ("unit_cost", FLAG_UNIT_COST)):
            # natural order: a placement is for ONE batch on an otherwise empty chip (latency); with batches in flight it front-loads every
            # launch's long searches and starves the HBM-bound short ones of overlap (rand32: 151 instead of 188 M maps/s)
            runs = [Runner(pr, dev, flags=flags, placement="natural") for _ in range(max(ks))]
            prewarm(runs[0], dev, 0.1)
            nbytes = 24 * runs[0].H * runs[0].W  # cost == passable (one tensor): the bytes that move (28 B/cell figure = x 7/6)
            sweep = {str(k): multi_stream_throughput(pr, steps, dev, k, runs=runs[:k]) for k in ks}
            best_k = max(sweep, key=sweep.get)
            ok = all(int(r.status.abs().sum().item()) == 0 for r in runs)
            del runs
            big = syn.Problems(*(np.concatenate([x] * 8) for x in pr))
            rb = Runner(big, dev, flags=flags)  # (a multi-round launch: order_out ranks the step counts, PLACEMENT applies)
            nbig = max(10, steps // 8)

            def big_ms():
                for _ in range(3):
                    rb.step()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(nbig):
                    rb.step()
                torch.cuda.synchronize(dev)
                return (time.perf_counter() - t0) / nbig * 1e3
            ms_big = big_ms()
            ms_big_nat = None
            if rb.placement == "hinted":
                rb.placement = "natural"
                ms_big_nat = big_ms()
            ok = ok and int(rb.status.abs().sum().item()) == 0
            del rb, big
            out.append({"workload": f"{w}: {B_PER_GPU} maps per launch", "kernel": label,
                        "streams_sweep_maps_per_s": sweep, "best_streams": int(best_k), "maps_per_s": sweep[best_k],
                        "hbm_frac": sweep[best_k] * nbytes / 1e9 / HBM_PEAK_GBS, "hbm_frac_bytes_per_cell": 24,
                        "one_launch_32768_maps": {"ms": ms_big, "maps_per_s": 8 * B_PER_GPU / (ms_big * 1e-3),
                                                  "hbm_frac": 8 * B_PER_GPU * nbytes / (ms_big * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                  "placement": PLACEMENT + (" (maps sorted by the step counts of the previous visit, longest first)" if ms_big_nat else ""),
                                                  "natural_order_ms": ms_big_nat,
                                                  "natural_order_hbm_frac": (8 * B_PER_GPU * nbytes / (ms_big_nat * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_big_nat else None},
                        "all_status_ok": ok})
        del pr
    return out


# ---- what a launch costs the CU's pipes ("report expansions/s against an issue model") ------------------------------------
# Instruction classes of ONE step of the shipped 32x32 stream (nastar_search_asm4.hip.h, g_ratio 0.5 form; counted in the disassembly)
# x the aggregate rates of one CU measured by tools/ubench/rate.hip (profiles/r04/rate.txt; cycles per wavefront instruction with
# >= 2 wavefronts per SIMD / >= 16 per CU).  Round 3 priced every VALU instruction at 4 cycles; the measured machine issues a plain one
# every 2.3 cycles per SIMD and only DPP / compare / lane-read forms at ~4, and the LDS pipe of the CU takes 2.45 (read) / 4.6 (write) /
# 6.0 (64-bit atomic) cycles per instruction WHATEVER the number of active lanes.
def two_stream_throughput(pr, steps, dev):
    """Extra (not the headline): the same steps issued round-robin on TWO HIP streams with their own output buffers,
    so the serial tail of one batch (its longest search) overlaps the bulk of the next -- the throughput a planning
    service that always has a next batch would see.  Per-launch latency gets worse, aggregate maps/s better."""
    runs = [Runner(pr, dev, placement="natural"), Runner(pr, dev, placement="natural")]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for i in range(4):
        with torch.cuda.stream(streams[i & 1]):
            runs[i & 1].step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i & 1]):
            runs[i & 1].step()
    torch.cuda.synchronize(dev)
    return runs[0].B * steps / (time.perf_counter() - t0)


def training_step_ms(pr, dev, reps=10):
    """Extra: forward + straight-through backward (nastar_backward_replay) of one 4096-map batch with U(0,1) costs in
    training mode, Tmax = 0.25 (the reference's scripts/config/train.yaml), through the torch custom ops."""
    from neural_astar import ops
    from neural_astar.utils import synthetic as syn
    m = torch.from_numpy(pr.map_designs[:, 0]).to(dev)
    s = torch.from_numpy(pr.start_maps[:, 0]).to(dev)
    g = torch.from_numpy(pr.goal_maps[:, 0]).to(dev)
    cost = torch.from_numpy(syn.random_costs(m.shape[0], H, W, seed=3)[:, 0]).to(dev)
    mi = int(0.25 * W * W)
    hist, _, iters, _, _ = torch.ops.nastar.astar_forward(cost, s, g, m, G_RATIO, mi, False)
    gh = torch.randn_like(hist)
    tb = (iters.amax() - 1).to(torch.int32).reshape(1)

    def replay():  # the forward logs its selections, the backward replays them (nastar_backward_replay)
        h, _, it, _, log = torch.ops.nastar.astar_forward(cost, s, g, m, G_RATIO, mi, True)
        torch.ops.nastar.astar_backward_replay(gh, cost, s, g, m, log, G_RATIO, mi, it, tb)
    out = {}
    variants = [("replay_ms", replay)]
    for name, once in variants:
        for _ in range(2):
            once()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            once()
        e1.record()
        torch.cuda.synchronize(dev)
        out[name] = e0.elapsed_time(e1) / reps
    # the replay backward on its own: one launch at a time and with several batches in flight, against the HBM roofline on
    # its algorithmic bytes -- reads cost + start + goal + passable + upstream gradient (5 x 4 B per cell) and the executed part of the
    # selection log (4 B per step), writes dL/dcost (4 B per cell); placed by the forward's completion order as the training paths do
    from neural_astar import _native
    lib = _native.load()
    B = m.shape[0]
    ord_buf = ops.new_placement_buffer(B, dev)
    h, _, it, _, log = torch.ops.nastar.astar_forward_ordered(cost, s, g, m, G_RATIO, mi, True, 0, None, ord_buf, False)
    order = ord_buf[:B].contiguous()
    torch.cuda.synchronize(dev)
    alg_bytes = B * H * W * 24 + int(it.sum().item()) * 4
    ws_bytes = int(lib.nastar_backward_workspace_bytes(B, H, W, mi))

    def make_set():
        return dict(gc=torch.empty_like(cost), ws=torch.empty((ws_bytes,), dtype=torch.uint8, device=dev))

    def launch(z, stream):
        rc = lib.nastar_backward_replay_ordered(gh.data_ptr(), None, None, None, cost.data_ptr(), s.data_ptr(), g.data_ptr(), m.data_ptr(), log.data_ptr(), B,
                                                H, W, G_RATIO, mi, it.data_ptr(), tb.data_ptr(), z["gc"].data_ptr(), z["ws"].data_ptr(), ws_bytes, 0,
                                                order.data_ptr(), stream.cuda_stream)
        _native.check(rc, "nastar_backward_replay_ordered")
    res = {}
    for k in (1, 2, 3, 4):
        streams = [torch.cuda.Stream(dev) for _ in range(k)]
        sets = [make_set() for _ in range(k)]
        n = 24 * k
        for i in range(2 * k):
            launch(sets[i % k], streams[i % k])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(n):
            launch(sets[i % k], streams[i % k])
        torch.cuda.synchronize(dev)
        dtk = (time.perf_counter() - t0) / n
        res[str(k)] = {"ms_per_batch": dtk * 1e3, "maps_per_s": B / dtk, "hbm_frac": alg_bytes / dtk / 1e9 / HBM_PEAK_GBS}
    out["backward_replay_alone"] = {"streams": res, "algorithmic_bytes_per_batch": alg_bytes, "mean_steps_per_map": float(it.float().mean().item()),
                                    "note": "nastar_backward_replay_ordered (hand-scheduled 32x32 loop, 16 B cell records + 16 B of history per step in LDS: 8 maps per "
                                            "CU), k batches in flight on k streams; a map's replay is a serial chain exactly as long as its search was"}
    return out


def l1_training_step_ms(pr, dev, batch, reps=20):
    """Extra: the reference's training step on `batch` maps (utils/training.py:55-61, Tmax = 0.25, cost = leaf tensor):
    L1Loss through autograd vs the fused node (nastar_l1_loss + nastar_backward_l1_replay)."""
    from neural_astar import ops
    from neural_astar.utils import synthetic as syn
    m = torch.from_numpy(pr.map_designs[:batch, 0]).to(dev).contiguous()
    s = torch.from_numpy(pr.start_maps[:batch, 0]).to(dev).contiguous()
    g = torch.from_numpy(pr.goal_maps[:batch, 0]).to(dev).contiguous()
    traj = ((torch.rand_like(m) < 0.2).float() * m).contiguous()
    cost = torch.from_numpy(syn.random_costs(batch, H, W, seed=3)[:, 0]).to(dev).requires_grad_(True)
    mi = int(0.25 * W * W)
    l1 = torch.nn.L1Loss()

    def unfused():
        cost.grad = None
        hist, _, _, _, _ = torch.ops.nastar.astar_forward(cost, s, g, m, G_RATIO, mi, True)  # the selection log is the backward's tape
        l1(hist, traj).backward()

    def fused():
        cost.grad = None
        ops.astar_l1_loss(cost, s, g, m, traj, G_RATIO, mi)[0].backward()
    out = {}
    for name, fn in (("autograd_l1loss_ms", unfused), ("fused_ms", fused)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        out[name] = e0.elapsed_time(e1) / reps
    return out


def data_path_ms(dev, n_maps=400, batch=100):
    """Extra: time to produce one collated training batch of `batch` maze problems (start sampling + optimal-trajectory roll-out):
    the reference-style per-sample host loop (DataLoader over MazeDataset.__getitem__) vs the device-resident loader."""
    import tempfile
    from neural_astar.utils import synthetic as syn
    from neural_astar.utils.data import create_dataloader, create_device_loader
    path = os.path.join(tempfile.mkdtemp(), "mazes.npz")
    syn.write_maze_npz(path, n_train=n_maps, n_valid=1, n_test=1, size=32, seed=11)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        host = create_dataloader(path, "train", batch, shuffle=True)
        devl = create_device_loader(path, "train", batch, dev, shuffle=True)
    t0 = time.perf_counter()
    n = 0
    for b in host:
        b = [x.to(dev, non_blocking=True) for x in b]
        n += 1
    torch.cuda.synchronize(dev)
    host_ms = (time.perf_counter() - t0) * 1e3 / n
    for _ in devl:
        pass
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    n = 0
    for _ in range(5):
        for b in devl:
            n += 1
    torch.cuda.synchronize(dev)
    dev_ms = (time.perf_counter() - t0) * 1e3 / n
    return {"host_dataloader_ms_per_batch": host_ms, "device_loader_ms_per_batch": dev_ms, "batch": batch, "maps": n_maps}


def neural_astar_f16x3_ms(pr, dev, reps=5):
    """Extra: the same NeuralAstar forward with the fp32-grade encoder (encoder_backend="hip_f16x3": cost maps within 1e-5 of the
    fp32 reference encoder, the north-star tolerance for float outputs)."""
    from neural_astar.planner import NeuralAstar
    torch.manual_seed(0)
    na = NeuralAstar(encoder_arch="CNN").to(dev).eval()
    na.encoder_backend = "hip_f16x3"
    na.astar.check_solvable = False
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    with torch.no_grad():
        for _ in range(2):
            na(m, s, g)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            c = na.encode(m, s, g)
        e1.record()
        torch.cuda.synchronize(dev)
        enc_ms = e0.elapsed_time(e1) / reps
        e0.record()
        for _ in range(reps):
            na(m, s, g)
        e1.record()
        torch.cuda.synchronize(dev)
        full_ms = e0.elapsed_time(e1) / reps
        na.encoder_backend = "torch"   # fp32 torch encoder on a 16-map slice only (MIOpen autotunes per shape; keep it short)
        ref = na.encode(m[:16], s[:16], g[:16])
        err = float((c[:16] - ref).abs().max())
    flop = 3 * 2.0 * m.shape[0] * H * W * 9 * (32 * 64 + 64 * 128 + 128 * 256 + 256)
    return {"encoder_ms": enc_ms, "forward_ms": full_ms, "maps_per_s": m.shape[0] / full_ms * 1e3,
            "max_abs_diff_vs_torch_fp32_encoder_16_maps": err, "mfma_tflops_incl_split_products": flop / enc_ms / 1e9,
            "dtype": "fp16 hi/lo split operands (3 products) / fp32 accumulate"}


def neural_astar_forward_ms(pr, dev, reps=10):
    """Extra (BASELINE config 3 stand-in): NeuralAstar(CNN encoder, depth 4) forward on the bench batch with the bf16-MFMA
    HIP encoder + the HIP search, eval mode.  (The torch/MIOpen encoder is not timed here: its first call autotunes for
    minutes; DESIGN.md quotes it from tools/probe_encoder.py.)"""
    from neural_astar.planner import NeuralAstar
    torch.manual_seed(0)
    na = NeuralAstar(encoder_arch="CNN").to(dev).eval()
    na.encoder_backend = "hip_bf16"
    na.astar.check_solvable = False
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    with torch.no_grad():
        for _ in range(2):
            na(m, s, g)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            na.encode(m, s, g)
        e1.record()
        torch.cuda.synchronize(dev)
        enc_ms = e0.elapsed_time(e1) / reps
        e0.record()
        for _ in range(reps):
            na(m, s, g)
        e1.record()
        torch.cuda.synchronize(dev)
        full_ms = e0.elapsed_time(e1) / reps
    flop = 2.0 * m.shape[0] * H * W * 9 * (2 * 32 + 32 * 64 + 64 * 128 + 128 * 256 + 256)
    tf = flop / enc_ms / 1e9
    return {"encoder_ms": enc_ms, "encoder_useful_tflops": tf, "forward_ms": full_ms,
            "maps_per_s": m.shape[0] / full_ms * 1e3, "dtype": "bf16 operands / fp32 accumulate (encoder), f32 (search)",
            "encoder_roofline": {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0,
                                 "note": "useful FLOPs of the 5 conv layers / wall time of the whole encoder; the matrix pipe itself "
                                         "sustains 1660 TFLOP/s on random bf16 operands at the power limit (tools/ubench/mfma_peak.hip)"}}


def neural_astar_unet_ms(pr, dev, precision, reps=5):
    """Extra (BASELINE config 3 literally: NeuralAstar UNet encoder + diff-A* on 32x32 mazes, fp16, batch 4096): Unet(vgg16_bn)
    through the generic fp16 MFMA convolution (csrc/nastar_conv_flat.hip.h) + the HIP search, eval mode, random-init weights with
    calibrated BatchNorm statistics.  ``precision``: "f16" (plain fp16 operands) or "f16x3" (split operands, fp32-grade)."""
    from neural_astar.planner import NeuralAstar
    torch.manual_seed(0)
    na = NeuralAstar(encoder_arch="Unet", encoder_depth=4).to(dev)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    with torch.no_grad():  # BatchNorm running statistics = batch statistics of 64 bench maps (activations stay O(1) through 26 layers)
        for mod in na.encoder.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.momentum = 1.0
        na.train()
        na.encoder(torch.cat((m[:64], s[:64] + g[:64]), dim=1))
    na.eval()
    na.astar.check_solvable = False
    with torch.no_grad():
        ref = na.encode(m[:64], s[:64], g[:64])
        na.encoder_backend = "hip_" + precision
        err = float((na.encode(m[:64], s[:64], g[:64]) - ref).abs().max())
        for _ in range(2):
            na(m, s, g)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            na.encode(m, s, g)
        e1.record()
        torch.cuda.synchronize(dev)
        enc_ms = e0.elapsed_time(e1) / reps
        e0.record()
        for _ in range(reps):
            na(m, s, g)
        e1.record()
        torch.cuda.synchronize(dev)
        full_ms = e0.elapsed_time(e1) / reps
    tf = na._hip_encoder.flops(H, W) * m.shape[0] / enc_ms / 1e9
    products = 3 if precision == "f16x3" else 1
    return {"encoder_ms": enc_ms, "encoder_useful_tflops": tf, "forward_ms": full_ms, "maps_per_s": m.shape[0] / full_ms * 1e3,
            "max_abs_diff_vs_torch_fp32_encoder_64_maps": err,
            "dtype": ("fp16 hi/lo split operands (3 products)" if products == 3 else "fp16 operands") + " / fp32 accumulate (encoder), f32 (search)",
            "encoder_roofline": {"bound": "mfma", "achieved": tf * products, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf * products / 2500.0,
                                 "note": "FLOPs of the 24 conv layers (real channel counts" + (", x3 split products" if products == 3 else "")
                                         + ") / wall time of the whole encoder incl. pooling and input assembly launches"}}


def encoder_train_step_ms(pr, dev):
    """Extra (training): forward + backward of the CNN encoder alone (loss = sum(cost * R)) through the MI355X training
    kernels (neural_astar/encoder_train.py: fp16-MFMA convolutions, input and weight gradients, batch-statistics BatchNorm), at the
    reference's training batch (100 maps) and at the bench batch.  The fp32 torch.nn encoder on the same box: 4.07 / 132.8 ms per
    100 / 4096 maps (profiles/r02/encoder_train_step_ms.json; not re-timed here, MIOpen's autotuning takes minutes)."""
    from neural_astar.planner import NeuralAstar
    out = {}
    for B in (100, 4096):
        m, s, g = (torch.from_numpy(x[:B]).to(dev) for x in pr)
        R = torch.randn((B, 1, H, W), device=dev) / (B * H * W)
        for backend in ("hip_f16x3", "hip_f16"):
            torch.manual_seed(0)
            na = NeuralAstar(encoder_arch="CNN").to(dev).train()
            na.encoder_backend = backend

            def one():
                for p in na.parameters():
                    p.grad = None
                (na.encode(m, s, g) * R).sum().backward()
            one()
            torch.cuda.synchronize(dev)
            reps = 10 if B == 100 else 3
            t0 = time.perf_counter()
            for _ in range(reps):
                one()
            torch.cuda.synchronize(dev)
            out[f"batch_{B}_{backend}_ms"] = (time.perf_counter() - t0) / reps * 1e3
            del na
    # BASELINE config 5's encoder: CNNDownSize, rgb+, depth 3, 96x96 RGB -> 12x12, the reference's batch of 100 (train_warcraft.yaml)
    img = torch.rand((100, 3, 96, 96), device=dev)
    s12 = torch.zeros((100, 1, 12, 12), device=dev)
    g12 = torch.zeros((100, 1, 12, 12), device=dev)
    s12[:, 0, 0, 0] = 1
    g12[:, 0, 11, 11] = 1
    R12 = torch.randn((100, 1, 12, 12), device=dev) / 14400
    for backend in ("hip_f16x3", "hip_f16"):
        torch.manual_seed(0)
        na = NeuralAstar(encoder_input="rgb+", encoder_arch="CNNDownSize", encoder_depth=3, const=10.0).to(dev).train()
        na.encoder_backend = backend

        def one_w():
            for p in na.parameters():
                p.grad = None
            (na.encode(img, s12, g12) * R12).sum().backward()
        one_w()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(10):
            one_w()
        torch.cuda.synchronize(dev)
        out[f"warcraft_batch_100_{backend}_ms"] = (time.perf_counter() - t0) / 10 * 1e3
        del na
    # Unet(vgg16_bn) (BASELINE config 3's encoder), 100 maps: ~330 launches per step, launch-bound at this batch
    m, s, g = (torch.from_numpy(x[:100]).to(dev) for x in pr)
    R = torch.randn((100, 1, H, W), device=dev) / (100 * H * W)
    for backend in ("hip_f16x3", "hip_f16"):
        torch.manual_seed(0)
        na = NeuralAstar(encoder_arch="Unet", encoder_depth=4).to(dev).train()
        na.encoder_backend = backend

        def one_u():
            for p in na.parameters():
                p.grad = None
            (na.encode(m, s, g) * R).sum().backward()
        one_u()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(5):
            one_u()
        torch.cuda.synchronize(dev)
        out[f"unet_batch_100_{backend}_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        del na
    out["torch_fp32_ms_same_box"] = {"batch_100": 4.07, "batch_4096": 132.75, "warcraft_batch_100": 3.13, "unet_batch_100": 6.65,
                                     "source": "profiles/r02/encoder_train_step_ms.json, encoder_train_step_warcraft_b100_ms.json, "
                                               "encoder_train_step_unet_b100_ms.json"}
    out["unit"] = "ms per encoder forward+backward (wall clock), random-init CNN depth 4, 32x32 maps"
    return out


def reference_on_this_gpu(pr, gpu_hist, gpu_paths, dev):
    """Extra: the REAL reference DifferentiableAstar.forward (the staged torch-only module, oracle/_ref/) run through PyTorch-ROCm on
    the SAME MI355X -- what a user of the reference gets on this hardware without this package (~45 ATen launches + one device->host
    sync per loop iteration).  One warm-up call on 256 maps, then the whole bench batch once; masks compared with the HIP kernel's."""
    import importlib.util
    if not os.path.exists(REF_STAGED):
        return {"available": False, "note": "oracle/_ref/differentiable_astar.py not staged (built outside the authoring container)"}
    spec = importlib.util.spec_from_file_location("ref_differentiable_astar_gpu", REF_STAGED)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    planner = ref.DifferentiableAstar(g_ratio=G_RATIO, Tmax=1.0).to(dev).eval()
    m, s_, g = (torch.from_numpy(x).to(dev) for x in pr)
    with torch.no_grad():
        planner(m[:256], s_[:256], g[:256], m[:256])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out = planner(m, s_, g, m)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
    ok = bool(np.array_equal(out.histories[:, 0].cpu().numpy(), gpu_hist) and np.array_equal(out.paths[:, 0].cpu().numpy(), gpu_paths))
    return {"available": True, "value": m.shape[0] / dt, "unit": "maps/s", "seconds_per_batch": dt, "batch": int(m.shape[0]),
            "torch": torch.__version__, "masks_equal_to_hip_kernel": ok,
            "note": "reference differentiable_astar.py on the same GPU via PyTorch-ROCm, eval mode, no_grad"}


def through_module_ms(pr, dev, reps=60):
    """End to end through the drop-in boundary (north_star: "keeps the forward() API"): ms per VanillaAstar.forward() call on
    the bench batch, wall clock -- output allocation, the launch, and the solvability policy: the default (True = "sync") waits for the
    kernel and raises in the same call (one stream wait + one 64-byte read of the pinned status summary the launch wrote), "deferred"
    (opt-in) hands the verdict to a later call (an event, no host wait), False skips it.  The batch carries the placement its loader
    attached (start_maps.placement_order, by the optimal distance of the start cells); `no_placement_*` = the same calls without it."""
    from neural_astar import ops
    from neural_astar.planner import VanillaAstar
    m, s_, g = (torch.from_numpy(x).to(dev) for x in (pr.map_designs, pr.start_maps, pr.goal_maps))
    dist = _device_distances(m[:, 0], g[:, 0]).reshape(m.shape[0], -1)
    levels = (dist * (s_.reshape(m.shape[0], -1) > 0)).sum(1).to(torch.int32).contiguous()
    out = {}
    for hinted in (True, False):
        if hinted:
            ops.attach_order(s_, levels)
        elif hasattr(s_, "placement_order"):
            del s_.placement_order
        for label, chk in (("check_solvable_default_sync", True), ("check_solvable_deferred", "deferred"), ("check_solvable_false", False)):
            va = VanillaAstar().to(dev).eval()
            va.astar.check_solvable = chk
            with torch.no_grad():
                for _ in range(5):
                    va(m, s_, g)
                torch.cuda.synchronize(dev)
                chunks = []  # three chunks, the fastest reported: a fresh process occasionally stalls ~50 ms once (NOTES.md round 4; seen again in round 6)
                for _c in range(3):
                    t0 = time.perf_counter()
                    for _ in range(max(reps // 3, 1)):
                        va(m, s_, g)
                    torch.cuda.synchronize(dev)
                    chunks.append((time.perf_counter() - t0) / max(reps // 3, 1) * 1e3)
                    va.astar.raise_if_unsolvable()
            out[("" if hinted else "no_placement_") + label] = min(chunks)
    out["unit"] = "ms per VanillaAstar.forward() call, wall clock, same input batch each call"
    out["note"] = ("all three modes run the general kernel (forward() takes the unit-cost layout only with unit_cost=True); the default waits for "
                   "the launch's completion flag in the same call, deferred / false issue the launches back to back without a host wait, i.e. "
                   "run at the kernel's own duration")
    return out


def in_flight_through_api(dev, steps, workloads=(("maze32", True), ("rand32", True), ("rand64", True), ("maze32", False)), ks=(3, 4, 6, 8, 12)):
    """Batches in flight THROUGH THE PYTHON OBJECT (neural_astar.parallel.InFlightPlanner around a VanillaAstar): whole 4096-map batches
    round-robin over k HIP streams, outputs allocated per batch, status summaries read once at collection, unit_cost="auto" without a
    per-call wait.  maps/s over `n` batches incl. submission, collection and the final host wait; outputs checked equal to sequential
    planner.forward() calls on the first batches."""
    from neural_astar.parallel import InFlightPlanner
    from neural_astar.planner import VanillaAstar
    res = []
    n = max(144, min(steps, 288))  # (not the K timed steps of the contract: 20 batches are a pipeline that never fills)
    for w, unit in workloads:
        prs = [make_problem(w, B_PER_GPU, seed=1234 + 1000 * k) for k in range(N_ROTATE)]
        batches = [tuple(torch.from_numpy(x).to(dev) for x in (pr.map_designs, pr.start_maps, pr.goal_maps)) for pr in prs]
        va = VanillaAstar().to(dev).eval()
        with torch.no_grad():
            seq = [va(*b) for b in batches]
        sweep = {}
        same = True
        for k in ks:
            fly = InFlightPlanner(va, streams=k, unit_cost="auto" if unit else False)
            outs = fly.plan_many(batches[i % N_ROTATE] for i in range(n))  # warm-up (the allocator then holds n output sets) + equality with the sequential calls
            same = same and all(torch.equal(o.histories, seq[i % N_ROTATE].histories) and torch.equal(o.paths, seq[i % N_ROTATE].paths)
                                for i, o in enumerate(outs))
            del outs
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            outs = fly.plan_many(batches[i % N_ROTATE] for i in range(n))
            dt = time.perf_counter() - t0
            del outs
            sweep[str(k)] = n * B_PER_GPU / dt
        best = max(sweep, key=sweep.get)
        Hh, Ww = batches[0][0].shape[-2:]
        res.append({"workload": f"{w}: {B_PER_GPU} maps per batch, {n} batches", "kernel": "unit_cost (auto)" if unit else "general",
                    "streams_sweep_maps_per_s": sweep, "best_streams": int(best), "maps_per_s": sweep[best],
                    "hbm_frac": sweep[best] * 24 * Hh * Ww / 1e9 / HBM_PEAK_GBS, "equal_to_sequential_forward": bool(same)})
        del batches, seq
    return res


# ---- --mode train: BASELINE config 5 (and the maze configuration of scripts/train.py) as a driver-runnable training bench ---------------
TRAIN_CONFIGS = {
    # constructor arguments = the reference's scripts (scripts/train.py:33-39 + config/train.yaml, scripts/train_warcraft.py:33-40 +
    # config/train_warcraft.yaml); batch_size 100 per step in both
    "maze": dict(kw=dict(encoder_input="m+", encoder_arch="CNN", encoder_depth=4, Tmax=0.25), chans=[2, 32, 64, 128, 256, 1], pool=False, hw=(32, 32)),
    "warcraft": dict(kw=dict(encoder_input="rgb+", encoder_arch="CNNDownSize", encoder_depth=3, const=10.0, learn_obstacles=True, Tmax=0.25),
                     chans=[4, 32, 64, 128, 1], pool=True, hw=(96, 96)),
}


def train_batch(config, B, seed, dev):
    """synthetic training batch in the reference loaders' layout (utils/data.py): (map_designs, start_maps, goal_maps, opt_trajs)"""
    from neural_astar.planner import VanillaAstar
    from neural_astar.utils import synthetic as syn
    if config == "warcraft":
        g_ = torch.Generator().manual_seed(seed)
        tiles = torch.rand((B, 3, 12, 12), generator=g_)
        img = (tiles.repeat_interleave(8, 2).repeat_interleave(8, 3) + 0.08 * torch.randn((B, 3, 96, 96), generator=g_)).clamp_(0, 1).to(dev)
        s = torch.zeros((B, 1, 12, 12), device=dev)
        gl = torch.zeros((B, 1, 12, 12), device=dev)
        s[:, 0, 0, 0] = 1
        gl[:, 0, -1, -1] = 1
        true_cost = (0.1 + 0.9 * torch.rand((B, 1, 12, 12), generator=g_)).to(dev)
        with torch.no_grad():  # the label of a WarCraft sample is the shortest path under hidden per-tile costs
            traj = VanillaAstar().to(dev).eval().astar(true_cost, s, gl, torch.ones_like(s)).paths.float()
        return img, s, gl, traj
    pr = syn.maze_maps(B, 32, seed=seed)
    m, s, gl = (torch.from_numpy(x).to(dev) for x in pr)
    with torch.no_grad():
        traj = VanillaAstar().to(dev).eval()(m, s, gl).paths.float()
    return m, s, gl, traj


def train_flops_per_map(config):
    """useful convolution FLOPs of one training step per map: forward + input gradient (not for the first layer) + weight gradient"""
    c = TRAIN_CONFIGS[config]
    h, w = c["hw"]
    fwd, total = 0.0, 0.0
    for l, (ci, co) in enumerate(zip(c["chans"][:-1], c["chans"][1:])):
        f = 2.0 * 9 * h * w * ci * co
        fwd += f
        total += f * (2 if l == 0 else 3)
        if c["pool"] and l < len(c["chans"]) - 2:
            h, w = h // 2, w // 2
    return fwd, total


def train_cpu_baseline(config, B, budget_s=25.0):
    """The reference training step on the host cores: the reference's OWN DifferentiableAstar (staged oracle/_ref module, ~45 ATen ops
    per search iteration under autograd) behind this package's torch.nn encoder (tests/test_reference_modules_cpu.py pins it to the
    reference's encoder classes: identical cost maps and gradients), nn.L1Loss, RMSprop -- utils/training.py:55-61 as the reference
    runs it on a CPU.  A bounded sample: `B` maps, as many steps as fit the budget (at least 1)."""
    import importlib.util
    from neural_astar.planner import NeuralAstar
    path = os.path.join(ROOT, "oracle", "_ref", "differentiable_astar.py")
    if not os.path.exists(path):
        return {"available": False, "note": "oracle/_ref not staged (run __graft_entry__.build() where /root/reference exists)"}
    spec = importlib.util.spec_from_file_location("ref_da_train", path)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    cores = min(32, os.cpu_count() or 1)  # small per-iteration tensors: more threads only add fork/join overhead (as in cpu_baseline)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    na = NeuralAstar(**TRAIN_CONFIGS[config]["kw"]).train()
    astar = ref.DifferentiableAstar(g_ratio=0.5, Tmax=0.25).train()
    opt = torch.optim.RMSprop(na.parameters(), 1e-3)
    g_ = torch.Generator().manual_seed(5)
    if config == "warcraft":
        m = torch.rand((B, 3, 96, 96), generator=g_)
        s = torch.zeros((B, 1, 12, 12)); gl = torch.zeros((B, 1, 12, 12))
        s[:, 0, 0, 0] = 1; gl[:, 0, -1, -1] = 1
        traj = torch.zeros((B, 1, 12, 12)); traj[:, 0, torch.arange(12), torch.arange(12)] = 1
        passable = torch.ones_like(s)
    else:
        from neural_astar.utils import synthetic as syn
        pr = syn.maze_maps(B, 32, seed=9)
        m, s, gl = (torch.from_numpy(x) for x in pr)
        traj = (torch.rand(m.shape, generator=g_) < 0.1).float() * m
        passable = m

    def step():
        opt.zero_grad(set_to_none=True)
        out = astar(na.encode(m, s, gl), s, gl, passable)
        torch.nn.L1Loss()(out.histories, traj).backward()
        opt.step()
    step()
    n, t0 = 0, time.perf_counter()
    while n < 1 or (time.perf_counter() - t0 < budget_s and n < 20):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": B / dt, "unit": "maps/s", "cores": cores, "kind": "reference",
            "sample": f"{n} training step(s) of {B} maps after 1 warm-up: the reference's DifferentiableAstar under autograd (staged module) + "
                      f"torch.nn encoder + L1Loss + RMSprop on {cores} host threads", "ms_per_step": dt * 1e3}


def train_main(args, real_stdout):
    """`bench.py --mode train --config maze|warcraft [--gpus N]`: the reference's training step (planner forward, L1 loss on histories,
    straight-through backward, RMSprop) with encoder AND search on the MI355X kernels; N > 1 = DataParallelTrainer over RCCL (each
    rank its own `--batch-per-gpu` maps: weak scaling; BatchNorm statistics of the global batch, one flat gradient all-reduce)."""
    from neural_astar.planner import NeuralAstar
    from neural_astar.utils import distributed as D
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1 or args.force_collate:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    n_gpus = max(world, 1)
    B = args.batch_per_gpu
    cfg = TRAIN_CONFIGS[args.config]
    torch.manual_seed(1234)
    planner = NeuralAstar(**cfg["kw"]).to(dev)
    planner.astar.check_solvable = "deferred"  # opt-in: no host sync inside the timed steps; the verdicts are collected after the loop
    planner.encoder_backend = args.encoder_backend  # "auto" (the package default) resolves to hip_f16x3 on a HIP device
    if args.encoder_backend == "auto":
        args.encoder_backend = planner.effective_encoder_backend(torch.empty(0, device=dev))
    multi = dist.is_initialized() and (world > 1 or args.force_collate)  # --force-collate: the RCCL path in a 1-rank group
    sync_bn = multi and args.encoder_backend.startswith(("hip", "auto"))
    trainer = D.DataParallelTrainer(planner, lr=1e-3, coupling="global" if multi else "local", sync_bn=sync_bn,
                                    force_collectives=args.force_collate)
    batches = [train_batch(args.config, B, 1234 + 17 * rank + 1000 * k, dev) for k in range(4)]  # a few distinct batches in rotation
    _log(f"train mode: {args.config}, {B} maps/GPU, encoder_backend={args.encoder_backend}, world={world}, sync_bn={sync_bn}")

    def run(n):
        last = None
        for i in range(n):
            last = trainer.train_step(*batches[i % len(batches)])
        return last
    t_pre = time.perf_counter()  # untimed: clocks out of their idle state (see PREWARM_S)
    while time.perf_counter() - t_pre < PREWARM_S:
        run(2)
        torch.cuda.synchronize(dev)
    import gc
    gc.collect()  # (before the warm-up steps; the cyclic collector then stays out of the timed region, as in timed_loop)
    gc.disable()
    run(args.warmup)
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    loss = run(args.steps)
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    gc.enable()
    planner.astar.raise_if_unsolvable()  # the deferred verdicts of every step above
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the same step WITHOUT sync BatchNorm (every rank normalises with its own rows: no per-layer collectives, only the flat gradient
    # all-reduce) so that a scaling curve can separate the number of small collectives from wire time
    dt_nosync = None
    if multi and sync_bn:
        trainer_ns = D.DataParallelTrainer(planner, lr=1e-3, coupling="global", sync_bn=False, force_collectives=args.force_collate)

        def run_ns(n):
            for i in range(n):
                trainer_ns.train_step(*batches[i % len(batches)])
        run_ns(max(2, args.warmup))
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        run_ns(args.steps)
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_nosync = float(t.item())
        planner.astar.raise_if_unsolvable()
    if rank == 0:
        fwd_f, step_f = train_flops_per_map(args.config)
        ms = dt / args.steps * 1e3
        split = args.encoder_backend == "hip_f16x3"
        out = {
            "metric": f"map-instances/s (NeuralAstar TRAINING step, {args.config} configuration, Tmax 0.25, batch {B}/GPU)",
            "value": n_gpus * B * args.steps / dt, "unit": "maps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"hip_f16x3": "f16x3 (split fp16 operands, fp32 accumulation: fp32-grade)", "hip_f16": "f16 (fp32 accumulation)",
                      "torch": "f32"}[args.encoder_backend] + " encoder, f32 search",
            "data": "synthetic",
            "config": {"workload": f"train/{args.config}: NeuralAstar({', '.join(f'{k}={v}' for k, v in cfg['kw'].items())}), {B} maps/GPU per step, "
                                   f"encoder forward+backward on {'the MI355X training kernels' if args.encoder_backend.startswith('hip') else 'torch.nn (MIOpen)'}, "
                                   "HIP search forward + replay backward, fused L1 loss, RMSprop(lr 1e-3); random-init weights",
                       "batch_per_gpu": B, "global_batch": B * n_gpus, "encoder_backend": args.encoder_backend,
                       "parallelism": (f"dp{n_gpus}: flat fp32 gradient all-reduce ({'RCCL' if args.dist_backend == 'nccl' else args.dist_backend}) + all-reduced BatchNorm sums (sync_bn={sync_bn}), "
                                       "coupling=global") if multi else "single"},
            "steps_per_s": args.steps / dt, "final_loss": float(loss),
            "sync_bn": {"on_ms_per_step": ms if sync_bn else None, "off_ms_per_step": (dt_nosync / args.steps * 1e3) if dt_nosync else (None if sync_bn else ms),
                        "note": "on = BatchNorm statistics of the GLOBAL batch (one small all-reduce per BatchNorm layer and direction: "
                                "the single-device step on the concatenated batch); off = per-rank statistics, only the flat gradient "
                                "all-reduce; `value` / `ms_per_step` are the sync_bn=on figures when n_gpus > 1"},
            "roofline": {"bound": "mfma", "achieved": B * step_f / (ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": B * step_f / (ms * 1e-3) / 1e12 / 2500.0, "traffic": None,
                         "kernel": "WHOLE STEP, not one kernel: useful convolution FLOPs (forward + input gradient + weight gradient, "
                                   "no split-operand products counted) / wall time of the step; at 100 maps the step is launch-bound",
                         "useful_flops_per_map": step_f, "forward_flops_per_map": fwd_f,
                         "matrix_products_issued_x": 3 if split else 1},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            _log("train: same step with the torch.nn encoder on this GPU")
            try:
                torch.manual_seed(1234)
                p2 = NeuralAstar(**cfg["kw"]).to(dev)
                t2 = D.DataParallelTrainer(p2, lr=1e-3, coupling="local")
                for i in range(3):
                    t2.train_step(*batches[i % len(batches)])
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for i in range(max(5, args.steps // 4)):
                    t2.train_step(*batches[i % len(batches)])
                torch.cuda.synchronize(dev)
                out["torch_encoder_on_this_gpu"] = {"ms_per_step": (time.perf_counter() - t0) / max(5, args.steps // 4) * 1e3,
                                                    "note": "HIP search kernels + torch.nn (MIOpen fp32) encoder"}
            except Exception as e:  # noqa: BLE001
                out["torch_encoder_on_this_gpu"] = {"available": False, "note": f"{type(e).__name__}: {e}"}
            _log("train: cpu baseline")
            try:
                out["cpu_baseline"] = train_cpu_baseline(args.config, min(B, 100))
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"available": False, "note": f"{type(e).__name__}: {e}"}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()




def _emit(obj: dict) -> None:
    """one finished section -> one JSON line on stdout (the parent keeps whatever arrived before a time limit)"""
    sys.stdout.write(json.dumps(obj) + "\n")
    sys.stdout.flush()


def extras_main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="maze32", choices=["maze32", "rand32", "rand64"])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-natural", action="store_true")
    ap.add_argument("--no-reference", action="store_true")
    args = ap.parse_args()
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)  # libraries that print to stdout go to stderr; _emit writes to the real one
    sys.stdout = os.fdopen(real_stdout, "w")
    assert torch.cuda.is_available(), "bench_extras.py needs a HIP device"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    prs = [make_problem(args.workload, B_PER_GPU, seed=1234 + 1000 * k) for k in range(N_ROTATE)]
    pr = prs[0]
    pool = FreshBatches(args.workload, dev, prs, seed=4321)
    Hh, Ww = pool.H, pool.W
    bytes_moved_per_map = 24 * Hh * Ww

    def section(name, fn):
        _log(f"extras: {name}")
        try:
            _emit({name: fn()})
        except Exception as e:  # noqa: BLE001 - one section never sinks the others
            _emit({name: {"error": f"{type(e).__name__}: {str(e)[:300]}"}})

    with torch.no_grad():
        section("through_module", lambda: through_module_ms(pr, dev))
        # recurring batches, each visit placed by the order its searches finished in at the previous visit (the round-4 headline)
        run_h = Runner([pool.fixed(k, B_PER_GPU) for k in range(N_ROTATE)], dev, placement="hinted")

        def hinted():
            prewarm(run_h, dev, 0.1)
            dth, _ = timed_loop(run_h, args.steps, args.warmup, 1, dev, None)
            hint_ms = kernel_launch_ms(run_h, min(args.steps, 100), dev)[0]
            return {"value": B_PER_GPU * args.steps / dth, "ms_per_step": dth / args.steps * 1e3, "launch_ms_avg": hint_ms,
                    "roofline_frac": bytes_moved_per_map * B_PER_GPU / (hint_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "note": "RECURRING batches (three sets in rotation, each visited many times before the clock), bare C-ABI launches: every visit is searched "
                            "longest-first by the order its searches finished in at the previous visit (nastar_forward_ex order_out -> order; planner.Placement) "
                            "-- what a validation loop over a fixed set has from its second epoch on; this was the round-4 headline"}
        section("hinted", hinted)

        def pipelined():
            n = max(args.steps, 60)
            dtp = fresh_batches_pipelined(run_h, n, max(args.warmup, 9), dev)
            return None if dtp is None else {"value": B_PER_GPU * n / dtp, "ms_per_step": dtp / n * 1e3,
                                             "note": "never-searched batches in a pipeline: a side stream computes the NEXT batch's placement from its maps alone "
                                                     "(nastar_placement_predict) while this batch is searched; nothing from an earlier visit is used"}
        section("pipelined_with_predictor", pipelined)

        def secondary():
            sec = []
            for other in ("maze32", "rand32", "rand64"):
                if other == args.workload:
                    continue
                pr2 = make_problem(other, B_PER_GPU, seed=1234)
                run2 = Runner(pr2, dev)
                prewarm(run2, dev, 0.1)
                dt2, _ = timed_loop(run2, max(50, args.steps // 4), max(2, args.warmup // 4), 1, dev)
                a2, _, _ = kernel_launch_ms(run2, 50, dev)
                nbytes = 24 * run2.H * run2.W * B_PER_GPU  # cost == passable: the bytes that move
                a2n = None
                if run2.placement == "hinted" and not args.no_natural:
                    run2.placement = "natural"
                    a2n = kernel_launch_ms(run2, 50, dev)[0]
                    run2.placement = "hinted"
                sec.append({"workload": f"{other}: {B_PER_GPU} maps of {run2.H}x{run2.W}", "value": B_PER_GPU * max(50, args.steps // 4) / dt2,
                            "unit": "maps/s", "launch_ms_avg": a2, "hbm_frac": nbytes / (a2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "hbm_frac_bytes_per_cell": 24,
                            "placement": run2.placement, "launch_ms_avg_natural_order": a2n,
                            "hbm_frac_natural_order": (nbytes / (a2n * 1e-3) / 1e9 / HBM_PEAK_GBS) if a2n else None,
                            "mean_iters_per_map": float(run2.iters.float().mean().item()),
                            "max_iters_per_map": int(run2.iters.max().item()),
                            "gpu_matches_oracle_on_sample": oracle_check(pr2, run2.hist.cpu().numpy(), run2.paths.cpu().numpy(), 256),
                            "oracle_sample": "first 256 maps"})
                del run2, pr2
            # the two other readings of BASELINE.json's "tau = 0.25" (SURVEY.md section 0.3): training-mode budget Tmax = 0.25
            # (searches truncated after 256 selections) and g_ratio = 0.8, on the headline maze batch
            for label, kw in ((("maze32, training-mode budget Tmax=0.25 (max 256 steps)", {"max_iters": int(0.25 * W * W)}),
                               ("maze32, g_ratio=0.8 (eval mode)", {"g_ratio": 0.8})) if args.workload == "maze32" else ()):
                run2 = Runner(pr, dev, **kw)
                dt2, _ = timed_loop(run2, max(10, args.steps // 4), max(2, args.warmup // 4), 1, dev)
                a2, _, _ = kernel_launch_ms(run2, max(10, min(args.steps // 4, 50)), dev)
                sec.append({"workload": f"{label}: {B_PER_GPU} maps of 32x32", "value": B_PER_GPU * max(10, args.steps // 4) / dt2,
                            "unit": "maps/s", "launch_ms_avg": a2, "hbm_frac": bytes_moved_per_map * B_PER_GPU / (a2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "hbm_frac_bytes_per_cell": 24, "mean_iters_per_map": float(run2.iters.float().mean().item()),
                            "max_iters_per_map": int(run2.iters.max().item())})
                del run2
            return sec
        section("secondary", secondary)
        del run_h

        def in_flight():
            r = in_flight_through_api(dev, args.steps)
            _emit({"value_in_flight": r[0]["maps_per_s"]})  # maze32, VanillaAstar, through the Python object
            return r
        section("in_flight_through_api", in_flight)
        section("throughput_regime", lambda: throughput_regime(dev, max(args.steps, 240)))
    if (Hh, Ww) == (32, 32):
        ex = {}
        for name, fn in (("neural_astar_cnn_hip_bf16", lambda: neural_astar_forward_ms(pr, dev)),
                         ("neural_astar_cnn_hip_f16x3", lambda: neural_astar_f16x3_ms(pr, dev)),
                         ("neural_astar_unet_hip_f16", lambda: neural_astar_unet_ms(pr, dev, "f16")),
                         ("neural_astar_unet_hip_f16x3", lambda: neural_astar_unet_ms(pr, dev, "f16x3")),
                         ("encoder_train_step", lambda: encoder_train_step_ms(pr, dev)),
                         ("train_fwd_bwd_ms_per_4096_maps_Tmax025", lambda: training_step_ms(pr, dev)),
                         ("data_path_32x32", lambda: data_path_ms(dev)),
                         ("train_l1_step_Tmax025", lambda: {"batch_100": l1_training_step_ms(pr, dev, 100),
                                                            "batch_4096": l1_training_step_ms(pr, dev, 4096)})):
            _log(f"extras: {name}")
            try:
                ex[name] = fn()
            except Exception as e:  # noqa: BLE001
                ex[name] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            _emit({"extra": {**ex, "note": "encoder / training / data-path figures on the headline batch; none of them is the headline value"}})
    if not args.no_reference:
        def ref_gpu():
            chk = Runner([pool.fixed(0, B_PER_GPU)], dev, placement="dataset")
            chk.step()
            torch.cuda.synchronize(dev)
            return reference_on_this_gpu(pr, chk.hist.cpu().numpy(), chk.paths.cpu().numpy(), dev)
        section("reference_torch_on_this_gpu", ref_gpu)


if __name__ == "__main__":
    extras_main()
