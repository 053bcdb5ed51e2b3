"""GPU tests (-m gpu) of the drop-in boundary's run-time services added in 0.5.0 -- everything here is about WHEN and WHERE a map is
searched and how its status reaches the host, never about what is computed (that is test_gpu_parity.py): outputs must stay bit-identical
to the plain launch, which is itself pinned to the reference's goldens and the oracle.

  * nastar_forward_ex: status summary (pinned host words written by the search launch), checked placement orders, self-healing counter
  * ops.search_nograd: the no-autograd fast path == the torch.library custom op
  * ops.order_from_levels / DeviceMazeBatches: a placement from the data set's own start distances (reference utils/data.py:127-134,200-221)
  * parallel.InFlightPlanner: batches in flight == sequential planner.forward() (reference utils/training.py:63-87 loops)
"""
import ctypes
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "gpu-marked test needs a HIP device"
    return torch.device("cuda:0")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(_dev())


def _problems(B=512, H=32, seed=5, kind="maze"):
    from neural_astar.utils import synthetic as syn
    pr = syn.maze_maps(B, H, seed=seed) if kind == "maze" else syn.random_obstacle_maps(B, H, H, 0.2, seed=seed)
    return pr, tuple(_t(x) for x in pr)


def _levels(pr):
    """|opt_dist[start]| as the data set would carry it: 8-connected unit-cost distance of the start cell to the goal"""
    from neural_astar.utils import synthetic as syn
    B = pr.map_designs.shape[0]
    gi = pr.goal_maps.reshape(B, -1).argmax(1)
    si = pr.start_maps.reshape(B, -1).argmax(1)
    d = syn.geodesic_distance(pr.map_designs[:, 0] > 0, gi).reshape(B, -1)
    return d[np.arange(B), si].astype(np.int32)


def test_status_summary_is_written_by_the_launch_into_pinned_host_memory():
    """status_summary of nastar_forward_ex: cell c = 1 when some map ends with per-map status c; a pinned HOST buffer works (plain stores, no
    atomics) and is all-zero for a clean batch; a device buffer works too."""
    from neural_astar import _native, ops
    lib = _native.load()
    dev = _dev()
    m = np.ones((3, 1, 16, 16), np.float32)
    m[0, 0, 8, :] = 0  # wall: map 0 is unsolvable
    s = np.zeros_like(m)
    g = np.zeros_like(m)
    s[:, 0, 0, 0] = 1
    g[:, 0, 15, 15] = 1
    mt, st, gt = _t(m), _t(s), _t(g)
    board = ops.StatusBoard.of(dev)
    for sub, want3 in ((slice(0, 3), 1), (slice(1, 3), 0)):
        row = board.acquire()
        out = ops.search_nograd(mt[sub], st[sub], gt[sub], mt[sub], 0.5, 256, summary_ptr=board.ptr(row))
        torch.cuda.synchronize()
        r = board.read(row)
        assert (r is not None and r[ops.STATUS_UNSOLVABLE] == 1 and r[1:].sum() == 1) if want3 else (r is None)
        assert out[3].tolist() == ([3, 0, 0] if want3 else [0, 0])
        board.release(row)
        assert board.read(row) is None
    # a device buffer as the summary, through the raw C ABI
    summ = torch.zeros(ops.SUMMARY_WORDS, dtype=torch.int32, device=dev)
    c = mt[:, 0].contiguous()
    hist = torch.empty_like(c)
    paths = torch.empty(c.shape, dtype=torch.int64, device=dev)
    it = torch.empty(3, dtype=torch.int32, device=dev)
    stt = torch.empty(3, dtype=torch.int32, device=dev)
    rc = lib.nastar_forward_ex(c.data_ptr(), st[:, 0].contiguous().data_ptr(), gt[:, 0].contiguous().data_ptr(), c.data_ptr(), 3, 16, 16, 0.5, 256,
                               hist.data_ptr(), paths.data_ptr(), None, it.data_ptr(), stt.data_ptr(), None, None, 0, 0, None, None, summ.data_ptr(), None,
                               torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert summ.tolist() == [0, 0, 0, 1] + [0] * 12
    # the unit-cost promise broken by one map -> cell NASTAR_ERR_NOT_UNIT_COST
    pr, (m2, s2, g2) = _problems(64, 32)
    m3 = m2.clone()
    m3[5, 0, 0, 0] = 0.5
    row = board.acquire()
    out = ops.search_nograd(m3, s2, g2, m3, 0.5, 1024, flags=ops.FLAG_UNIT_COST, summary_ptr=board.ptr(row))
    torch.cuda.synchronize()
    r = board.read(row)
    assert r is not None and r[ops.STATUS_NOT_UNIT_COST] == 1 and int(out[3][5]) == ops.STATUS_NOT_UNIT_COST
    board.release(row)


def test_checked_order_falls_back_to_the_natural_order_and_says_so():
    """an `order` that is not a permutation.  Checked (the Python default): the launch ignores it -- every map is searched, outputs
    equal the plain launch, summary[BAD_ORDER] is set, the planner warns.  Unchecked (trusted) orders only promise not to touch memory
    outside the batch.  A wrong length is refused.  The counter cell of order_out heals itself."""
    from neural_astar import ops
    pr, (m, s, g) = _problems(300, 32)
    B = 300
    dev = _dev()
    ref = ops.search_nograd(m, s, g, m, 0.5, 1024)
    board = ops.StatusBoard.of(dev)
    dup = torch.arange(B, dtype=torch.int32, device=dev)
    dup[7] = 8          # 8 named twice, 7 never
    oob = torch.arange(B, dtype=torch.int32, device=dev)
    oob[0] = B + 5
    for bad in (dup, oob):
        row = board.acquire()
        buf = ops.new_placement_buffer(B, dev)
        got = ops.search_nograd(m, s, g, m, 0.5, 1024, order=bad, order_out=buf, check_order=True, summary_ptr=board.ptr(row))
        torch.cuda.synchronize()
        for k in range(4):
            assert torch.equal(ref[k], got[k]), k
        r = board.read(row)
        assert r is not None and r[ops.SUMMARY_BAD_ORDER] == 1 and r[:ops.SUMMARY_BAD_ORDER].sum() == 0
        board.release(row)
        w = buf.cpu().numpy()
        assert w[B] == 0 and np.array_equal(np.sort(w[:B]), np.arange(B))  # the order it leaves is a permutation again
    # a valid order passes the check and places
    perm = torch.randperm(B, device=dev).to(torch.int32)
    row = board.acquire()
    got = ops.search_nograd(m, s, g, m, 0.5, 1024, order=perm, check_order=True, summary_ptr=board.ptr(row))
    torch.cuda.synchronize()
    assert board.read(row) is None and all(torch.equal(ref[k], got[k]) for k in range(4))
    board.release(row)
    # the custom op checks by default as well
    got = torch.ops.nastar.astar_forward_ordered(m[:, 0], s[:, 0], g[:, 0], m[:, 0], 0.5, 1024, False, 0, dup, None)
    torch.cuda.synchronize()
    assert all(torch.equal(ref[k], got[k]) for k in range(4))
    # trusted + invalid: unspecified rows, but nothing outside the batch is touched (must simply not fault)
    ops.search_nograd(m, s, g, m, 0.5, 1024, order=oob, check_order=False)
    torch.cuda.synchronize()
    # lengths are exact (a hint built for another batch size is refused, not half-used)
    with pytest.raises(ValueError, match="exactly"):
        ops.search_nograd(m, s, g, m, 0.5, 1024, order=torch.arange(B + 1, dtype=torch.int32, device=dev))
    with pytest.raises(ValueError, match="exactly"):
        ops.search_nograd(m, s, g, m, 0.5, 1024, order_out=torch.zeros(B + 2, dtype=torch.int32, device=dev))
    # a counter cell that was not zeroed: one launch with a rotated (still complete) order, clean afterwards
    buf = ops.new_placement_buffer(B, dev)
    buf[B] = 17
    ops.search_nograd(m, s, g, m, 0.5, 1024, order_out=buf)
    torch.cuda.synchronize()
    w = buf.cpu().numpy()
    assert np.array_equal(np.sort(w[:B]), np.arange(B)) and w[B] == 17  # B increments from 17 wrap back to 17 ...
    buf2 = ops.new_placement_buffer(B, dev)
    buf2[B] = B + 1000  # ... and a value outside [0, B) is pulled inside by the first increment
    ops.search_nograd(m, s, g, m, 0.5, 1024, order_out=buf2)
    torch.cuda.synchronize()
    w2 = buf2.cpu().numpy()
    assert 0 <= w2[B] < B and np.array_equal(np.sort(w2[:B]), np.arange(B))
    # the planner warns once when a hint attached to the batch is rejected
    from neural_astar.planner import VanillaAstar
    import neural_astar.planner.differentiable_astar as DA
    DA._BAD_ORDER_WARNED = False
    va = VanillaAstar().to(dev).eval()
    s_bad = s.clone()
    s_bad.placement_order = ops.OrderHint(dup, trusted=False)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        out = va(m, s_bad, g)
    assert torch.equal(out.histories[:, 0], ref[0]) and any("not a permutation" in str(w.message) for w in rec)


def test_backward_replay_checks_its_order_too():
    from neural_astar import _native, ops
    from neural_astar.utils import synthetic as syn
    lib = _native.load()
    dev = _dev()
    pr, (m, s, g) = _problems(96, 32)
    B = 96
    cost = _t(syn.random_costs(B, 32, 32, seed=3))
    hist, paths, iters, status, log = torch.ops.nastar.astar_forward(cost[:, 0], s[:, 0], g[:, 0], m[:, 0], 0.5, 256, True)
    up = torch.randn(B, 32, 32, device=dev)
    tb = (iters.amax() - 1).to(torch.int32).reshape(1)
    ref = torch.ops.nastar.astar_backward_replay(up, cost[:, 0], s[:, 0], g[:, 0], m[:, 0], log, 0.5, 256, iters, tb)
    dup = torch.arange(B, dtype=torch.int32, device=dev)
    dup[3] = 4
    nbytes = int(lib.nastar_backward_workspace_bytes(B, 32, 32, 256))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = torch.full((B, 32, 32), float("nan"), device=dev)
    c, st, gt, mt = (x[:, 0].contiguous() for x in (cost, s, g, m))
    rc = lib.nastar_backward_replay_ordered(up.data_ptr(), None, None, None, c.data_ptr(), st.data_ptr(), gt.data_ptr(), mt.data_ptr(), log.data_ptr(),
                                            B, 32, 32, 0.5, 256, iters.data_ptr(), tb.data_ptr(), out.data_ptr(), ws.data_ptr(), nbytes,
                                            ops.FLAG_CHECK_ORDER, dup.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(out, ref)  # every map replayed although the order named map 3 never


def test_fast_path_equals_the_custom_op():
    """ops.search_nograd (what forward() calls when no gradient can flow) == torch.ops.nastar.astar_forward, for the reference's [B,1,H,W]
    tensors, a multi-channel input (channel 0 is the map, reference :177-180), non-contiguous inputs, with a selection log."""
    from neural_astar import ops
    from neural_astar.utils import synthetic as syn
    pr, (m, s, g) = _problems(200, 32)
    u = _t(syn.random_costs(200, 32, 32, seed=9))
    for cost, passable, log in ((m, m, False), (u, m, True)):
        ref = torch.ops.nastar.astar_forward(cost[:, 0], s[:, 0], g[:, 0], passable[:, 0], 0.5, 1024, log)
        got = ops.search_nograd(cost, s, g, passable, 0.5, 1024, log)
        torch.cuda.synchronize()
        for k in range(4):
            assert torch.equal(ref[k], got[k]), k
        if log:
            mask = torch.arange(1024, device=m.device)[None, :] < ref[2][:, None]
            assert torch.equal(torch.where(mask, ref[4], -1), torch.where(mask, got[4], -1))
    m3 = torch.cat((m, torch.rand_like(m), torch.rand_like(m)), 1)  # [B,3,H,W]: channel 0 counts
    s3, g3 = torch.cat((s, s), 1), torch.cat((g, g), 1)
    ref = torch.ops.nastar.astar_forward(m[:, 0], s[:, 0], g[:, 0], m[:, 0], 0.5, 1024, False)
    got = ops.search_nograd(m3, s3, g3, m3, 0.5, 1024)
    assert all(torch.equal(ref[k], got[k]) for k in range(4))
    mT = m.transpose(2, 3).contiguous().transpose(2, 3)  # same values, non-contiguous strides
    got = ops.search_nograd(mT, s, g, mT, 0.5, 1024)
    assert all(torch.equal(ref[k], got[k]) for k in range(4))
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.search_nograd(m.cpu(), s.cpu(), g.cpu(), m.cpu(), 0.5, 1024)


def test_placement_from_the_data_set_distances():
    """ops.order_from_levels: a permutation, longest route first; searching with it changes nothing but the launch's duration; the module picks
    it up from start_maps.placement_order (what DeviceMazeBatches attaches), under no_grad and under autograd, and the fused training step too."""
    from neural_astar import ops
    from neural_astar.planner import NeuralAstar, VanillaAstar
    from neural_astar.utils.training import fused_l1_step
    pr, (m, s, g) = _problems(1024, 32)
    B = 1024
    lv = _levels(pr)
    order = ops.order_from_levels(_t(lv))
    o = order.cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(B)) and (np.diff(lv[o]) <= 0).all()
    o2 = ops.order_from_levels(_t(-lv.astype(np.float32))).cpu().numpy()  # the files store NEGATIVE float distances: |.| is taken
    assert np.array_equal(np.sort(o2), np.arange(B)) and np.array_equal(lv[o2], lv[o])
    va = VanillaAstar().to(_dev()).eval()
    ref = va(m, s, g)
    it_ref = va.astar.last_iters.clone()
    sh = ops.attach_order(s.clone(), _t(lv))
    assert isinstance(sh.placement_order, ops.OrderHint) and sh.placement_order.trusted
    for chk in (True, "deferred", False):
        va.astar.check_solvable = chk
        out = va(m, sh, g)
        assert torch.equal(out.histories, ref.histories) and torch.equal(out.paths, ref.paths) and torch.equal(va.astar.last_iters, it_ref)
    va.astar.raise_if_unsolvable()
    # long routes are long searches: the head of the order holds more steps than its tail (why the placement pays)
    it = it_ref.cpu().numpy()
    assert it[o[:128]].mean() > 1.5 * it[o[-128:]].mean()
    # under autograd (NeuralAstar training mode): same gradients with and without the hint
    torch.manual_seed(0)
    na = NeuralAstar(encoder_arch="CNN", Tmax=0.25).to(_dev()).train()
    traj = ref.paths.float()
    grads = []
    for starts in (s[:256].clone(), ops.attach_order(s[:256].clone(), _t(lv[:256]))):
        na.zero_grad()
        loss, out = fused_l1_step(na, m[:256], starts, g[:256], traj[:256])
        loss.backward()
        # (BatchNorm running statistics advance between the two passes; in train mode the step does not read them)
        grads.append([p.grad.clone() for p in na.encoder.parameters() if p.grad is not None] + [loss.detach().clone()])
    for a, b in zip(*grads):
        assert torch.equal(a, b)


def test_device_maze_batches_attach_a_placement(tmp_path):
    from neural_astar import ops
    from neural_astar.utils import synthetic as syn
    from neural_astar.utils.data import create_device_loader
    f = str(tmp_path / "mazes.npz")
    syn.write_maze_npz(f, n_train=64, n_valid=8, n_test=8, size=32, seed=3)
    loader = create_device_loader(f, "train", 32, _dev())
    maps, starts, goals, trajs = next(iter(loader))
    hint = starts.placement_order
    assert isinstance(hint, ops.OrderHint) and hint.trusted and hint.order.dtype == torch.int32
    o = hint.order.cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(32))
    # levels = optimal distance of the sampled start = length of the optimal trajectory
    tl = trajs.sum((1, 2, 3)).cpu().numpy()
    assert (np.diff(tl[o]) <= 0).all()
    loader.emit_placement = False
    assert not hasattr(next(iter(loader))[1], "placement_order")


def test_in_flight_planner_equals_sequential_forward():
    """8 x 4096 maps in flight over 4 streams == 8 sequential forward() calls (histories, paths); a batch with a NON-BINARY map is re-run on
    the general kernel at collection (unit_cost='auto' without a per-call wait); an unsolvable map raises at collection, naming the batch;
    NeuralAstar (encoder + search per stream) works too."""
    from neural_astar.parallel import InFlightPlanner
    from neural_astar.planner import NeuralAstar, VanillaAstar
    from neural_astar.planner.differentiable_astar import UnsolvableMapError
    from neural_astar.utils import synthetic as syn
    dev = _dev()
    batches = []
    for k in range(8):
        pr = syn.maze_maps(4096, 32, seed=100 + k) if k % 2 == 0 else syn.random_obstacle_maps(4096, 32, 32, 0.25, seed=100 + k)
        batches.append(tuple(_t(x) for x in pr))
    va = VanillaAstar().to(dev).eval()
    with torch.no_grad():
        seq = [va(*b) for b in batches]
    fly = InFlightPlanner(va, streams=4)
    outs = fly.plan_many(batches)
    assert len(outs) == 8 and fly.reruns == 0
    for a, b in zip(seq, outs):
        assert torch.equal(a.histories, b.histories) and torch.equal(a.paths, b.paths) and b.intermediate_results == []
    lazy = list(fly.plan_iter(batches, window=3))
    assert len(lazy) == 8 and all(torch.equal(a.paths, b.paths) for a, b in zip(seq, lazy))
    # a non-binary map in batch 2: that batch alone is re-run on the general kernel, outputs == the sequential call's
    nb = [tuple(x.clone() for x in b) for b in batches[:4]]
    nb[2][0][17, 0, 3, 3] = 0.25
    with torch.no_grad():
        seq_nb = va(*nb[2])
    outs = fly.plan_many(nb)
    assert fly.reruns == 1 and torch.equal(outs[2].histories, seq_nb.histories) and torch.equal(outs[2].paths, seq_nb.paths)
    assert torch.equal(outs[1].paths, seq[1].paths)
    # an unsolvable map: raised when the results are collected
    bad = [tuple(x.clone() for x in b) for b in batches[:3]]
    bad[1][0][5].zero_()
    with pytest.raises(UnsolvableMapError, match="batch #1"):
        fly.plan_many(bad)
    assert fly.plan_many(batches[:2])[1].paths.equal(seq[1].paths)  # the planner is usable afterwards
    fly_nc = InFlightPlanner(va, streams=2, check_solvable=False, unit_cost=False)
    assert len(fly_nc.plan_many(bad)) == 3
    # NeuralAstar: encoder and search of a batch on its stream
    torch.manual_seed(0)
    na = NeuralAstar(encoder_arch="CNN").to(dev).eval()
    small = [tuple(x[:256] for x in b) for b in batches[:4]]
    with torch.no_grad():
        seq_na = [na(*b) for b in small]
    outs = InFlightPlanner(na, streams=2).plan_many(small)
    for a, b in zip(seq_na, outs):
        assert torch.equal(a.histories, b.histories) and torch.equal(a.paths, b.paths)


def test_in_flight_planner_holds_what_a_foreign_stream_uses():
    """Memory the caching allocator hands out for the CURRENT stream but a launch uses on ANOTHER one must outlive that launch: the HBM
    slab of maps larger than LDS (every batch in flight needs its own -- a freed slab would be handed to the next submit while the first
    search still runs in it), the 16-byte workspace of a checked order, and the contiguous copies of strided inputs (made before the
    streams are ordered, held with the batch).  6 batches of 150x150 maps over 3 streams, strided inputs with a hand-made (checked)
    order == the sequential forward() calls."""
    from neural_astar import ops
    from neural_astar.parallel import InFlightPlanner
    from neural_astar.planner import VanillaAstar
    from neural_astar.utils import synthetic as syn
    dev = _dev()
    va = VanillaAstar().to(dev).eval()
    big = [tuple(_t(x) for x in syn.random_obstacle_maps(24, 150, 150, 0.2, seed=900 + k)) for k in range(6)]
    assert not ops.in_lds(150, 150)
    with torch.no_grad():
        seq = [va(*b) for b in big]
    fly = InFlightPlanner(va, streams=3)
    for rep in range(2):
        outs = fly.plan_many(big)
        for a, b in zip(seq, outs):
            assert torch.equal(a.histories, b.histories) and torch.equal(a.paths, b.paths)
    # a launch on a foreign stream that needs a workspace refuses to run without a holder for it; strided maps are refused too
    m, s, g = big[0]
    st = torch.cuda.Stream(dev)
    with pytest.raises(ValueError, match="keep"):
        ops.search_nograd(m, s, g, m, 0.5, 150 * 150, stream_ptr=st.cuda_stream)
    wide = torch.zeros((24, 1, 150, 300), device=dev)
    with pytest.raises(ValueError, match="contiguous"):
        ops.search_nograd(wide[..., ::2], s, g, wide[..., ::2], 0.5, 150 * 150, stream_ptr=st.cuda_stream, keep=[])
    # strided inputs + a checked order through the planner: copies and the order-check workspace are held until collection
    small = []
    for k in range(6):
        mm, ss, gg = (_t(x) for x in syn.maze_maps(512, 32, seed=950 + k))
        pad = torch.zeros((512, 1, 32, 64), device=dev)
        pad[..., ::2] = mm
        ss.placement_order = torch.randperm(512, device=dev).to(torch.int32)  # (a bare tensor: not trusted, checked on the device)
        small.append((pad[..., ::2], ss, gg))
    with torch.no_grad():
        seq = [va(b[0].contiguous(), b[1], b[2]) for b in small]
    fly = InFlightPlanner(va, streams=3, use_placement=True)
    for rep in range(2):
        outs = fly.plan_many(small)
        for a, b in zip(seq, outs):
            assert torch.equal(a.histories, b.histories) and torch.equal(a.paths, b.paths)


def test_forward_host_overhead_is_bounded():
    """The module boundary must not dominate a 4096-map call (+70 %).  Deferred checking never waits for the device: 48 calls
    (fewer than the 64 verdicts the module lets queue up before it waits for the oldest) are ISSUED in far less time than they take to
    run -- the host runs ahead -- and a checked (sync) call costs at most ~60 us more than the launch it waits for."""
    import gc
    import time
    from neural_astar import ops
    from neural_astar.planner import VanillaAstar
    pr, (m, s, g) = _problems(4096, 32, seed=1234)
    dev = _dev()
    va = VanillaAstar().to(dev).eval()
    # best of three rounds, cyclic garbage collector off: one round in twenty or so contained ONE host-side stall of 50-160 ms (seen in the
    # bench's child process and here, never in a small dedicated probe) -- what a generation-2 collection costs in a process whose heap holds
    # a whole test session; bench.py's timed loop switches the collector off for the same reason.  The bounds are on what a call costs
    gc_was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        with torch.no_grad():
            va.astar.check_solvable = "deferred"
            for _ in range(10):
                va(m, s, g)
            torch.cuda.synchronize()
            t_issue = t_all = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(48):
                    va(m, s, g)
                ti = time.perf_counter() - t0
                torch.cuda.synchronize()
                ta = time.perf_counter() - t0
                va.astar.raise_if_unsolvable()
                if ti < t_issue:
                    t_issue, t_all = ti, ta
            va.astar.check_solvable = True
            for _ in range(10):
                va(m, s, g)
            torch.cuda.synchronize()
            t_sync = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(40):
                    va(m, s, g)
                t_sync = min(t_sync, (time.perf_counter() - t0) / 40)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.search_nograd(m, s, g, m, 0.5, 1024, flags=ops.FLAG_UNIT_COST)
            e1.record()
            torch.cuda.synchronize()
            kern = e0.elapsed_time(e1) / 20 * 1e-3
    finally:
        if gc_was:
            gc.enable()
    assert t_issue / 48 < 60e-6, f"issuing a deferred forward() takes {t_issue / 48 * 1e6:.0f} us of host time"
    assert t_issue < 0.6 * t_all
    assert t_sync < kern + 80e-6, f"sync forward() {t_sync * 1e6:.0f} us vs kernel {kern * 1e6:.0f} us"


def test_packed_sink_is_filled_by_the_search_launch_itself():
    """planner.astar.packed_sink (the collation slot of parallel.BucketedCollator): the checked no-grad call hands it to nastar_forward_ex as
    packed_out -- the launch emits the 2-bit-per-cell masks itself (32x32: fused; 20x45: the C ABI adds its pack launch) -- and reports it as
    last_packed; a call that cannot (autograd) reports None.  The packed bytes equal pack_masks() of the outputs."""
    from neural_astar import parallel
    from neural_astar.planner import VanillaAstar
    from neural_astar.utils import synthetic as syn
    dev = torch.device("cuda:0")
    for (H, W, B) in ((32, 32, 512), (20, 45, 16)):
        pr = syn.random_obstacle_maps(B, H, W, 0.2, seed=11)
        m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
        va = VanillaAstar().to(dev).eval()
        col = parallel.BucketedCollator(bucket=4)
        slot = col.next_slot(B, H, W, dev)
        slot.fill_(0xAA)
        va.astar.packed_sink = slot
        with torch.no_grad():
            out = va(m, s, g)
        assert va.astar.last_packed is slot
        torch.cuda.synchronize()
        assert torch.equal(slot, parallel.pack_masks(out.histories, out.paths))
        h2, p2 = parallel.unpack_masks(slot, H, W)
        assert torch.equal(h2, out.histories) and torch.equal(p2, out.paths)
        va.astar.packed_sink = None
        with torch.no_grad():
            va(m, s, g)
        assert va.astar.last_packed is None
    da_in = torch.from_numpy(syn.random_costs(4, 32, 32, seed=1)).to(dev).requires_grad_(True)
    pr = syn.random_obstacle_maps(4, 32, 32, 0.2, seed=12)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    va = VanillaAstar().to(dev).eval()
    va.astar.packed_sink = torch.empty((4, 256), dtype=torch.uint8, device=dev)
    va.astar(da_in, s, g, m)
    assert va.astar.last_packed is None  # (under autograd the custom op runs: the collator packs the outputs itself)


def test_native_host_lane_equals_the_python_lane():
    """lib/_nastar_fastlane.so (csrc/nastar_fastlane.cpp: allocation + nastar_forward_ex + the poll of the completion flag in C++) is a faster
    way to ISSUE the same launch: outputs, verdicts, placements and the fall-through cases must be those of the Python lane."""
    from neural_astar import _native, ops
    from neural_astar.planner import VanillaAstar
    from neural_astar.planner.differentiable_astar import DifferentiableAstar, UnsolvableMapError
    fl = _native.load_fastlane()
    assert fl is not None, "lib/_nastar_fastlane.so missing: __graft_entry__.build() builds it (make -C neural-astar_amd/csrc fastlane)"
    dev = _dev()
    pr, (m, s, g) = _problems(1024, 32, seed=77)
    u = torch.rand_like(m) + 0.05

    calls = {"n": 0}
    real = fl[0].search

    class _Counting:
        @staticmethod
        def search(*a):
            calls["n"] += 1
            return real(*a)
    _native._fastlane = (_Counting, fl[1], fl[2])
    fl_counting = _native._fastlane
    try:
        va = VanillaAstar().to(dev).eval()
        da = DifferentiableAstar(0.5, 1.0).to(dev).eval()
        with torch.no_grad():
            n0 = calls["n"]
            a = va(m, s, g)
            assert calls["n"] == n0 + 1  # the lane was taken
            _native._fastlane = None
            b = va(m, s, g)
            _native._fastlane = fl_counting
            assert torch.equal(a.histories, b.histories) and torch.equal(a.paths, b.paths) and a.histories.shape == (1024, 1, 32, 32)
            a = da(u, s, g, m)
            _native._fastlane = None
            b = da(u, s, g, m)
            _native._fastlane = fl_counting
            assert torch.equal(a.histories, b.histories) and torch.equal(a.paths, b.paths)
            # a placement hint rides along; outputs do not depend on it -- as an order sorted at batch assembly, or as the loader's levels, which
            # the native call sorts in front of its search launch (and the Python lane on first use)
            lv = torch.from_numpy(_levels(pr)).to(dev)
            ops.attach_order(s, lv)
            c = va(m, s, g)
            ops.attach_levels(s, lv)
            assert s.placement_order.order is None
            c2 = va(m, s, g)
            assert s.placement_order.order is None  # (sorted inside the native call: nothing cached on the hint)
            _native._fastlane = None
            c3 = va(m, s, g)
            _native._fastlane = fl_counting
            o = s.placement_order.order  # (the Python lane sorts on first use and keeps the order on the hint; equal levels come in arbitrary order)
            assert o is not None and torch.equal(torch.sort(o.long())[0], torch.arange(o.numel(), device=dev)) and bool((lv[o.long()][1:] <= lv[o.long()][:-1]).all())
            del s.placement_order
            plain = va(m, s, g)
            assert all(torch.equal(x.histories, plain.histories) and torch.equal(x.paths, plain.paths) for x in (c, c2, c3))
            # strided inputs, gradients, deferred checking: not this lane's calls -- same answers from the general path
            n0 = calls["n"]
            big = torch.cat((m, m), 1)
            out = va(big[:, 1:2], s, g)
            assert torch.equal(out.histories, c.histories)
            va.astar.check_solvable = "deferred"
            va(m, s, g)
            va.astar.raise_if_unsolvable()
            va.astar.check_solvable = True
        ug = u.clone().requires_grad_(True)
        da(ug, s, g, m).histories.sum().backward()
        assert calls["n"] == n0 + 1 and ug.grad is not None  # (only the strided call tried the lane, and was sent back)
        # an unsolvable map raises in the same call, naming its row
        mm = m.clone()
        mm[3, 0] = 0
        mm[3, 0].view(-1)[s[3, 0].view(-1).argmax()] = 1
        mm[3, 0].view(-1)[g[3, 0].view(-1).argmax()] = 1
        with torch.no_grad(), pytest.raises(UnsolvableMapError, match="rows \\[3\\]"):
            va(mm, s, g)
        with torch.no_grad():
            assert torch.equal(va(m, s, g).histories, c.histories)  # the status row went back clean
    finally:
        _native._fastlane = fl


def test_encoder_routes_are_recorded_and_a_fall_back_to_torch_nn_speaks_up():
    """NeuralAstar.encode decides per call between the MI355X encoder kernels and torch.nn.  The decision is recorded
    (planner.last_encoder_route), the BASELINE configurations take the kernels in eval() AND train() -- config 3 (U-Net on 32x32 mazes),
    config 5 (CNNDownSize, 96x96 RGB -> 12x12), the default CNN -- and a shape the kernels do not take warns once ('hip_strict': raises)."""
    from neural_astar.planner import NeuralAstar
    dev = _dev()
    torch.manual_seed(0)
    pr, (m, s, g) = _problems(32, 32)
    traj = torch.zeros_like(s)

    def routes(na, maps, starts, goals):
        na.to(dev)
        out = {}
        na.eval()
        with torch.no_grad():
            na(maps, starts, goals)
        out["eval"] = na.last_encoder_route
        na.train()
        o = na(maps, starts, goals)
        torch.nn.L1Loss()(o.histories, torch.zeros_like(o.histories)).backward()
        out["train"] = na.last_encoder_route
        return out

    r = routes(NeuralAstar(encoder_arch="CNN", Tmax=0.25), m, s, g)
    assert r == {"eval": "hip:CNN-infer-img32/f16x3", "train": "hip:CNN-train/f16x3"}, r
    r = routes(NeuralAstar(encoder_arch="Unet", Tmax=0.25), m, s, g)   # BASELINE config 3
    assert r == {"eval": "hip:Unet-infer/f16x3", "train": "hip:Unet-train/f16x3"}, r
    img = torch.rand(16, 3, 96, 96, device=dev)                          # BASELINE config 5 (scripts/config/train_warcraft.yaml)
    s12 = torch.zeros(16, 1, 12, 12, device=dev)
    g12 = torch.zeros(16, 1, 12, 12, device=dev)
    s12[:, 0, 0, 0] = 1
    g12[:, 0, -1, -1] = 1
    r = routes(NeuralAstar(encoder_input="rgb+", encoder_arch="CNNDownSize", encoder_depth=3, const=10.0, learn_obstacles=True, Tmax=0.25), img, s12, g12)
    assert r == {"eval": "hip:CNNDownSize-infer/f32", "train": "hip:CNNDownSize-train/f16x3"}, r
    # a map wider than the generic convolution's flat row tile (126 px): round 6 -- 2-D tiles, still the kernels (eval and train)
    wide = torch.ones(2, 1, 8, 160, device=dev)
    sw = torch.zeros_like(wide)
    gw = torch.zeros_like(wide)
    sw[:, 0, 0, 0] = 1
    gw[:, 0, 7, 159] = 1
    r = routes(NeuralAstar(encoder_arch="CNN", encoder_depth=2, Tmax=0.25), wide, sw, gw)
    assert r == {"eval": "hip:CNN-infer-flat/f16x3", "train": "hip:CNN-train/f16x3"}, r
    # EVAL mode with gradients on (BatchNorm on its running statistics under autograd): round 6 -- the CNN stacks take the kernels there too
    na = NeuralAstar(encoder_arch="CNN", encoder_depth=2).to(dev).eval()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        na(wide, sw, gw).histories.sum().backward()
    assert na.last_encoder_route == "hip:CNN-evalgrad/f16x3", na.last_encoder_route
    # ... and, later in round 6, the U-Net
    na = NeuralAstar(encoder_arch="Unet").to(dev).eval()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        na(m[:2], s[:2], g[:2]).histories.sum().backward()
    assert na.last_encoder_route == "hip:Unet-evalgrad/f16x3", na.last_encoder_route
    # what the kernels still do not take -- a U-Net of encoder_depth 5 on 32x32 maps in training mode (its deepest level would be 1x1: one pixel
    # per image is below the weight-gradient kernel's chunks; from 64x64 on depth 5 trains on the kernels too): torch.nn, said out loud once
    na = NeuralAstar(encoder_arch="Unet", encoder_depth=5).to(dev).train()
    wide, sw, gw = m[:2], s[:2], g[:2]
    with pytest.warns(RuntimeWarning, match="not covered by the MI355X encoder kernels"):
        na(wide, sw, gw)
    assert na.last_encoder_route.startswith("torch.nn (fell through from hip_f16x3")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        na(wide, sw, gw)  # once per reason
    na.encoder_backend = "hip_strict"
    with pytest.raises(RuntimeError, match="hip_strict"):
        na(wide, sw, gw)
    na.encoder_backend = "torch"
    na(wide, sw, gw)
    assert na.last_encoder_route == "torch.nn"


def test_validation_pass_in_flight_equals_the_per_batch_steps(tmp_path):
    """PlannerModule.validate(loader): planner + VanillaAstar pair per batch in one launch, launches of consecutive batches in flight ==
    the mean of what validation_step logs batch by batch (reference utils/training.py:63-87)."""
    from types import SimpleNamespace
    from neural_astar.planner import NeuralAstar
    from neural_astar.utils import synthetic as syn
    from neural_astar.utils.data import create_device_loader
    from neural_astar.utils.training import PlannerModule
    f = str(tmp_path / "mazes.npz")
    syn.write_maze_npz(f, n_train=8, n_valid=96, n_test=8, size=32, seed=5)
    gen = torch.Generator(device=_dev())
    gen.manual_seed(3)
    batches = list(create_device_loader(f, "valid", 32, _dev(), generator=gen))
    assert len(batches) == 3
    torch.manual_seed(0)
    mod = PlannerModule(NeuralAstar(encoder_arch="CNN").to(_dev()), SimpleNamespace(params=SimpleNamespace(lr=1e-3))).to(_dev())
    mod.planner.eval()
    mod.vanilla_astar.eval()
    acc = {}
    with torch.no_grad():
        for i, b in enumerate(batches):
            mod.logged = {}
            mod.validation_step(b, i)
            for k, v in mod.logged.items():
                acc[k] = acc.get(k, 0.0) + float(v)
    ref = {k: v / len(batches) for k, v in acc.items()}
    got = mod.validate(batches, streams=3)
    assert set(got) == set(ref) == {"metrics/val_loss", "metrics/p_opt", "metrics/p_exp", "metrics/h_mean"}
    for k in ref:
        assert abs(float(got[k]) - ref[k]) < 1e-6, (k, float(got[k]), ref[k])


def test_status_verdicts_under_stress_are_never_lost_or_misattributed():
    """The completion flag is raised by the LAST search to end, ordered behind every status cell of its launch by a system-scope release /
    acquire pair; rows and counters are reused by the very next call.  400 calls in a row, a random third of them with ONE unsolvable map at
    a random row (early or late finisher), sync and deferred checking: every bad call must raise exactly once, no good call may."""
    from neural_astar.planner import VanillaAstar
    from neural_astar.planner.differentiable_astar import UnsolvableMapError
    rng = np.random.default_rng(7)
    pr, (m, s, g) = _problems(1024, 32, seed=21)
    dev = _dev()
    bad_maps = []
    for _ in range(8):
        mb = m.clone()
        b = int(rng.integers(0, 1024))
        gi = int(g[b].reshape(-1).argmax())
        gy, gx = divmod(gi, 32)
        mb[b, 0, max(gy - 1, 0):gy + 2, max(gx - 1, 0):gx + 2] = 0  # wall the goal in
        mb[b, 0, gy, gx] = 1
        if int(s[b].reshape(-1).argmax()) != gi and abs(int(s[b].reshape(-1).argmax()) // 32 - gy) + abs(int(s[b].reshape(-1).argmax()) % 32 - gx) > 2:
            bad_maps.append((mb, b))
    assert bad_maps
    va = VanillaAstar().to(dev).eval()
    with torch.no_grad():
        ref = va(m, s, g)
        raised = expected = 0
        for i in range(400):
            if rng.random() < 0.33:
                mb, b = bad_maps[int(rng.integers(len(bad_maps)))]
                expected += 1
                with pytest.raises(UnsolvableMapError):
                    va(mb, s, g)
                raised += 1
                assert int(va.astar.last_status[b]) == 3 and int((va.astar.last_status != 0).sum()) == 1
            else:
                out = va(m, s, g)
                if i % 50 == 0:
                    assert torch.equal(out.paths, ref.paths)
        assert raised == expected > 50
        # deferred: verdicts arrive with later calls, each exactly once, naming its own call
        va2 = VanillaAstar().to(dev).eval()
        va2.astar.check_solvable = "deferred"
        want, got = [], []
        for i in range(300):
            bad = rng.random() < 0.2
            launched = True
            try:
                va2(bad_maps[0][0] if bad else m, s, g)
            except UnsolvableMapError as e:  # the verdict of an EARLIER call, delivered before this call launched anything
                got.append(int(str(e).split("call #")[1].split(" ")[0]))
                launched = False
            if bad and launched:
                want.append(va2.astar._calls)
        while True:
            try:
                va2.astar.raise_if_unsolvable()
                break
            except UnsolvableMapError as e:
                got.append(int(str(e).split("call #")[1].split(" ")[0]))
        assert sorted(got) == want, (want[:10], sorted(got)[:10])
