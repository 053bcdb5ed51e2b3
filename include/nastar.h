/*
 * nastar.h -- C ABI of the MI355X-native differentiable A* hot path (libnastar_hip.so).
 *
 * The reference (omron-sinicx/neural-astar) is pure Python/PyTorch and has no FFI; these entry points are
 * what a binding for its ONE hot path would bind.  Each function names the reference code it replaces
 * (paths relative to /root/reference/src/neural_astar/planner/):
 *
 *   nastar_forward          <- DifferentiableAstar.forward   differentiable_astar.py:150-267
 *                              (get_heuristic :26-52, _st_softmax_noexp :55-74, expand :77-93,
 *                               backtrack :96-125 all fused into one launch)
 *   nastar_forward_ex       <- the same code; every option of the launch in one call (0.5.0): placement (order / order_out, optionally
 *                              checked on the device), bit-packed masks, and the launch's STATUS SUMMARY + completion flag in (pinned host)
 *                              memory -- what the Python layer calls.  nastar_forward / _ordered / _packed are its subsets.
 *   nastar_placement_from_levels <- (new) a placement from per-map levels the caller has: |opt_dists[start]| of the reference's maze
 *                              files (utils/data.py:127-134, 200-221)
 *   nastar_backward_replay  <- the autograd graph PyTorch records for that forward
 *                              (what loss.backward() runs in utils/training.py:55-61)
 *   nastar_heuristic        <- get_heuristic                 differentiable_astar.py:26-52 (debug/parity)
 *   nastar_workspace_bytes  <- (new) workspace sizing; PyTorch owns every allocation
 *   nastar_l1_loss / nastar_backward_l1_replay <- loss = nn.L1Loss()(histories, opt_trajs); loss.backward()   utils/training.py:55-61
 *   nastar_policy_rollout   <- MazeDataset.get_opt_traj / next_loc                                      utils/data.py:171-199,222-244
 *   nastar_encoder_cnn_forward <- NeuralAstar.encode with the CNN encoder in eval mode                  astar.py:154-180, encoder.py:32-34,60-78
 *   nastar_pack_outputs / nastar_unpack_outputs <- (new) multi-GPU collation payload, see below
 *
 * Conventions
 *   - plain C types only; the caller owns every buffer (inputs, outputs, workspace); no allocation and no
 *     exception crosses the ABI; every function returns an int status (0 = ok).
 *   - all pointers are DEVICE pointers (HBM) unless the name ends in _host.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are asynchronous with
 *     respect to the host and re-entrant per stream.
 *   - maps are dense row-major [B,H,W] fp32 (the reference's [B,1,H,W] tensors with the unit channel dropped):
 *     cost >= 0 (the reference's encoders emit sigmoid outputs), start/goal one-hot, passable 1 = free cell.
 *   - max_iters = int(Tmax_eff * W * W) exactly as differentiable_astar.py:200-202 computes it
 *     (Tmax_eff = Tmax in training mode, 1.0 in eval mode).
 */
#ifndef NASTAR_H_
#define NASTAR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NASTAR_VERSION 600 /* 0.6.0: nastar_forward_batchloop_finish (the reference's batch loop to the letter, any size, no host round trip), NASTAR_FLAG_MARK_COUPLED; the A/B flags left the ABI; 0.5.0: nastar_forward_ex (status summary, checked placement), nastar_placement_from_levels; 0.4.1: nastar_forward_ordered (placement); 0.4.0: round-4 search instruction stream, unit-cost LDS layout */

/* status codes (function return values) */
#define NASTAR_OK 0
#define NASTAR_ERR_BAD_SHAPE 1   /* B,H,W,max_iters out of range                                     */
#define NASTAR_ERR_UNSUPPORTED 2 /* map too large for the implemented kernels                        */
#define NASTAR_ERR_UNSOLVABLE 3  /* per-map status only: open list ran empty (reference: NaN+IndexError) */
#define NASTAR_ERR_HIP 4         /* a HIP runtime call failed; see nastar_last_error()               */
#define NASTAR_ERR_NULL 5        /* a required pointer is NULL                                       */
#define NASTAR_ERR_WORKSPACE 6   /* workspace_bytes smaller than nastar_workspace_bytes()            */
#define NASTAR_ERR_NOT_UNIT_COST 7 /* per-map status only: NASTAR_FLAG_UNIT_COST was passed but this map holds a value other than 0.0 / 1.0;
                                      its outputs are all-zero -- run it again without the flag */

/* flags for nastar_workspace_bytes / nastar_forward* / nastar_backward_replay*.  (The A/B switches of earlier rounds -- older instruction
 * streams, the compiler-generated step, variants of the large-map kernel -- are not part of this ABI any more: they exist in the development
 * build only, `make -C neural-astar_amd/csrc dev`, csrc/nastar_dev_flags.h; an unknown flag bit is NASTAR_ERR_UNSUPPORTED.) */
#define NASTAR_FLAG_NONE 0
#define NASTAR_FLAG_UNIT_COST 64 /* forward: the caller promises that `cost` and `passable` are ONE binary tensor (VanillaAstar, reference
                                    astar.py:93-94; pass the same pointer twice): the LDS state drops the per-cell cost word (5.5 instead
                                    of 9.75 B/cell, 29 instead of 16 resident 32x32 maps per CU).  Same outputs as without the flag; the
                                    kernel checks every map while loading it and marks a map that breaks the promise with status
                                    NASTAR_ERR_NOT_UNIT_COST.  Ignored (general kernel) when cost != passable, a selection log is
                                    wanted or the map is not 32x32 / 64x64 */
#define NASTAR_FLAG_LOCKSTEP 1024 /* forward: the reference's batch loop TO THE LETTER for every map -- no exit at the goal: a selected goal is expanded like any cell, stays
                                   * on the open list (differentiable_astar.py:224) and the map is stepped on until exactly max_iters steps have been executed (or its
                                   * open list is empty); sel_log_out then records every step, goal selections included.  Any map size.  nastar_forward_batchloop_finish
                                   * is built from it.  backward (nastar_backward_replay*): the selection log comes from such a run -- the goal may be selected before
                                   * the log's last entry -- and is replayed by the general loop */
#define NASTAR_FLAG_MARK_COUPLED 32768 /* nastar_forward_ex: the launch also writes, per map, whether the map is in the BATCH-COUPLED class (see
                                   * NASTAR_SUMMARY_COUPLED) into the workspace (nastar_workspace_bytes(B,H,W,flags) bytes, or the larger
                                   * nastar_batchloop_workspace_bytes): the input of nastar_forward_batchloop_finish */
#define NASTAR_FLAG_CHECK_ORDER 256 /* nastar_forward_ex / nastar_forward_ordered / nastar_backward_replay_ordered: verify on the device that `order` is a
                                     * permutation of 0..B-1 (one small launch before the search) and IGNORE it when it is not -- every map is then
                                     * searched in the natural order and status_summary[NASTAR_SUMMARY_BAD_ORDER] is set.  Needs the workspace that
                                     * nastar_workspace_bytes(B,H,W,flags) / nastar_backward_workspace_bytes() report (16 bytes for LDS-resident
                                     * forward searches).  Without the flag `order` is TRUSTED: a map it never names is never searched and its
                                     * output rows are left as the caller allocated them. */
/* status_summary of nastar_forward_ex: NASTAR_SUMMARY_WORDS int32 cells, device or host-mapped memory, zeroed by the caller.  Cell c (a per-map
 * status code, 1..14) becomes 1 when SOME map of the launch ends with that status; cell NASTAR_SUMMARY_BAD_ORDER when a checked `order` was
 * rejected.  Plain idempotent stores, no atomics: a pinned host buffer works and turns "did any map fail?" into one 64-byte host read after the
 * stream (or an event) has been waited for -- no reduction launch, no device-to-host copy. */
#define NASTAR_SUMMARY_WORDS 16
#define NASTAR_SUMMARY_BAD_ORDER 15
/* cell 14, a NOTE, not an error: some map that reached its goal is not at a fixed point of the reference's batch loop.  The reference steps a
 * finished map until EVERY map of its batch selects its goal in the same step (differentiable_astar.py:224, :251); the kernels stop each map
 * at its own goal.  The outputs agree iff the goal's own expansion would open nothing that beats the goal -- always true for g_ratio in
 * [0.5, 1) with costs >= 0 (every shipped configuration), not for g_ratio < 0.5 with an expensive goal cell, g_ratio = 1 with a zero-cost one,
 * or negative costs: there the reference's histories of that map depend on the rest of its batch; ONE launch returns what the reference
 * returns for the map searched alone, nastar_forward_batchloop_finish completes it to the batch run.  (The unit-cost layout never sets it:
 * cost = 1 everywhere.) */
#define NASTAR_SUMMARY_COUPLED 14

int nastar_version(void);

/* Human-readable description of the last NASTAR_ERR_HIP on this thread ("" if none). */
const char* nastar_last_error(void);

/* Bytes of device workspace the forward/backward need for this problem size (may be 0). */
size_t nastar_workspace_bytes(int B, int H, int W, int flags);

/*
 * Forward search for B independent maps; one launch, no host synchronisation.
 *   histories_out [B,H,W] fp32   exact 0.0/1.0: cells moved to the closed list   (AstarOutput.histories)
 *   paths_out     [B,H,W] int64  0/1: back-tracked path incl. goal               (AstarOutput.paths)
 *   sel_log_out   [B,max_iters] int32 or NULL: flat index selected at each executed step of map b; entries at
 *                 positions >= iters_out[b] are left untouched (store_intermediate_results side channel,
 *                 differentiable_astar.py:210-216)
 *   iters_out     [B] int32      number of selection steps map b executed = (index of the step that selected
 *                 the goal)+1, or max_iters if the budget ran out first.  The reference's batch-wide loop
 *                 index is t_batch = max_b(iters_out[b]) - 1.
 *   status_out    [B] int32      NASTAR_OK or NASTAR_ERR_UNSOLVABLE per map
 */
int nastar_forward(const float* cost, const float* start, const float* goal, const float* passable, int B,
                   int H, int W, double g_ratio, int max_iters, float* histories_out, int64_t* paths_out,
                   int32_t* sel_log_out, int32_t* iters_out, int32_t* status_out, void* workspace,
                   size_t workspace_bytes, int flags, void* stream);

/*
 * nastar_forward with a PLACEMENT: workgroup i of the launch searches map order[i].  Outputs are indexed by map as always -- the
 * placement changes when and where a map is searched, never what is computed (same histories, paths, iters, status, log).
 * Why: one launch lasts as long as its longest search, and the hardware starts and arbitrates workgroups in index order.  With the
 * longest searches first they start at t = 0 and are the oldest wavefronts of their SIMD (4096 mazes of 32x32: 149 -> 112 us per
 * launch, 64x64 random maps 277 -> 234 us; profiles/r04/order_*.jsonl).  Nothing cheap predicts a search's length from its map, but
 * a data set is searched once per epoch (reference scripts/train.py:43-50 validates after every epoch on a fixed, unshuffled
 * loader, utils/data.py:40-47): the previous visit of the same batch is the predictor.
 *   order      [B] int32 device, a permutation of 0..B-1, or NULL (identity).  TRUSTED unless flags has NASTAR_FLAG_CHECK_ORDER.
 *   order_out  [B+1] int32 device or NULL: receives the `order` to pass at the next visit of the same batch -- the maps in REVERSE
 *              order of search completion in THIS launch when all B maps are resident at once (one atomic per map inside the
 *              kernel), sorted by their step counts (one more small launch on the same stream) when the launch takes several
 *              rounds of workgroups.  order_out[B] is the launch's counter: zero the buffer once, then reuse it (one buffer per launch in
 *              flight).  The counter wraps at B: it is back at its entry value when the launch has finished, and every rank is handed
 *              out exactly once whatever it held -- a buffer that was not zeroed gets a rotated, still complete order.
 *   packed_out NULL, or the bit-packed masks of nastar_forward_packed.
 * Maps whose search state lives in HBM (nastar_workspace_bytes > 0) take no placement: NASTAR_ERR_UNSUPPORTED if either is given.
 */
int nastar_forward_ordered(const float* cost, const float* start, const float* goal, const float* passable, int B,
                           int H, int W, double g_ratio, int max_iters, float* histories_out, int64_t* paths_out,
                           int32_t* sel_log_out, int32_t* iters_out, int32_t* status_out, uint8_t* packed_out,
                           void* workspace, size_t workspace_bytes, int flags, const int32_t* order, int32_t* order_out,
                           void* stream);

/*
 * nastar_forward with every option of the search launch (0.5.0): placement (`order` / `order_out`, see nastar_forward_ordered), the bit-packed
 * masks (`packed_out`, see nastar_forward_packed) and the per-launch STATUS SUMMARY (`status_summary`, NASTAR_SUMMARY_WORDS int32 cells, see
 * above).  Every optional pointer may be NULL; with all of them NULL this is nastar_forward.  flags: NASTAR_FLAG_*; with
 * NASTAR_FLAG_CHECK_ORDER the order is verified on the device first (workspace: nastar_workspace_bytes(B,H,W,flags) bytes).
 * Replaces the same reference code as nastar_forward (differentiable_astar.py:150-267).
 */
int nastar_forward_ex(const float* cost, const float* start, const float* goal, const float* passable, int B, int H, int W,
                      double g_ratio, int max_iters, float* histories_out, int64_t* paths_out, int32_t* sel_log_out,
                      int32_t* iters_out, int32_t* status_out, uint8_t* packed_out, void* workspace, size_t workspace_bytes,
                      int flags, const int32_t* order, int32_t* order_out, int32_t* status_summary, int32_t* completion_counter,
                      void* stream);

/*
 * completion_counter of nastar_forward_ex (optional, with status_summary): ONE int32 in DEVICE memory, 0 on entry and 0 again when the
 * launch has finished (it wraps at B; one cell per launch in flight).  Every search counts itself when it ENDS; the workgroup that
 * counts last sets status_summary[0] = 1, ordered behind every other summary cell of the launch (system-scope release / acquire).  With
 * status_summary in pinned host memory the host learns "every search is over, and this is the verdict" by POLLING one word
 * (nastar_host_wait_nonzero) -- no stream wait, no driver wake-up; the output tensors themselves are complete only in stream order
 * (the backtrack and the output stores of the last maps follow the flag).  LDS-resident map sizes only (nastar_completion_supported);
 * ignored for larger maps, whose status_summary[0] is never set.
 */
int nastar_completion_supported(int H, int W);

/*
 * The reference's BATCH LOOP to the letter (differentiable_astar.py:203-252, :219-225, :251-252), for the class of inputs in which it matters.
 * The reference steps EVERY map until all maps of the batch select their goal in the same step; a finished map keeps its goal on the open
 * list.  The search kernels stop each map at its own goal, which is the same thing iff the goal's own expansion opens nothing that beats the
 * goal (NASTAR_SUMMARY_COUPLED above: always so for g_ratio in [0.5, 1) with costs >= 0).  Otherwise the finished map goes on closing cells
 * while the rest of the batch searches, and its histories / paths -- and the gradient -- depend on when the batch stops.  Two calls on one
 * stream reproduce that exactly, with no host round trip in between:
 *   1. nastar_forward_ex(..., flags | NASTAR_FLAG_MARK_COUPLED, workspace)   the ordinary launch; it marks the maps of the class
 *   2. nastar_forward_batchloop_finish(same buffers, same workspace)         three launches that do nothing when no map is marked:
 *        PROBE  the marked maps in lock-step mode over the whole budget: a bitmap of the steps at which each selects its goal
 *        T_END  the first step at which every map of the batch selects its goal (one workgroup; unmarked maps do so from their goal step on)
 *        FINAL  the marked maps again in lock-step mode for exactly t_end + 1 steps: their rows of histories / paths / sel_log are rewritten,
 *               iters_out[b] = t_end + 1 (so that max(iters) - 1 is the reference's loop index, as always)
 * Unmarked maps are untouched: at a fixed point their outputs do not depend on when the loop stops.  The marks are cheap enough to ask for
 * always; a caller that reads the status summary may skip call 2 when NASTAR_SUMMARY_COUPLED is clear.  Gradients: pass
 * NASTAR_FLAG_LOCKSTEP to nastar_backward_replay* for a log this produced.  Any map size the forward takes.
 * workspace: nastar_batchloop_workspace_bytes(B,H,W,max_iters) bytes, the SAME buffer in both calls.
 */
size_t nastar_batchloop_workspace_bytes(int B, int H, int W, int max_iters);
int nastar_forward_batchloop_finish(const float* cost, const float* start, const float* goal, const float* passable, int B, int H, int W,
                                    double g_ratio, int max_iters, float* histories_out, int64_t* paths_out, int32_t* sel_log_out,
                                    int32_t* iters_out, int32_t* status_out, void* workspace, size_t workspace_bytes, void* stream);
/* Spin (pause loop, at most timeout_us) until *word_host != 0; returns 1 when it is, 0 on timeout.  HOST pointer (pinned memory). */
int nastar_host_wait_nonzero(const volatile int32_t* word_host, int timeout_us);

/*
 * A placement from data the CALLER already has: `levels[b]` = any non-negative integer that grows with the expected length of map b's
 * search -- for the reference's maze data sets the optimal distance of the sampled start cell, |opt_dists[start]|, which every sample
 * carries (utils/data.py:127-134, :200-221).  order_out [B] = the maps sorted by level, largest first (one counting-sort launch, one
 * workgroup; levels are clamped to 4095; maps of equal level in arbitrary order).  A permutation by construction: pass it to
 * nastar_forward_ex without NASTAR_FLAG_CHECK_ORDER.
 */
int nastar_placement_from_levels(const int32_t* levels, int B, int32_t* order_out, void* stream);

/*
 * A placement (the `order` of nastar_forward_ordered) for a batch that has never been searched: maps sorted, longest first, by the
 * level at which a unit-cost 8-connected breadth-first wave from the start reaches the goal over passable cells -- the length of the
 * shortest route (csrc/nastar_placement.hip.h).  A crude predictor of the search length (correlation 0.5-0.8) is as good as the
 * exact step counts here: 4096 mazes 150 -> 115.5 us per search launch (exact: 114.8), 64x64 random maps 273 -> 229.  Two small
 * launches (bit-parallel wave, one wavefront per map; counting sort, one workgroup); 32x32 and 64x64 maps, 16-byte aligned
 * (NASTAR_ERR_UNSUPPORTED otherwise: search without a placement).  Maps of equal level come in arbitrary order.
 *   order_out  [B] int32 device     workspace  >= 4 B bytes of device memory
 */
int nastar_placement_predict(const float* passable, const float* start, const float* goal, int B, int H, int W,
                             int32_t* order_out, void* workspace, size_t workspace_bytes, void* stream);

/*
 * nastar_forward that ALSO emits the bit-packed masks (layout: see nastar_pack_outputs) in the same launch where the
 * shape allows it (W % 4 == 0, H*W % 8 == 0, LDS-resident), otherwise by one extra pack launch on the same stream.
 * This is what one rank of the multi-GPU path calls right before the all-gather.
 */
int nastar_forward_packed(const float* cost, const float* start, const float* goal, const float* passable, int B,
                          int H, int W, double g_ratio, int max_iters, float* histories_out, int64_t* paths_out,
                          int32_t* sel_log_out, int32_t* iters_out, int32_t* status_out, uint8_t* packed_out,
                          void* workspace, size_t workspace_bytes, int flags, void* stream);

/*
 * The replay backward with a PLACEMENT (0.4.1): workgroup i replays map order[i]; everything else as nastar_backward_replay
 * (grad_histories != NULL) or nastar_backward_l1_replay (grad_histories == NULL, histories / opt_trajs / grad_loss_dev given).
 * A map's replay is as long as its search was, and those lengths are known: pass the order_out of the forward launch that wrote
 * sel_log (nastar_forward_ordered) -- 4096 mazes at Tmax 0.25: 187 -> 133 us, bit-identical gradients.  order NULL = identity.
 */
int nastar_backward_replay_ordered(const float* grad_histories, const float* histories, const float* opt_trajs,
                                   const float* grad_loss_dev, const float* cost, const float* start, const float* goal,
                                   const float* passable, const int32_t* sel_log, int B, int H, int W, double g_ratio,
                                   int max_iters, const int32_t* iters, const int32_t* t_batch_dev, float* grad_cost_out,
                                   void* workspace, size_t workspace_bytes, int flags, const int32_t* order, void* stream);

/*
 * Backward of `histories` w.r.t. `cost` (paths carry no gradient):
 *   dL/dcost = sum_t (1-g_ratio) * (-1/sqrt(W)) * y_t * (G_t - <G_t, y_t>)       
 * including the reference's batch-coupled terms: a map that reaches its goal at step tau < t_batch keeps being
 * stepped at its fixed point until the slowest map of the batch finishes (differentiable_astar.py:251), which
 * (i) adds (t_batch - tau) copies of the fixed-point term and (ii) zeroes the upstream gradient of its goal cell
 * (torch.clamp backward at :223).
 *   sel_log        [B,max_iters] int32  sel_log_out of the matching forward (required: the forward's selections ARE the tape)
 *   iters          [B] int32     iters_out of the matching forward
 *   t_batch_dev    device int32* holding t_batch = max(iters)-1 over the WHOLE logical batch (across shards if
 *                  the caller wants single-device semantics), or NULL to treat every map as its own batch
 *                  (t_batch = iters[b]-1: no fixed-point terms).
 *   grad_cost_out  [B,H,W] fp32, fully written
 * By REPLAY of the selection log: no selection is repeated and the softmax is accounted per open-list event, O(9) work per
 * step instead of O(open list) (csrc/nastar_backward_replay.hip.h).  (Rounds 1-3 also exported nastar_backward /
 * nastar_backward_l1, which repeated the selection instead of reading a log; superseded, removed in 0.4.0.)  Any map size the forward takes (1,179,648 cells): the per-map state lives in LDS up to ~11.6 k cells and in
 * the workspace beyond (maps above 65,519 cells, or a history of more than 65535 entries: 32-bit history stamps, 16 instead of 14 B per cell).  workspace: nastar_backward_workspace_bytes(B,H,W,max_iters) bytes (per-step history of the running
 * sums, 16 B per executed step, + the state slabs of maps too large for LDS).  grad_cost_out is fully written.
 * nastar_backward_l1_replay: the fused-L1 form (see nastar_l1_loss below).
 */
size_t nastar_backward_workspace_bytes(int B, int H, int W, int max_iters);
int nastar_backward_replay(const float* grad_histories, const float* cost, const float* start, const float* goal,
                           const float* passable, const int32_t* sel_log, int B, int H, int W, double g_ratio, int max_iters,
                           const int32_t* iters, const int32_t* t_batch_dev, float* grad_cost_out, void* workspace,
                           size_t workspace_bytes, int flags, void* stream);
int nastar_backward_l1_replay(const float* histories, const float* opt_trajs, const float* grad_loss_dev, const float* cost,
                              const float* start, const float* goal, const float* passable, const int32_t* sel_log, int B, int H,
                              int W, double g_ratio, int max_iters, const int32_t* iters, const int32_t* t_batch_dev,
                              float* grad_cost_out, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Training step with the loss fused in (SURVEY.md 8f "next #3"; reference utils/training.py:55-61):
 *   loss = nn.L1Loss()(outputs.histories, opt_trajs); loss.backward()
 * nastar_l1_loss: loss_out[0] = mean |histories - opt_trajs| over numel elements (fixed-order double reduction, bitwise
 *   reproducible); workspace >= 2048 bytes.
 * nastar_backward_l1_replay: nastar_backward_replay with dL/dhistories = (*grad_loss_dev or 1) * sign(histories - opt_trajs) / (B*H*W)
 *   formed inside the kernel (no gradient tensor is materialised).  histories = the forward's own output.
 */
int nastar_l1_loss(const float* histories, const float* opt_trajs, long long numel, float* loss_out, void* workspace,
                   size_t workspace_bytes, void* stream);

/*
 * Dataset path (SURVEY.md 8f "next #4"; reference utils/data.py:171-199 MazeDataset.get_opt_traj, :222-244 next_loc): roll the
 * pre-computed optimal policy out from every start cell until the goal, for a whole batch in one launch.
 *   opt_policies [n_maps, n_actions<=8, H, W] fp32 (one-hot over actions, planning-datasets "moore" action order),
 *   start_idx [n_maps*starts_per_map] / goal_idx [n_maps] int32 flat cell indices,
 *   opt_trajs_out [n_maps, starts_per_map, H, W] fp32 0/1 (start and intermediate cells, NOT the goal; fully written),
 *   status_out [n_maps*starts_per_map]: 0 ok, 1 policy revisits a cell (the reference asserts), 2 leaves the map, 3 no goal in H*W steps.
 */
int nastar_policy_rollout(const float* opt_policies, const int32_t* start_idx, const int32_t* goal_idx, int n_maps,
                          int starts_per_map, int n_actions, int H, int W, float* opt_trajs_out, int32_t* status_out, void* stream);

/* h0 = get_heuristic(goal) for B maps: out [B,H,W] fp32 (parity/debug; the forward computes it on the fly). */
int nastar_heuristic(const float* goal, int B, int H, int W, float* h0_out, void* stream);

/*
 * AstarOutput <-> bit-packed masks: the payload of the multi-GPU collation (one RCCL all-gather of 2 bits per cell
 * instead of the reference's fp32 + int64 = 12 bytes per cell).
 *   packed [B, 2*ceil(H*W/8)] uint8: row b = histories bits then path bits of map b, most significant bit first.
 */
int nastar_pack_outputs(const float* histories, const int64_t* paths, int B, int H, int W, uint8_t* packed_out,
                        void* stream);
int nastar_unpack_outputs(const uint8_t* packed, int B, int H, int W, float* histories_out, int64_t* paths_out,
                          void* stream);

/*
 * CNN cost-map encoder of NeuralAstar, inference only (SURVEY.md section 8f "next #1"; reference planner/encoder.py:60-78,
 * :32-34 and the input assembly of astar.py:171-177), bf16 MFMA with fp32 accumulation:
 *     cost = sigmoid(conv5(relu(bn4(conv4(...relu(bn1(conv1(cat(map, start+goal)))))))))) * final_mul
 *   map/start/goal [B,H,W] fp32 (start/goal may be NULL when plus == 0), cost_out [B,H,W] fp32;
 *   H % 16 == 0 and W % 32 == 0; channels fixed to the reference's depth-4 CNN (2 -> 32 -> 64 -> 128 -> 256 -> 1).
 *   wpack[l]  device bf16 [9][CINp/8][COUTp][8]  (tap = ky*3+kx; CINp = 16,32,64,128,256; COUTp = 32,64,128,256,32; zero padded)
 *   scale[l], shift[l]  device fp32 [COUTp]: eval-mode BatchNorm and conv bias folded: y = acc*scale + shift
 *   wpack/scale/shift themselves are HOST arrays of 5 device pointers.
 *   workspace: nastar_encoder_workspace_bytes(B,H,W) bytes (fewer = more, smaller passes).
 */
size_t nastar_encoder_workspace_bytes(int B, int H, int W);
int nastar_encoder_cnn_forward(const float* map, const float* start, const float* goal, int plus, int B, int H, int W,
                               const uint16_t* const* wpack, const float* const* scale, const float* const* shift,
                               float final_mul, float* cost_out, void* workspace, size_t workspace_bytes, void* stream);
/*
 * The same encoder at fp32-grade accuracy ("f16x3": operands split into two fp16 terms, products hi*hi + lo*hi + hi*lo on the
 * fp16 MFMA, fp32 accumulation; cost maps within 1e-5 of the reference's fp32 encoder).  H, W multiples of 32.
 *   w1_f32   first conv weight [32][1|2][3][3] fp32 (torch layout): that layer runs in plain fp32
 *   wsplit   5 device pointers: layers 2..4 packed over 3*cin virtual channels [W_hi | W_hi | W_lo] as fp16
 *            ([tap][3*cin/8][cout][8]), then the last layer's W_hi and W_lo packs ([tap][256/8][32][8], channel 0 real)
 *   scale/shift as in nastar_encoder_cnn_forward (5 layers); workspace: nastar_encoder_workspace_bytes_f16x3(B,H,W)
 */
size_t nastar_encoder_workspace_bytes_f16x3(int B, int H, int W);
int nastar_encoder_cnn_forward_f16x3(const float* map, const float* start, const float* goal, int plus, int B, int H, int W,
                                     const float* w1_f32, const uint16_t* const* wsplit, const float* const* scale,
                                     const float* const* shift, float final_mul, float* cost_out, void* workspace,
                                     size_t workspace_bytes, void* stream);
/* plain fp16 operands (11 significant bits instead of bf16's 8), same arguments; wpack16 = the five layers packed exactly as for
 * nastar_encoder_cnn_forward but in fp16 (w1_f32 serves sizes other than 32x32); workspace: nastar_encoder_workspace_bytes_f16(B,H,W) */
size_t nastar_encoder_workspace_bytes_f16(int B, int H, int W);
int nastar_encoder_cnn_forward_f16(const float* map, const float* start, const float* goal, int plus, int B, int H, int W,
                                   const float* w1_f32, const uint16_t* const* wpack16, const float* const* scale,
                                   const float* const* shift, float final_mul, float* cost_out, void* workspace,
                                   size_t workspace_bytes, void* stream);
/*
 * CNNDownSize encoder of NeuralAstar in eval mode (reference planner/encoder.py:81-97 + :32-34 and the input assembly of
 * astar.py:171-177; the WarCraft configuration scripts/config/train_warcraft.yaml:6-10 is C = 3, plus = 1, depth = 3, 96x96 -> 12x12)
 * on the f32-input MFMA (fp32 accuracy: csrc/nastar_encoder_downsize.hip.h):
 *     x = cat(image, upsample_nearest(start + goal));  depth x [conv3x3, BN, ReLU, maxpool 2x2];  cost = sigmoid(BN(conv3x3(x))) * final_mul
 *   image [B,C,H,W] fp32 (NCHW, C + plus <= 4), start/goal [B,h,w] fp32 (NULL when plus == 0), cost_out [B, H>>depth, W>>depth] fp32;
 *   H, W multiples of 2^depth.  wts[l] device fp32 [9][CINp][COUTp] (tap = ky*3+kx; CINp = 2|4, 32, 64, 128; COUTp = 32, 64, 128,
 *   256 for the hidden blocks and 32 (channel 0 real) for the last one), scale/shift [COUTp] = eval-mode BatchNorm and conv bias folded;
 *   wts/scale/shift are HOST arrays of depth+1 device pointers.  workspace: nastar_encoder_downsize_workspace_bytes(B, C+plus, H, W, depth).
 */
size_t nastar_encoder_downsize_workspace_bytes(int B, int C, int H, int W, int depth);
int nastar_encoder_cnn_downsize_forward(const float* image, const float* start, const float* goal, int plus, int B, int C, int H,
                                        int W, int h, int w, int depth, const float* const* wts, const float* const* scale,
                                        const float* const* shift, float final_mul, float* cost_out, void* workspace,
                                        size_t workspace_bytes, void* stream);
/* one layer of the above on its own (unit tests): (cin, cout) in {(16,32), (32,64), (64,128), (128,256)} */
int nastar_conv3x3_bf16(const uint16_t* in, const uint16_t* wpack, const float* scale, const float* shift, uint16_t* out,
                        int B, int H, int W, int cin, int cout, int relu, void* stream);

/*
 * Generic 3x3 convolution layer (padding 1) on the fp16 MFMA for any image size and channel count: the building block of the
 * U-Net encoder (reference planner/encoder.py:37-57: segmentation_models_pytorch Unet(vgg16_bn); VGG stages at 32x32 ... 2x2 pixels,
 * decoder blocks = nearest x2 upsampling + skip concatenation + two conv-BN-ReLU) and of layers the fixed-shape kernels above do not
 * cover (csrc/nastar_conv_flat.hip.h).  Activations are NHWC fp16; with NASTAR_CONV_SPLIT every pixel is [hi(C) | lo(C)] (two fp16
 * terms per value, "f16x3": products hi*hi + lo*hi + hi*lo, fp32-grade results).
 *   y[b,y,x,n] = act(scale[n] * sum_{tap,c} w[tap][c][n] * x[b, y+dy, x+dx, c] + shift[n]),  x = cat(in (upsampled), in2)
 *   in   [B, H, W, c1]  (NASTAR_CONV_UPSAMPLE: [B, H/2, W/2, c1], read at (y/2, x/2))      in2  [B, H, W, c2] or NULL (c2 = 0)
 *   wpack  fp16 [9][CINV/8][cout][8], tap = ky*3+kx, CINV = c1+c2 input channels -- or 3*(c1+c2) virtual channels
 *          [W_hi | W_hi | W_lo] with NASTAR_CONV_SPLIT;   scale/shift fp32 [cout] (folded eval-mode BatchNorm / bias)
 *   out  [B, H, W, cout] fp16 (x2 when split), or with NASTAR_CONV_FINAL out_f32 [B,H,W] = sigmoid(y[..., 0]) * final_mul (cout == 32,
 *        channel 0 real: reference encoder.py:32-34)
 *   c1, c2 multiples of 32, cout a multiple of 32, any W (flat 256-pixel tiles up to 126 pixels per row, 64 x 4-pixel tiles of one image
 *   beyond), B*H*W*max(fp16 per pixel) < 2^34 (32 GiB per tensor).
 */
#define NASTAR_CONV_RELU 1
#define NASTAR_CONV_FINAL 2
#define NASTAR_CONV_UPSAMPLE 4
#define NASTAR_CONV_SPLIT 8
#define NASTAR_CONV_RAW 16 /* with NASTAR_CONV_FINAL: out_f32 = y[..., 0] itself (no sigmoid): the training path, BatchNorm follows */
int nastar_conv3x3_f16(const uint16_t* in, const uint16_t* in2, const uint16_t* wpack, const float* scale, const float* shift,
                       uint16_t* out, float* out_f32, int B, int H, int W, int c1, int c2, int cout, int flags, float final_mul,
                       void* stream);
/* The same layer for 32x32 images on the CNN encoder's persistent whole-image kernel (csrc/nastar_encoder.hip.h: ~1.4x the generic
 * kernel's rate): (cin, cout) in {(32,64), (64,128), (128,256), (256,128), (128,64)}, flags: NASTAR_CONV_RELU | NASTAR_CONV_SPLIT;
 * same tensor / weight-pack layouts as nastar_conv3x3_f16.  The training path uses it for its forward and input-gradient convolutions. */
int nastar_conv3x3_img32_f16(const uint16_t* in, const uint16_t* wpack, const float* scale, const float* shift, uint16_t* out, int B,
                             int cin, int cout, int flags, void* stream);
/* 2x2 max-pool of an NHWC fp16 tensor [B,H,W,C] -> [B,H/2,W/2,C] (split: [hi | lo] pairs, the pair with the larger hi + lo wins) */
int nastar_maxpool2x2_f16(const uint16_t* in, uint16_t* out, int B, int H, int W, int C, int split, void* stream);
/* input assembly of NeuralAstar.encode (reference astar.py:171-177): (map, start + goal, 0, ...) as cp-channel NHWC fp16
 * (split: followed by cp zero lo halves); map/start/goal fp32 [npix], start/goal may be NULL when plus == 0 */
int nastar_encoder_prep_f16(const float* map, const float* start, const float* goal, int plus, long long npix, int cp, int split,
                            uint16_t* out, void* stream);

/*
 * Encoder TRAINING kernels (autograd of reference planner/encoder.py:60-78 as utils/training.py:55-61 runs it): together with
 * nastar_conv3x3_f16 (forward, and the input gradient = the same convolution with transposed, flipped weights) they make the
 * conv + batch-statistics BatchNorm + ReLU stack differentiable on the device without torch.nn.  NHWC fp16 tensors, `split` as above.
 *
 * nastar_conv3x3_wgrad_f16: dw[co][ci][ky][kx] = (out_scale / *grad_scale_dev) * sum_{b,y,x} dz[b,y,x,co] * a[b,y+ky-1,x+kx-1,ci]  (zero
 *   padding) on the fp16 MFMA with gfx950's LDS transpose reads (csrc/nastar_conv_wgrad.hip.h).  dz [B,H,W,co], a [B,H,W,ci] with co, ci the
 *   PADDED channel counts (multiples of 32); dw fp32 [co_real][ci_real][3][3] = torch's weight layout, cropped.  grad_scale_dev: device
 *   float holding the power-of-two gradient scale to divide out, or NULL.  Deterministic (per-workgroup partial sums in `workspace`,
 *   nastar_conv3x3_wgrad_workspace_bytes, summed in a fixed order).  W >= 2; a chunk is R whole image rows -- for W > 96 R row SEGMENTS: the widest divisor of W in [64, 96], else equal ragged segments of ceil(W / ceil(W / 96)) pixels, the last one of a row staged with zeros beyond the image -- (R*W <= 96 pixels, 64 for
 *   the power-of-two widths) and H must be a multiple of R (workspace_bytes == 0 flags an unsupported shape); images of <= 48 pixels
 *   (the 4x4 / 2x2 levels of a U-Net) are taken several per chunk, each with its own zero frame.
 * nastar_chan_stats_f16: per-channel sums over all pixels in double: sums[c] = (sum v, sum v^2), or with u != NULL
 *   (sum u*m, sum u*m*v), m = [ms[c]*v + mt[c] > 0] (the ReLU mask).  sums double [C][2], zeroed inside the call; amax_out
 *   (optional device float): max |u*m| over the tensor (feeds the power-of-two gradient re-scaling).
 * nastar_chan_affine_f16: out = k1[c]*u*[ms[c]*v + mt[c] > 0] + k2[c]*v + k3[c], optional ReLU; u == NULL drops the first term
 *   (forward: BatchNorm folded to k2, k3 + ReLU; backward: ReLU mask + closed-form BatchNorm backward).
 */
size_t nastar_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int co, int ci);
int nastar_conv3x3_wgrad_f16(const uint16_t* dz, const uint16_t* a, float* dw, int B, int H, int W, int co, int ci, int co_real,
                             int ci_real, int split, float out_scale, const float* grad_scale_dev, void* workspace,
                             size_t workspace_bytes, void* stream);
int nastar_chan_stats_f16(const uint16_t* u, const uint16_t* v, const float* ms, const float* mt, double* sums, float* amax_out,
                          long long npix, int C, int split, void* stream);
/* The closing block of the CNN / CNNDownSize encoders in TRAINING mode (reference encoder.py:60-97 last block + :32-34): 1-channel
 * BatchNorm with batch statistics + sigmoid * const on z [n] fp32 (raw output of the last convolution), two launches each way:
 *   nastar_bn1_fwd_partial          part[nastar_bn1_parts(n)][2] = per-workgroup (sum z, sum z^2) in double
 *   nastar_bn1_sigmoid_fwd          cost = cmul * sigmoid(gamma (z - mean) invstd + beta); stat_out = (mean, invstd); running statistics
 *                                   updated with `momentum` (unbiased variance) when running_mean != NULL
 *   nastar_bn1_sigmoid_bwd_partial  part[..][3] = (sum dy, sum dy xhat, sum dcost s)
 *   nastar_bn1_sigmoid_bwd          dz (closed-form BatchNorm backward), dgamma, dbeta, dconst (NULL when const is not a parameter)
 * `part` / `nparts` / `n_total`: the kernel adds the partial rows itself in a fixed order; data-parallel training passes the all-reduced
 * global sums as ONE row with n_total = the global element count (encoder_train.SyncBatchNorm).  cmul: device scalar or NULL (= 1). */
int nastar_bn1_parts(long long n);
int nastar_bn1_fwd_partial(const float* z, long long n, double* part, void* stream);
int nastar_bn1_sigmoid_fwd(const float* z, long long n, const double* part, int nparts, double n_total, const float* gamma, const float* beta,
                           double eps, const float* cmul, double momentum, float* running_mean, float* running_var, float* cost_out,
                           double* stat_out, void* stream);
int nastar_bn1_sigmoid_bwd_partial(const float* z, const float* dcost, long long n, const double* stat, const float* gamma, const float* beta,
                                   const float* cmul, double* part, void* stream);
int nastar_bn1_sigmoid_bwd(const float* z, const float* dcost, long long n, const double* stat, const float* gamma, const float* beta,
                           const float* cmul, const double* part, int nparts, double n_total, float* dz_out, float* dgamma_out,
                           float* dbeta_out, float* dconst_out, void* stream);

/* max|w| of n fp32 tensors in ONE launch: table = device array [n][2] of (pointer, element count) as int64; scal [n][3] floats is zeroed
 * and scal[t][2] receives max|w_t| -- the value nastar_pack_conv_weight_f16(..., reuse_max = 1) expects there. */
int nastar_absmax_multi_f32(const long long* table, int n, float* scal, void* stream);

/* One-launch forms of the training step's small work (round 3: at the reference's batch of 100 a U-Net step was 437 launches).
 * nastar_pack_conv_weights_multi_f16: every weight pack of a step at once.  table = device int64 [n][8]: {w, bias or 0, co, ci,
 *   transpose_flip, offset of the pack in flat16 (fp16 elements), offset in flatf (floats; scale[cout_p] then shift[cout_p]), row of
 *   scal}; scal [rows][3] as left by nastar_absmax_multi_f32 (same arithmetic as nastar_pack_conv_weight_f16 with reuse_max); max_tiles =
 *   max over the tensors of ceil(co / 32) * ceil(ci / 32) (a workgroup moves one 32 x 32 x 9 tile through LDS).
 * nastar_rmsprop_multi_f32: torch.optim.RMSprop's plain step (no momentum / centering / weight decay) for n tensors: table = device
 *   int64 [n][4]: {param, grad, square_avg, element count}. */
int nastar_pack_conv_weights_multi_f16(const long long* table, int n, int max_tiles, int split, float* scal, uint16_t* flat16, float* flatf,
                                       void* stream);
int nastar_rmsprop_multi_f32(const long long* table, int n, float lr, float alpha, float eps, void* stream);

/* Two-stage form of nastar_chan_stats_f16 (what the training path uses): per-workgroup partial sums in the caller's workspace
 * (nastar_chan_stats_workspace_bytes), added by a second launch in a fixed order: bitwise reproducible, no zero-fill launches, no fp64
 * atomics, and as many workgroups as the batch allows. */
size_t nastar_chan_stats_workspace_bytes(long long npix, int C);
int nastar_chan_stats_f16_ws(const uint16_t* u, const uint16_t* v, const float* ms, const float* mt, double* sums, float* amax_out,
                             long long npix, int C, int split, void* workspace, size_t workspace_bytes, void* stream);
int nastar_chan_affine_f16(const uint16_t* u, const uint16_t* v, const float* k1, const float* k2, const float* k3, const float* ms,
                           const float* mt, uint16_t* out, long long npix, int C, int relu, int split, void* stream);
/*
 * Device-side glue of a training step (one or two tiny launches each, no host synchronisation; at the reference's batch of 100 maps the
 * step is launch-bound, so nothing between the big kernels is left to framework ops):
 * nastar_pack_conv_weight_f16: torch conv weight w fp32 [co][ci][3][3] -> the wpack of nastar_conv3x3_f16 (channels padded to 32;
 *   transpose_flip = the input-gradient form W'[ci][co][ky][kx] = w[co][ci][2-ky][2-kx]; split = [W_hi | W_hi | W_lo] of w * 2^s with
 *   2^s bringing max|w| to ~2^14), scale_out[cout_p] = 2^-s, shift_out[cout_p] = bias (or 0), scal_out[3] = (2^-s, 2^s, max|w|);
 *   reuse_max != 0: scal_out[2] already holds max|w| (a previous pack of the same weight), the reduction pass is skipped.
 * nastar_bn_coef_fwd: batch sums [C][2] (nastar_chan_stats_f16) -> k2 = gamma*invstd, k3 = beta - mean*k2, mean / invstd (double [C]),
 *   running_mean / running_var updated like nn.BatchNorm2d in training mode (may be NULL).
 * nastar_bn_coef_bwd: backward sums + amax|dy| + forward mean / invstd -> dgamma, dbeta (already divided by the gradient scale
 *   gscale[0]), the closed-form BatchNorm-backward coefficients c1, c2, c3 multiplied by a fresh power of two, gscale[0] updated.
 * nastar_grad_seed_f16: d fp32 [npix] (dL/dz of the 1-channel last block) -> dzb [npix][32 (x2)] fp16, channel 0 = d * S, S = 2^floor(log2(
 *   1024 / max|d|)) written to gscale[0]; amax_scratch: one device float.
 */
int nastar_pack_conv_weight_f16(const float* w, int co, int ci, int transpose_flip, int split, const float* bias, uint16_t* wpack,
                                float* scale_out, float* shift_out, float* scal_out, int reuse_max, void* stream);
/* One BatchNorm pass of the training step in TWO launches instead of three: nastar_chan_stats_f16_ws's partial pass, then ONE kernel that adds the
 * partial rows (same fixed order) and turns each channel's sums into the coefficients of nastar_bn_coef_fwd / nastar_bn_coef_bwd (workspace:
 * nastar_chan_stats_workspace_bytes; sums_out: optional double [C][2]; gscale_out must not alias gscale_in).  Data-parallel training with global statistics needs the sums between
 * the halves (all-reduce) and keeps the separate entry points. */
int nastar_bn_stats_coef_fwd_f16(const uint16_t* z, long long npix, int C, int split, const float* gamma, const float* beta, double eps,
                                 double momentum, float* running_mean, float* running_var, float* k2, float* k3, double* mean_out,
                                 double* invstd_out, double* sums_out, void* workspace, size_t workspace_bytes, void* stream);
int nastar_bn_stats_coef_bwd_f16(const uint16_t* da, const uint16_t* z, const float* ms, const float* mt, long long npix, int C, int split,
                                 const double* mean, const double* invstd, const float* gamma, const float* gscale_in, float* gscale_out,
                                 float* dgamma, float* dbeta, float* c1, float* c2, float* c3, double* sums_out, void* workspace,
                                 size_t workspace_bytes, void* stream);
/* nastar_bn_coef_bwd with the gradient scale read from gscale_in and the re-centred one written to gscale_out (a scale shared by two
 * branches of a U-Net is not overwritten: no copy launch per block) */
int nastar_bn_coef_bwd_io(const double* sums, const float* amax_dy, const double* mean, const double* invstd, const float* gamma,
                          long long npix, const float* gscale_in, float* gscale_out, float* dgamma, float* dbeta, float* c1, float* c2,
                          float* c3, int C, void* stream);
int nastar_bn_coef_fwd(const double* sums, const float* gamma, const float* beta, double eps, long long npix, double momentum,
                       float* running_mean, float* running_var, float* k2, float* k3, double* mean_out, double* invstd_out, int C,
                       void* stream);
int nastar_bn_coef_bwd(const double* sums, const float* amax_dy, const double* mean, const double* invstd, const float* gamma,
                       long long npix, float* gscale, float* dgamma, float* dbeta, float* c1, float* c2, float* c3, int C, void* stream);
int nastar_grad_seed_f16(const float* d, long long npix, int split, uint16_t* dzb, float* gscale, float* amax_scratch, void* stream);
/*
 * The 1-channel closing convolution of the CNN encoders in the training step as STREAMS (csrc/nastar_encoder_co1.hip.h; reference
 * planner/encoder.py:77 `Conv2d(C, 1, 3, padding=1)` under autograd).  Padded to 32 output channels on the matrix cores, its forward,
 * weight gradient and input gradient each spent 31/32 of their work on zeros.  C a power of two, 8 <= C <= 512; `a` = the layer's input
 * [B,H,W,C] fp16 (split: [hi(C) | lo(C)]), `w` = torch's weight [1][C][3][3] fp32, `d` = dL/dz [B,H,W] fp32.
 *   nastar_conv3x3_co1_f16        z [B,H,W] fp32 = conv(a, w) + bias[0]: per-pixel projection onto the 9 taps (every pixel read once)
 *                                 + shifted sum; workspace nastar_conv3x3_co1_workspace_bytes
 *   nastar_conv3x3_co1_wgrad_f16  dw [C][3][3] fp32 = sum_p d[p - off(tap)] a[p][c]  (fixed-order partial rows: bitwise reproducible)
 *   k2 / k3 (both or NULL)        `a` holds the PRE-activation z of the block in front and the layer's input relu(k2[c] z + k3[c]) is
 *                                 formed while loading: that block's activation tensor is never written or read
 *   nastar_grad_scale_f32         gscale[0] = 2^floor(log2(1024 / max|d|)): nastar_grad_seed_f16 without the padded fp16 tensor
 *   nastar_bn_stats_coef_bwd_u1_f16 / nastar_chan_affine_u1_f16
 *                                 nastar_bn_stats_coef_bwd_f16 / nastar_chan_affine_f16 (backward form) for the block in FRONT of the
 *                                 closing convolution with da = gscale * (input gradient of d) formed on the fly from d and w: da is
 *                                 never stored, and the two passes do not read it.
 */
size_t nastar_conv3x3_co1_workspace_bytes(int B, int H, int W, int C);
int nastar_conv3x3_co1_f16(const uint16_t* a, const float* w, const float* bias, int B, int H, int W, int C, int split, const float* k2,
                           const float* k3, float* z_out, void* workspace, size_t workspace_bytes, void* stream);
int nastar_conv3x3_co1_wgrad_f16(const float* d, const uint16_t* a, int B, int H, int W, int C, int split, const float* k2, const float* k3,
                                 float* dw_out, void* workspace, size_t workspace_bytes, void* stream);
int nastar_grad_scale_f32(const float* d, long long npix, float* gscale, float* amax_scratch, void* stream);
int nastar_bn_stats_coef_bwd_u1_f16(const float* d, const float* wlast, int B, int H, int W, const uint16_t* z, const float* ms,
                                    const float* mt, int C, int split, const double* mean, const double* invstd, const float* gamma,
                                    const float* gscale_in, float* gscale_out, float* dgamma, float* dbeta, float* c1, float* c2,
                                    float* c3, double* sums_out, void* workspace, size_t workspace_bytes, void* stream);
int nastar_chan_stats_u1_f16_ws(const float* d, const float* wlast, const float* gscale, int B, int H, int W, const uint16_t* v,
                                const float* ms, const float* mt, double* sums, float* amax_out, int C, int split, void* workspace,
                                size_t workspace_bytes, void* stream);  /* the same sums as a separate entry (sync-BN all-reduces them) */
int nastar_chan_affine_u1_f16(const float* d, const float* wlast, const float* gscale, int B, int H, int W, const uint16_t* z,
                              const float* k1, const float* k2, const float* k3, const float* ms, const float* mt, uint16_t* out, int C,
                              int split, void* stream);
/* 2x2 max-pool backward (CNNDownSize blocks, reference encoder.py:91-95 under autograd): r [B,H,W,C] the pool's input, dp [B,H/2,W/2,C]
 * the gradient w.r.t. its output -> dr [B,H,W,C]: dp at each window's FIRST maximum (torch's tie rule), 0 elsewhere. */
int nastar_maxpool2x2_bwd_f16(const uint16_t* r, const uint16_t* dp, uint16_t* dr, int B, int H, int W, int C, int split, void* stream);
/*
 * U-Net decoder plumbing under autograd (reference encoder.py:37-57: nearest x2 upsampling + skip concatenation in front of a conv):
 * nastar_upcat_f16: out [B,H,W,c1+c2] = cat(x [B,H/2,W/2,c1] read at (y/2, x/2), skip [B,H,W,c2]) -- materialised for the weight gradient
 *   (the convolution itself gathers, NASTAR_CONV_UPSAMPLE);   nastar_upcat_bwd_f16: dx [B,H/2,W/2,c1] = 2x2 block sums of dcat[..., :c1],
 *   dskip [B,H,W,c2] = dcat[..., c1:].
 * nastar_grad_add_f16: two gradients of one tensor carried with different power-of-two scales (device floats):
 *   out = a * (So/Sa) + b * (So/Sb), *scale_out = So = min(Sa, Sb)   (a skip feature collects a decoder and an encoder gradient).
 */
int nastar_upcat_f16(const uint16_t* x, const uint16_t* skip, uint16_t* out, int B, int H, int W, int c1, int c2, int split, void* stream);
int nastar_upcat_bwd_f16(const uint16_t* dcat, uint16_t* dx, uint16_t* dskip, int B, int H, int W, int c1, int c2, int split, void* stream);
int nastar_grad_add_f16(const uint16_t* a, const float* scale_a, const uint16_t* b, const float* scale_b, uint16_t* out, float* scale_out,
                        long long npix, int C, int split, void* stream);

/* Resident forward workgroups (= maps) per CU the runtime reports for an HxW map, and the LDS bytes one map takes
 * (diagnostics for DESIGN.md / bench.py; returns -1 on error, 0 if the size is unsupported). */
int nastar_debug_occupancy(int H, int W, int* lds_bytes_out);

#ifdef __cplusplus
}
#endif
#endif /* NASTAR_H_ */
