/*
 * nastar_oracle.c -- CPU restatement of the reference's DifferentiableAstar hot path.
 *
 * *** TEST INFRASTRUCTURE ONLY. ***  Nothing under neural-astar_amd/ (the product) may import,
 * link or call this file.  Its only users are tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py, where it is the CHECKER (or the timed CPU port), never the
 * thing shipped.
 *
 * Parity status: PINNED.  tests/golden/ holds outputs of the reference itself (generated in the
 * authoring container by oracle/gen_golden.py, which imports
 * /root/reference/src/neural_astar/planner/differentiable_astar.py by file path); the oracle is
 * checked against every one of them plus the known-answer table of SURVEY.md section 8(c).
 *
 * Two restatements live here:
 *
 *  (1) nastar_oracle_forward_dense / nastar_oracle_backward_dense
 *      A LITERAL per-line restatement of the reference's tensor program: every [H,W] fp32 map the
 *      reference keeps (open, histories, g, h, parents-as-float) is kept as an fp32 array, every
 *      elementwise op is done in the reference's order with one fp32 rounding per ATen op, the
 *      selection is the first arg-max of exp(-f/sqrt(W))*open / sum, expand() is the dense 3x3
 *      zero-centre stencil, the batch runs until EVERY map selected its goal in the same step and
 *      backtrack() walks exactly `t` steps.  (reference: differentiable_astar.py:150-267)
 *
 *  (2) nastar_oracle_forward_sm
 *      The integer "state machine" reading of the same algorithm: first-index
 *      arg-min over open cells of q = fl(f / fl32(sqrt(W))) (the IEEE division of :207 merges f values
 *      one ulp apart into exact ties, so ordering by f itself is NOT equivalent), <=8 neighbour
 *      updates, per-map early exit, walk-to-start backtrack.  This is the algorithm the HIP kernel implements; the CPU tests prove (1)==(2)
 *      on the golden vectors so that kernel-vs-(1) failures can be told apart from algorithmic ones.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_ERR_ARG 1
#define ORACLE_ERR_UNSOLVABLE 3 /* open list ran empty: the reference would NaN + IndexError */
#define ORACLE_ERR_ALLOC 5

/* ---- get_heuristic (differentiable_astar.py:26-52) -------------------------------------- */
/* h0 = fl( fl(sum - min) + fl( fl32(tb) * fl(sqrt(dr^2+dc^2)) ) ), all fp32.                */
static float heuristic0(int r, int c, int gr, int gc, float tb)
{
    float dr = fabsf((float)r - (float)gr);          /* torch.abs(loc_expand - goal_loc_expand)  :47 */
    float dc = fabsf((float)c - (float)gc);
    float sum = dr + dc;                              /* dxdy.sum(dim=1)                          :48 */
    float mn = dr < dc ? dr : dc;                     /* dxdy.min(dim=1)[0]                       :48 */
    float h = sum - mn;
    float a = (float)r - (float)gr, b = (float)c - (float)gc;
    float sq = a * a + b * b;                         /* ((loc - goal) ** 2).sum(1)               :49 */
    float euc = sqrtf(sq);
    float t = tb * euc;                               /* tb_factor * euc  (python scalar -> fp32) :50 */
    return h + t;
}

/* dense 3x3 stencil, zero centre, zero padding == F.conv2d(x, neighbor_filter, padding=1)
 * (differentiable_astar.py:77-93, filter at :140-141).  Accumulation order = kernel raster order. */
static void expand_dense(const float* x, float* y, int H, int W)
{
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            float acc = 0.0f;
            for (int dr = -1; dr <= 1; ++dr)
                for (int dc = -1; dc <= 1; ++dc) {
                    if (dr == 0 && dc == 0) continue;
                    int rr = r + dr, cc = c + dc;
                    if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
                    acc += x[rr * W + cc];
                }
            y[r * W + c] = acc;
        }
}

typedef struct {
    float *open, *hist, *g, *h, *par, *f, *fe, *y, *sel, *nb, *g2, *idx, *tmp;
    int goal_idx;
    int solved_step; /* first step at which this map selected its goal, -1 if never */
} dense_map_t;

static int dense_alloc(dense_map_t* m, int HW)
{
    float* p = (float*)calloc((size_t)13 * HW, sizeof(float));
    if (!p) return 0;
    m->open = p; m->hist = p + HW; m->g = p + 2 * HW; m->h = p + 3 * HW; m->par = p + 4 * HW;
    m->f = p + 5 * HW; m->fe = p + 6 * HW; m->y = p + 7 * HW; m->sel = p + 8 * HW;
    m->nb = p + 9 * HW; m->g2 = p + 10 * HW; m->idx = p + 11 * HW; m->tmp = p + 12 * HW;
    return 1;
}

static float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }

/* One loop iteration (differentiable_astar.py:203-252) for one map.  Returns is_unsolved (0/1)
 * or -1 when the open list is empty or the exponentials overflow (sum == 0 or inf -> NaN in the reference).  If y_out != NULL the
 * softmax y_t (needed by the backward) is copied there; *pass_goal gets the clamp-backward mask. */
static int dense_step(dense_map_t* m, const float* cost, const float* goal, const float* passable,
                      int H, int W, float gr, float omg, float sqrtW, int* sel_idx,
                      float* y_out, unsigned char* passmask_out)
{
    const int HW = H * W;
    /* :206  f = g_ratio * g + (1 - g_ratio) * h */
    for (int i = 0; i < HW; ++i) {
        float a = gr * m->g[i];
        float b = omg * m->h[i];
        m->f[i] = a + b;
    }
    /* :207-208  f_exp = exp(-1 * f / sqrt(W)) * open_maps */
    float s = 0.0f;
    for (int i = 0; i < HW; ++i) {
        float nf = -1.0f * m->f[i];
        float q = nf / sqrtW;
        float e = expf(q);
        m->fe[i] = e * m->open[i];
        s += m->fe[i]; /* :67-68 val_.sum(dim=-1) (summation order differs from ATen's; only ordering of y matters) */
    }
    if (!(s > 0.0f) || isinf(s)) return -1; /* empty open list (0 / 0), or exp overflowed (f / sqrt(W) < -88.7: costs below zero summed over a long
                                             * route): inf / inf -- the reference's softmax is NaN either way and its arg-max meaningless */
    /* :68-69  y = val / sum ; _, ind = y.max(dim=-1)  (first maximal index) */
    int ind = 0;
    float best = -1.0f;
    for (int i = 0; i < HW; ++i) {
        m->y[i] = m->fe[i] / s;
        if (m->y[i] > best) { best = m->y[i]; ind = i; }
    }
    if (y_out) memcpy(y_out, m->y, sizeof(float) * HW);
    /* :70-74  y_hard one-hot; forward value of (y_hard - y).detach() + y is exactly one-hot */
    for (int i = 0; i < HW; ++i) m->sel[i] = 0.0f;
    m->sel[ind] = 1.0f;
    *sel_idx = ind;
    /* :219-220 dist_to_goal, is_unsolved */
    float dist = 0.0f;
    for (int i = 0; i < HW; ++i) dist += m->sel[i] * goal[i];
    float unsolved = (dist < 1e-8f) ? 1.0f : 0.0f;
    /* :222-225 histories / open update */
    for (int i = 0; i < HW; ++i) {
        float hs = m->hist[i] + m->sel[i];
        if (passmask_out) passmask_out[i] = (hs >= 0.0f && hs <= 1.0f); /* clamp backward passes inside [min,max] */
        m->hist[i] = clamp01(hs);
        m->open[i] = clamp01(m->open[i] - unsolved * m->sel[i]);
    }
    /* :228-229 neighbor_nodes = expand(sel) * obstacles_maps */
    expand_dense(m->sel, m->nb, H, W);
    for (int i = 0; i < HW; ++i) m->nb[i] = m->nb[i] * passable[i];
    /* :234 g2 = expand((g + cost) * sel) */
    for (int i = 0; i < HW; ++i) m->tmp[i] = (m->g[i] + cost[i]) * m->sel[i];
    expand_dense(m->tmp, m->g2, H, W);
    /* :235-243 idx, g, open */
    for (int i = 0; i < HW; ++i) {
        float gt = (m->g[i] > m->g2[i]) ? 1.0f : 0.0f;
        float idx = (1.0f - m->open[i]) * (1.0f - m->hist[i]) + m->open[i] * gt;
        idx = idx * m->nb[i];
        m->idx[i] = idx;
        m->g[i] = m->g2[i] * idx + m->g[i] * (1.0f - idx);
        m->open[i] = clamp01(m->open[i] + idx);
    }
    /* :246-249 parents = new_parents * idx + parents * (1 - idx)  (fp32 holding flat indices) */
    for (int i = 0; i < HW; ++i)
        m->par[i] = (float)ind * m->idx[i] + m->par[i] * (1.0f - m->idx[i]);
    return unsolved != 0.0f;
}

static int first_argmax(const float* x, int n)
{
    int ind = 0;
    float best = x[0];
    for (int i = 1; i < n; ++i)
        if (x[i] > best) { best = x[i]; ind = i; }
    return ind;
}

static void dense_init(dense_map_t* m, const float* cost, const float* start, const float* goal,
                       int H, int W)
{
    const int HW = H * W;
    /* :40-44 goal location through the einsum with the meshgrid (exact for a one-hot) */
    float grf = 0.0f, gcf = 0.0f;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            grf += (float)r * goal[r * W + c];
            gcf += (float)c * goal[r * W + c];
        }
    m->goal_idx = first_argmax(goal, HW); /* :197 goal_maps.max(-1)[-1] */
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            int i = r * W + c;
            float dr = fabsf((float)r - grf), dc = fabsf((float)c - gcf);
            float hh = (dr + dc) - (dr < dc ? dr : dc);
            float a = (float)r - grf, b = (float)c - gcf;
            float euc = sqrtf(a * a + b * b);
            float h0 = hh + 0.001f * euc;
            m->open[i] = start[i];                 /* :187 */
            m->hist[i] = 0.0f;                     /* :188 */
            m->h[i] = h0 + cost[i];                /* :191-192 */
            m->g[i] = 0.0f;                        /* :193 */
            m->par[i] = (float)m->goal_idx;        /* :195-198 */
        }
    m->solved_step = -1;
    (void)heuristic0;
}

/* backtrack (differentiable_astar.py:96-125): exactly `t` steps. */
static void backtrack_dense(const float* goal, const float* par, int HW, int t, int64_t* path)
{
    for (int i = 0; i < HW; ++i) path[i] = (int64_t)goal[i];       /* :117 */
    int64_t loc = 0;
    for (int i = 0; i < HW; ++i) loc += (int64_t)par[i] * (int64_t)goal[i]; /* :121 */
    for (int k = 0; k < t; ++k) {                                    /* :122-124 */
        path[loc] = 1;
        loc = (int64_t)par[loc];
    }
}

/*
 * Literal batch forward.  Layouts: all maps [B,H,W] contiguous fp32; histories [B,H,W] fp32;
 * paths [B,H,W] int64; sel_log [B,max_iters] int32 (selected flat index per executed step, -1
 * beyond t_batch) or NULL; iters_out [B] = (step at which the map first selected its goal)+1,
 * or the number of executed steps if it never did; t_batch_out = last executed loop index `t`.
 * max_iters = int(Tmax_eff * W * W) (differentiable_astar.py:200-202).
 */
int nastar_oracle_forward_dense(const float* cost, const float* start, const float* goal,
                                const float* passable, int B, int H, int W, double g_ratio,
                                int max_iters, float* histories, int64_t* paths, int32_t* sel_log,
                                int32_t* iters_out, int32_t* t_batch_out)
{
    if (B <= 0 || H <= 0 || W <= 0 || max_iters <= 0) return ORACLE_ERR_ARG;
    const int HW = H * W;
    const float gr = (float)g_ratio;
    const float omg = (float)(1.0 - g_ratio);   /* python evaluates (1 - g_ratio) in double      :206 */
    const float sqrtW = (float)sqrt((double)W); /* math.sqrt(W) -> fp32 scalar for the division  :207 */
    dense_map_t* maps = (dense_map_t*)calloc((size_t)B, sizeof(dense_map_t));
    if (!maps) return ORACLE_ERR_ALLOC;
    int rc = ORACLE_OK;
    for (int b = 0; b < B; ++b) {
        if (!dense_alloc(&maps[b], HW)) { rc = ORACLE_ERR_ALLOC; goto done; }
        dense_init(&maps[b], cost + (size_t)b * HW, start + (size_t)b * HW, goal + (size_t)b * HW, H, W);
    }
    if (sel_log) for (size_t i = 0; i < (size_t)B * max_iters; ++i) sel_log[i] = -1;
    int t = 0;
    for (t = 0; t < max_iters; ++t) {
        int all_solved = 1, bad = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(&& : all_solved) reduction(|| : bad)
        for (int b = 0; b < B; ++b) {
            int ind = -1;
            int u = dense_step(&maps[b], cost + (size_t)b * HW, goal + (size_t)b * HW,
                               passable + (size_t)b * HW, H, W, gr, omg, sqrtW, &ind, NULL, NULL);
            if (u < 0) { bad = 1; continue; }
            if (sel_log) sel_log[(size_t)b * max_iters + t] = ind;
            if (!u && maps[b].solved_step < 0) maps[b].solved_step = t;
            all_solved = all_solved && !u;
        }
        if (bad) { rc = ORACLE_ERR_UNSOLVABLE; goto done; }
        if (all_solved) break;                                   /* :251-252 */
    }
    if (t == max_iters) t = max_iters - 1;                        /* python `t` after an exhausted range */
    for (int b = 0; b < B; ++b) {
        memcpy(histories + (size_t)b * HW, maps[b].hist, sizeof(float) * HW);
        backtrack_dense(goal + (size_t)b * HW, maps[b].par, HW, t, paths + (size_t)b * HW); /* :255 */
        if (iters_out) iters_out[b] = maps[b].solved_step >= 0 ? maps[b].solved_step + 1 : t + 1;
    }
    if (t_batch_out) *t_batch_out = t;
done:
    for (int b = 0; b < B; ++b) free(maps[b].open);
    free(maps);
    return rc;
}

/*
 * Literal reverse-mode restatement of what autograd does for `histories` -> `cost_maps`
 * (SURVEY.md 8a-8).  Per map: re-run the forward keeping y_t and the clamp pass-mask of every
 * executed step (0..t_batch), then walk the tape backwards:
 *     G_hist(T) = grad_histories ; for t = T-1..0:  G = G_hist(t+1) * passmask_t ;
 *     grad_cost += (1-g_ratio) * (-1/sqrt(W)) * y_t * (G - <G, y_t>) ;  G_hist(t) = G
 * Accumulations are done in double; the reference's own fp32 accumulation order is ATen's.
 */
int nastar_oracle_backward_dense(const float* grad_hist, const float* cost, const float* start,
                                 const float* goal, const float* passable, int B, int H, int W,
                                 double g_ratio, int max_iters, float* grad_cost)
{
    if (B <= 0 || H <= 0 || W <= 0 || max_iters <= 0) return ORACLE_ERR_ARG;
    const int HW = H * W;
    const float gr = (float)g_ratio;
    const float omg = (float)(1.0 - g_ratio);
    const float sqrtW = (float)sqrt((double)W);
    /* pass 1: find t_batch with the plain forward */
    int32_t t_batch = 0;
    {
        float* hist = (float*)malloc(sizeof(float) * (size_t)B * HW);
        int64_t* paths = (int64_t*)malloc(sizeof(int64_t) * (size_t)B * HW);
        if (!hist || !paths) { free(hist); free(paths); return ORACLE_ERR_ALLOC; }
        int rc = nastar_oracle_forward_dense(cost, start, goal, passable, B, H, W, g_ratio, max_iters,
                                             hist, paths, NULL, NULL, &t_batch);
        free(hist); free(paths);
        if (rc) return rc;
    }
    const int T = t_batch + 1;
    int rc_all = ORACLE_OK;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        dense_map_t m;
        float* ys = (float*)malloc(sizeof(float) * (size_t)T * HW);
        unsigned char* pm = (unsigned char*)malloc((size_t)T * HW);
        double* G = (double*)malloc(sizeof(double) * HW);
        double* acc = (double*)calloc(HW, sizeof(double));
        if (!dense_alloc(&m, HW) || !ys || !pm || !G || !acc) { rc_all = ORACLE_ERR_ALLOC; continue; }
        const float* cb = cost + (size_t)b * HW;
        dense_init(&m, cb, start + (size_t)b * HW, goal + (size_t)b * HW, H, W);
        for (int t = 0; t < T; ++t) {
            int ind;
            dense_step(&m, cb, goal + (size_t)b * HW, passable + (size_t)b * HW, H, W, gr, omg, sqrtW,
                       &ind, ys + (size_t)t * HW, pm + (size_t)t * HW);
        }
        for (int i = 0; i < HW; ++i) G[i] = grad_hist[(size_t)b * HW + i];
        const double k = (double)omg * (-1.0 / (double)sqrtW);
        for (int t = T - 1; t >= 0; --t) {
            const float* y = ys + (size_t)t * HW;
            const unsigned char* p = pm + (size_t)t * HW;
            double dot = 0.0;
            for (int i = 0; i < HW; ++i) { if (!p[i]) G[i] = 0.0; dot += G[i] * (double)y[i]; }
            for (int i = 0; i < HW; ++i) acc[i] += k * (double)y[i] * (G[i] - dot);
        }
        for (int i = 0; i < HW; ++i) grad_cost[(size_t)b * HW + i] = (float)acc[i];
        free(m.open); free(ys); free(pm); free(G); free(acc);
    }
    return rc_all;
}

/* ------------------------------------------------------------------------------------------ */
/* (2) state-machine reading (SURVEY.md 8a "Forward state machine"), one map at a time.        */
/* ------------------------------------------------------------------------------------------ */
int nastar_oracle_forward_sm(const float* cost, const float* start, const float* goal,
                             const float* passable, int B, int H, int W, double g_ratio,
                             int max_iters, float* histories, int64_t* paths, int32_t* sel_log,
                             int32_t* iters_out, int32_t* status_out)
{
    if (B <= 0 || H <= 0 || W <= 0 || max_iters <= 0) return ORACLE_ERR_ARG;
    const int HW = H * W;
    const float gr = (float)g_ratio;
    const float omg = (float)(1.0 - g_ratio);
    const float sqrtW = (float)sqrt((double)W);
    int any_unsolvable = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(|| : any_unsolvable)
    for (int b = 0; b < B; ++b) {
        const float* cb = cost + (size_t)b * HW;
        const float* pb = passable + (size_t)b * HW;
        float* g = (float*)calloc(HW, sizeof(float));
        float* hh = (float*)malloc(sizeof(float) * HW);
        float* key = (float*)malloc(sizeof(float) * HW);
        int32_t* par = (int32_t*)malloc(sizeof(int32_t) * HW);
        unsigned char* st = (unsigned char*)calloc(HW, 1); /* bit0 open, bit1 closed */
        int s_idx = first_argmax(start + (size_t)b * HW, HW);
        int g_idx = first_argmax(goal + (size_t)b * HW, HW);
        int gr_ = g_idx / W, gc_ = g_idx % W;
        for (int i = 0; i < HW; ++i) {
            float h = heuristic0(i / W, i % W, gr_, gc_, 0.001f) + cb[i];
            hh[i] = omg * h;
            par[i] = g_idx;
        }
        st[s_idx] = 1;
        key[s_idx] = (gr * 0.0f + hh[s_idx]) / sqrtW;
        int iters = 0, solved = 0, empty = 0;
        if (sel_log) for (int t = 0; t < max_iters; ++t) sel_log[(size_t)b * max_iters + t] = -1;
        while (iters < max_iters) {
            int s = -1;
            float best = 0.0f;
            for (int i = 0; i < HW; ++i)
                if ((st[i] & 1) && (s < 0 || key[i] < best)) { best = key[i]; s = i; }
            if (s < 0) { empty = 1; break; }
            if (sel_log) sel_log[(size_t)b * max_iters + iters] = s;
            ++iters;
            st[s] |= 2;
            if (s == g_idx) { solved = 1; break; } /* fixed point from here on */
            st[s] &= ~1;
            float g2 = g[s] + cb[s];
            int r = s / W, c = s % W;
            for (int dr = -1; dr <= 1; ++dr)
                for (int dc = -1; dc <= 1; ++dc) {
                    if (!dr && !dc) continue;
                    int rr = r + dr, cc = c + dc;
                    if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
                    int n = rr * W + cc;
                    if (pb[n] == 0.0f || (st[n] & 2)) continue;
                    if ((st[n] & 1) && !(g[n] > g2)) continue;
                    g[n] = g2;
                    key[n] = (gr * g2 + hh[n]) / sqrtW;
                    st[n] |= 1;
                    par[n] = s;
                }
        }
        float* hb = histories + (size_t)b * HW;
        int64_t* pth = paths + (size_t)b * HW;
        for (int i = 0; i < HW; ++i) { hb[i] = (st[i] & 2) ? 1.0f : 0.0f; pth[i] = 0; }
        /* backtrack: goal, then parents until the start; capped at (iters-1) steps when the budget
         * ran out (differentiable_astar.py:255 passes t = max_iters-1 in that case). */
        pth[g_idx] = 1;
        int loc = par[g_idx];
        int cap = solved ? HW : iters - 1;
        for (int k = 0; k < cap; ++k) {
            pth[loc] = 1;
            if (loc == s_idx || loc == g_idx) break;
            loc = par[loc];
        }
        if (iters_out) iters_out[b] = iters;
        if (status_out) status_out[b] = empty ? ORACLE_ERR_UNSOLVABLE : 0;
        any_unsolvable = any_unsolvable || empty;
        free(g); free(hh); free(key); free(par); free(st);
    }
    return any_unsolvable ? ORACLE_ERR_UNSOLVABLE : ORACLE_OK;
}

/* heuristic table for spot checks against SURVEY.md 8(c) known values */
void nastar_oracle_heuristic(int H, int W, int goal_r, int goal_c, float* out)
{
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) out[r * W + c] = heuristic0(r, c, goal_r, goal_c, 0.001f);
}
