// nastar_train_capi.hip -- training-harness kernels either side of the search (include/nastar.h): the dataset's optimal-trajectory
// roll-out and the L1 loss reduction.  (nastar_backward_l1 lives with the backward kernels in nastar_capi.hip.)
#include <hip/hip_runtime.h>
#include <math.h>

#include "nastar_host.hip.h"

namespace nastar {

// ---- optimal-trajectory roll-out of the dataset path (reference utils/data.py:171-199 get_opt_traj + :222-244 next_loc) --------------
// One thread per roll-out (map n, start s): follow argmax_a policy[n][a][cell] from the start cell until the goal cell; every visited
// cell except the goal is set to 1.  A serial chain of at most H*W dependent 8-way loads; the batch supplies the parallelism.
// status: 0 ok, 1 = the policy revisits a cell (the reference asserts), 2 = it walks off the map / start invalid, 3 = no goal within H*W.
__global__ __launch_bounds__(64) void nastar_policy_rollout_kernel(const float* pol, const int* start_idx, const int* goal_idx,
                                                                  int n_roll, int starts_per_map, int A, int H, int W,
                                                                  float* traj, int* status)
{
    const int HW = H * W;
    const int r0 = blockIdx.x * 64;
    const int nr = (n_roll - r0 < 64) ? n_roll - r0 : 64;
    for (long long i = threadIdx.x; i < (long long)nr * HW; i += 64) traj[(size_t)r0 * HW + i] = 0.f;
    __syncthreads();
    const int r = r0 + threadIdx.x;
    if (r >= n_roll) return;
    const int n = r / starts_per_map;
    const float* p = pol + (size_t)n * A * HW;
    float* t = traj + (size_t)r * HW;
    const int goal = goal_idx[n];
    int cur = start_idx[r];
    int st = 0;
    if ((unsigned)cur >= (unsigned)HW || (unsigned)goal >= (unsigned)HW) st = 2;
    int steps = 0;
    while (st == 0 && cur != goal) {
        t[cur] = 1.0f;                                   // :190
        int best = 0;
        float bv = p[cur];
        for (int a = 1; a < A; ++a) {                    // np.argmax: first maximum (:243)
            const float v = p[(size_t)a * HW + cur];
            if (v > bv) { bv = v; best = a; }
        }
        // action -> (dy, dx), :232-241
        const int dy = (best == 0 || best == 4 || best == 5) ? -1 : ((best == 3 || best == 6 || best == 7) ? 1 : 0);
        const int dx = (best == 1 || best == 4 || best == 6) ? 1 : ((best == 2 || best == 5 || best == 7) ? -1 : 0);
        const int y = cur / W + dy, x = cur % W + dx;
        if (best > 7 || (unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) { st = 2; break; }
        const int nxt = y * W + x;
        if (t[nxt] != 0.f) { st = 1; break; }            // :193-195
        cur = nxt;
        if (++steps > HW) st = 3;
    }
    status[r] = st;
}

// ---- mean |histories - opt_trajs| (nn.L1Loss, training.py:58): fixed-order two-stage reduction in double, deterministic --------
constexpr int kL1Blocks = 256;
__device__ __forceinline__ double block_sum_256(double v, double* sh)
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    return sh[0];
}
__global__ __launch_bounds__(256) void nastar_l1_partial_kernel(const float* h, const float* t, long long n, double* part)
{
    __shared__ double sh[256];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)kL1Blocks * 256) acc += (double)fabsf(h[i] - t[i]);
    const double tot = block_sum_256(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void nastar_l1_final_kernel(const double* part, long long n, float* loss)
{
    __shared__ double sh[256];
    const double tot = block_sum_256(part[threadIdx.x], sh);
    if (threadIdx.x == 0) loss[0] = (float)(tot / (double)n);
}

}  // namespace nastar

using namespace nastar;

extern "C" {

int nastar_policy_rollout(const float* opt_policies, const int32_t* start_idx, const int32_t* goal_idx, int n_maps,
                          int starts_per_map, int n_actions, int H, int W, float* opt_trajs_out, int32_t* status_out, void* stream)
{
    if (!opt_policies || !start_idx || !goal_idx || !opt_trajs_out || !status_out) return NASTAR_ERR_NULL;
    if (n_maps <= 0 || starts_per_map <= 0 || H <= 0 || W <= 0 || n_actions <= 0 || n_actions > 8) return NASTAR_ERR_BAD_SHAPE;
    const long long n_roll = (long long)n_maps * starts_per_map;
    if (n_roll > (1ll << 30)) return NASTAR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(nastar_policy_rollout_kernel, dim3((unsigned)((n_roll + 63) / 64)), dim3(64), 0,
                       reinterpret_cast<hipStream_t>(stream), opt_policies, start_idx, goal_idx, (int)n_roll, starts_per_map,
                       n_actions, H, W, opt_trajs_out, status_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_l1_loss(const float* histories, const float* opt_trajs, long long numel, float* loss_out, void* workspace,
                   size_t workspace_bytes, void* stream)
{
    if (!histories || !opt_trajs || !loss_out || !workspace) return NASTAR_ERR_NULL;
    if (numel <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (workspace_bytes < (size_t)kL1Blocks * sizeof(double)) return NASTAR_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    double* part = static_cast<double*>(workspace);
    hipLaunchKernelGGL(nastar_l1_partial_kernel, dim3(kL1Blocks), dim3(256), 0, s, histories, opt_trajs, numel, part);
    hipLaunchKernelGGL(nastar_l1_final_kernel, dim3(1), dim3(256), 0, s, part, numel, loss_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

}  // extern "C"
