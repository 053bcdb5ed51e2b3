// nastar_fastlane.cpp -- the host side of ONE checked search call in native code (lib/_nastar_fastlane.so, a CPython extension).
//
// `DifferentiableAstar.forward()` of the reference is a Python loop with a device->host sync per iteration (differentiable_astar.py:203-252).
// Its replacement is one kernel launch of ~0.12 ms for 4096 maps -- short enough that the Python around the launch showed: in round 5 a
// checked `VanillaAstar.forward()` cost 0.155 ms against 0.123 ms for the bare launch (two nn.Module.__call__, three torch.empty, eight
// data_ptr(), one 23-argument ctypes call, the status-board bookkeeping).  This file is that host work in C++, behind the SAME C ABI:
//
//   search(fn, cost, start, goal, passable | None, g_ratio, max_iters, want_log, flags, order | None, order_out | None, ws_bytes,
//          summary_host, counter_dev, stream, spin_us, levels | None, sort_fn, packed | None)
//     0. (a batch that carries its loader's levels instead of an order: launches nastar_placement_from_levels in front of the search)
//     1. allocates the AstarOutput tensors in their final layout ([B,1,H,W] fp32 / int64, iters + status as one [2,B] int32 block) through ATen
//        (the caching allocator; no torch types cross the C ABI below),
//        (`packed`: the caller's uint8 slot for the bit-packed masks, 2 bits per cell, which the search launch then emits itself),
//     2. calls nastar_forward_ex -- `fn` is its address, taken from the ctypes handle of libnastar_hip.so: this extension links against neither
//        HIP nor the kernel library,
//     3. with the GIL released, polls the launch's completion flag in pinned host memory (include/nastar.h: completion_counter /
//        status_summary[0]) for at most spin_us, and
//     4. returns (histories, paths, iters, status, sel_log | None, rc, verdict): verdict = -1 when the flag was not seen (the caller waits for
//        the stream instead), otherwise a bit mask of the summary cells 1..15 that are set (0 = every map ended with status 0); the row is
//        zeroed for its next user.
// The torch.library custom ops stay for autograd / torch.compile / fake tensors (neural_astar/ops.py); this is the lane a no-grad call takes.
#include <torch/extension.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>

namespace {

using sort_t = int (*)(const int32_t*, int, int32_t*, void*);  // nastar_placement_from_levels
using fwd_ex_t = int (*)(const float*, const float*, const float*, const float*, int, int, int, double, int, float*, int64_t*, int32_t*, int32_t*,
                         int32_t*, uint8_t*, void*, size_t, int, const int32_t*, int32_t*, int32_t*, int32_t*, void*);

inline bool plain_f32(const at::Tensor& t) { return t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous(); }

py::tuple search(uintptr_t fn, const at::Tensor& cost, const at::Tensor& start, const at::Tensor& goal, const c10::optional<at::Tensor>& passable,
                 double g_ratio, int64_t max_iters, bool want_log, int64_t flags, const c10::optional<at::Tensor>& order,
                 const c10::optional<at::Tensor>& order_out, int64_t ws_bytes, uintptr_t summary_host, uintptr_t counter_dev, uintptr_t stream,
                 int64_t spin_us, const c10::optional<at::Tensor>& levels, uintptr_t sort_fn, const c10::optional<at::Tensor>& packed)
{
    const at::Tensor& pas = passable.has_value() ? *passable : cost;
    // anything the fast lane does not take goes back to the Python path, which raises the proper errors: rc = -1
    if (!plain_f32(cost) || !plain_f32(start) || !plain_f32(goal) || !plain_f32(pas) || cost.dim() != 4 || cost.size(1) != 1 ||
        start.sizes() != cost.sizes() || goal.sizes() != cost.sizes() || pas.sizes() != cost.sizes())
        return py::make_tuple(py::none(), py::none(), py::none(), py::none(), py::none(), -1, -1);
    const int64_t B = cost.size(0), H = cost.size(2), W = cost.size(3);
    const auto fopt = cost.options();
    at::Tensor hist = at::empty({B, 1, H, W}, fopt);
    at::Tensor paths = at::empty({B, 1, H, W}, fopt.dtype(at::kLong));
    at::Tensor meta = at::empty({2, B}, fopt.dtype(at::kInt));
    at::Tensor log, ws;
    if (want_log) log = at::empty({B, max_iters}, fopt.dtype(at::kInt));
    if (ws_bytes > 0) ws = at::empty({ws_bytes}, fopt.dtype(at::kByte));
    const int32_t* op = nullptr;
    int32_t* oo = nullptr;
    if (order.has_value()) {
        if (order->scalar_type() != at::kInt || order->numel() != B || !order->is_contiguous() || order->device() != cost.device())
            return py::make_tuple(py::none(), py::none(), py::none(), py::none(), py::none(), -1, -1);
        op = order->data_ptr<int32_t>();
    }
    at::Tensor sorted;
    if (!order.has_value() && levels.has_value() && sort_fn != 0) {
        // the batch carries its loader's LEVELS (the start cells' optimal distances), not an order yet: the counting sort that turns them into a
        // placement is launched here, in front of the search, on the same stream (a permutation by construction: nothing to check)
        if (levels->scalar_type() != at::kInt || levels->numel() != B || !levels->is_contiguous() || levels->device() != cost.device())
            return py::make_tuple(py::none(), py::none(), py::none(), py::none(), py::none(), -1, -1);
        sorted = at::empty({B}, fopt.dtype(at::kInt));
        const int src = reinterpret_cast<sort_t>(sort_fn)(levels->data_ptr<int32_t>(), (int)B, sorted.data_ptr<int32_t>(), reinterpret_cast<void*>(stream));
        if (src != 0) return py::make_tuple(py::none(), py::none(), py::none(), py::none(), py::none(), src, -1);
        op = sorted.data_ptr<int32_t>();
    }
    if (order_out.has_value()) {
        if (order_out->scalar_type() != at::kInt || order_out->numel() != B + 1 || !order_out->is_contiguous() || order_out->device() != cost.device())
            return py::make_tuple(py::none(), py::none(), py::none(), py::none(), py::none(), -1, -1);
        oo = order_out->data_ptr<int32_t>();
    }
    uint8_t* pk = nullptr;
    if (packed.has_value()) {
        // the caller's slot for the bit-packed masks (2 bits per cell: a collation bucket of parallel.BucketedCollator): the search launch emits them
        if (packed->scalar_type() != at::kByte || packed->numel() != B * 2 * ((H * W + 7) / 8) || !packed->is_contiguous() || packed->device() != cost.device())
            return py::make_tuple(py::none(), py::none(), py::none(), py::none(), py::none(), -1, -1);
        pk = packed->data_ptr<uint8_t>();
    }
    int32_t* iters = meta.data_ptr<int32_t>();
    int32_t* status = iters + B;
    volatile int32_t* summ = reinterpret_cast<volatile int32_t*>(summary_host);
    const int rc = reinterpret_cast<fwd_ex_t>(fn)(cost.data_ptr<float>(), start.data_ptr<float>(), goal.data_ptr<float>(), pas.data_ptr<float>(), (int)B, (int)H,
                                                  (int)W, g_ratio, (int)max_iters, hist.data_ptr<float>(), paths.data_ptr<int64_t>(),
                                                  want_log ? log.data_ptr<int32_t>() : nullptr, iters, status, pk,
                                                  ws_bytes > 0 ? ws.data_ptr() : nullptr, (size_t)ws_bytes, (int)flags, op, oo,
                                                  reinterpret_cast<int32_t*>(summary_host), summary_host ? reinterpret_cast<int32_t*>(counter_dev) : nullptr,
                                                  reinterpret_cast<void*>(stream));
    int verdict = -1;
    if (rc == 0 && summ != nullptr && counter_dev != 0 && spin_us > 0) {
        py::gil_scoped_release nogil;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0;; ++spin) {
            if (summ[0] != 0) {
                std::atomic_thread_fence(std::memory_order_acquire);
                int v = 0;
                for (int c = 1; c < 16; ++c)
                    if (summ[c] != 0) v |= 1 << c;
                for (int c = 0; c < 16; ++c) summ[c] = 0;  // the row goes back clean
                verdict = v;
                break;
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#elif defined(__aarch64__)
            asm volatile("yield");
#endif
            if ((spin & 255u) == 255u &&
                std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= spin_us)
                break;
        }
    }
    py::object logo = want_log ? py::cast(log) : py::none();
    return py::make_tuple(hist, paths, meta.select(0, 0), meta.select(0, 1), logo, rc, verdict);
}

}  // namespace

PYBIND11_MODULE(_nastar_fastlane, m)
{
    m.doc() = "host side of one checked nastar_forward_ex call in native code (csrc/nastar_fastlane.cpp)";
    m.def("search", &search, py::arg("fn"), py::arg("cost"), py::arg("start"), py::arg("goal"), py::arg("passable"), py::arg("g_ratio"),
          py::arg("max_iters"), py::arg("want_log"), py::arg("flags"), py::arg("order"), py::arg("order_out"), py::arg("ws_bytes"),
          py::arg("summary_host"), py::arg("counter_dev"), py::arg("stream"), py::arg("spin_us"), py::arg("levels") = py::none(), py::arg("sort_fn") = 0,
          py::arg("packed") = py::none());
}
