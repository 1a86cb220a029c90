// polympc_amd — the context object behind the opaque pmpc_context handle and the host-side staging helpers shared by the
// translation units of libpolympc_amd.so (pmpc_api.hip and one pmpc_model_*.hip per built-in OCP).
#pragma once
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>
#include "../../include/polympc_amd.h"
#include "pmpc_cheb.hpp"

using pmpc::ChebData;
using pmpc::make_cheb_data;

// The host thread that drives one context's shard in pmpc_sqp_solve_batch_multi (one host thread + stream per device). It is started
// on the first sharded call and lives as long as the context, so a receding-horizon loop does not create and join a thread per device and step.
struct ShardWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = false, stop = false;
    bool start() {
        if (th.joinable()) return true;
        try { th = std::thread([this] { loop(); }); } catch (...) { return false; }
        return true;
    }
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [this] { return has_job || stop; });
            if (stop) return;
            std::function<void()> j; j.swap(job); has_job = false;
            lk.unlock();
            j();
            lk.lock();
            done = true;
            cv.notify_all();
        }
    }
    void post(std::function<void()> j) { { std::lock_guard<std::mutex> lk(mu); job = std::move(j); has_job = true; done = false; } cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return done; }); }
    ~ShardWorker() {
        if (!th.joinable()) return;
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        th.join();
    }
};

// =====================================================================================================================
// context
// =====================================================================================================================
struct pmpc_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    size_t lds_limit = 64 * 1024;
    unsigned long long* phase_cycles = nullptr;   // PMPC_PHASE_PROFILE=1: per-phase shader-clock totals of the SQP kernels
    int simd_count = 1024;         // compute units x 4
    int sqp_slice = 0;             // PMPC_SQP_SLICE=k: run k SQP iterations per kernel launch with per-instance state in HBM (finished
                                   // instances free their slots); 0 (default) = whole solve in one launch — measured faster on config A
    int sqp_rr = 0;                // PMPC_SQP_RR=1: batches beyond the resident wavefronts run one SQP iteration per work item from a ready queue (sqp_kernel_rr,
                                   // pmpc_launch.hpp); 0 (default) = one workgroup per instance — measured equal or faster on configs A and D
    int last_route = 0;            // pmpc_route of the last fused SQP launch (pmpc_sqp_last_route)
    int poison = 0;                // PMPC_POISON=1 / pmpc_debug_set_poison: fill the HBM workspace, the staging buffers, every CU's LDS and every SIMD's register
                                   // file with signalling NaNs before each launch (pmpc_poison.hip) — an uninitialised read returns NaN, not a plausible stale value
    size_t lds_limit_device = 64 * 1024;   // the device's opt-in maximum of dynamic LDS per workgroup (lds_limit may be lowered by PMPC_LDS_LIMIT)
    bool force_lds_path = false;   // PMPC_FORCE_LDS_PATH=1: disable the register-resident specialisations (A/B testing)
    unsigned dev_switches = 0;     // the launcher's developer switches (PMPC_NO_REDO_LAUNCH, PMPC_NO_CONDREG, PMPC_NO_SCHUR, PMPC_SCHUR_SMALL, PMPC_BIG_WG4), read ONCE at pmpc_create:
                                   // no getenv on the launch path (ShardWorker threads would race a host program's setenv), see pmpc_internal_switch
    std::map<std::tuple<int, int, double, double>, ChebData*> cheb_cache;
    double* ws = nullptr; size_t ws_bytes = 0;       // SQP HBM workspace (H, J)
    void* scratch[24] = {nullptr}; size_t scratch_bytes[24] = {0};  // host-buffer API staging
    ShardWorker* shard_worker = nullptr;             // pmpc_sqp_solve_batch_multi: this context's persistent host thread (created on first use)
};

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "polympc_amd: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); return PMPC_ERR_HIP; } } while (0)

constexpr unsigned PMPC_POISON_DWORD = 0x7FF47FF4u;   // any pair of these dwords is an fp64 signalling NaN (pmpc_poison.hip)
inline pmpc_status ensure_ws(pmpc_context* ctx, size_t bytes) {
    if (ctx->ws_bytes < bytes) {
        if (ctx->ws) HIPCHK(hipFree(ctx->ws));
        ctx->ws = nullptr; ctx->ws_bytes = 0;
        HIPCHK(hipMalloc((void**)&ctx->ws, bytes));
        ctx->ws_bytes = bytes;
    }
    if (ctx->poison && ctx->ws_bytes >= 4) HIPCHK(hipMemsetD32Async((hipDeviceptr_t)ctx->ws, (int)PMPC_POISON_DWORD, ctx->ws_bytes / 4, ctx->stream));   // the workspace is scratch between calls: every kernel must write what it reads
    return PMPC_OK;
}
inline pmpc_status ensure_scratch(pmpc_context* ctx, int slot, size_t bytes, void** out) {
    if (bytes == 0) bytes = 8;
    if (ctx->scratch_bytes[slot] < bytes) {
        if (ctx->scratch[slot]) HIPCHK(hipFree(ctx->scratch[slot]));
        ctx->scratch[slot] = nullptr; ctx->scratch_bytes[slot] = 0;
        HIPCHK(hipMalloc(&ctx->scratch[slot], bytes));
        ctx->scratch_bytes[slot] = bytes;
    }
    // (poison mode: the whole slot — inputs are copied over it on the same stream, outputs must be written by the kernels in full)
    if (ctx->poison && ctx->scratch_bytes[slot] >= 4) HIPCHK(hipMemsetD32Async((hipDeviceptr_t)ctx->scratch[slot], (int)PMPC_POISON_DWORD, ctx->scratch_bytes[slot] / 4, ctx->stream));
    *out = ctx->scratch[slot];
    return PMPC_OK;
}
extern "C" pmpc_status pmpc_internal_poison_device(pmpc_context* ctx);   // pmpc_poison.hip: LDS, register files and low scratch of every CU (no-op unless ctx->poison)
#define PMPC_POISON_DEVICE(ctx) do { if ((ctx)->poison) { const pmpc_status ps_ = pmpc_internal_poison_device(ctx); if (ps_ != PMPC_OK) return ps_; } } while (0)
inline pmpc_status get_cheb(pmpc_context* ctx, int P, int S, double t0, double tf, const ChebData** out) {
    auto key = std::make_tuple(P, S, t0, tf);
    auto it = ctx->cheb_cache.find(key);
    if (it != ctx->cheb_cache.end()) { *out = it->second; return PMPC_OK; }
    ChebData cd;
    if (!make_cheb_data(P, S, t0, tf, cd)) return PMPC_ERR_UNSUPPORTED_SIZE;
    if (ctx->cheb_cache.size() >= 64) {   // a receding-horizon caller that shifts (t0, tf) every step must not grow the cache without bound
        HIPCHK(hipStreamSynchronize(ctx->stream));   // (kernels in flight may still read the entries)
        for (auto& kv : ctx->cheb_cache) (void)hipFree(kv.second);
        ctx->cheb_cache.clear();
    }
    ChebData* dptr = nullptr;
    HIPCHK(hipMalloc((void**)&dptr, sizeof(ChebData)));
    HIPCHK(hipMemcpyAsync(dptr, &cd, sizeof(ChebData), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // cd is a stack object
    ctx->cheb_cache[key] = dptr;
    *out = dptr;
    return PMPC_OK;
}


#define H2D(slot, host, count, devptr)                                                                        \
    do {                                                                                                      \
        void* p_ = nullptr;                                                                                   \
        if (host) {                                                                                           \
            pmpc_status st_ = ensure_scratch(ctx, slot, (size_t)(count) * sizeof(double), &p_);               \
            if (st_ != PMPC_OK) return st_;                                                                   \
            HIPCHK(hipMemcpyAsync(p_, host, (size_t)(count) * sizeof(double), hipMemcpyHostToDevice, ctx->stream)); \
        }                                                                                                     \
        devptr = (double*)p_;                                                                                 \
    } while (0)
#define DEVOUT(slot, bytes, devptr)                                                \
    do {                                                                           \
        void* p_ = nullptr;                                                        \
        pmpc_status st_ = ensure_scratch(ctx, slot, (size_t)(bytes), &p_);         \
        if (st_ != PMPC_OK) return st_;                                            \
        devptr = (decltype(devptr))p_;                                             \
    } while (0)
