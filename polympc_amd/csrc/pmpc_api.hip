// polympc_amd — kernels + the C ABI declared in include/polympc_amd.h (gfx950 only, no CPU fallback).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/polympc_amd.h"
#include "pmpc_context.hpp"
#include "pmpc_builtin.hpp"
#include "pmpc_ocp.hpp"
#include "pmpc_qp.hpp"
#include "pmpc_qp_reg.hpp"
#include "pmpc_sqp.hpp"
#include "pmpc_launch.hpp"
#include "pmpc_ruiz.hpp"
#include "pmpc_admm.hpp"


using namespace pmpc;

extern "C" int pmpc_internal_qp_reg2_launch(void* stream, int B, int n, int m, const double* H, const double* h, const double* A, const double* Alb,
                                            const double* Aub, const double* xlb, const double* xub, const double* x0, const double* y0,
                                            const pmpc_qp_settings* s, double* x, double* y, pmpc_qp_info* info);

// =====================================================================================================================
// kernels: one 64-lane workgroup (= one wavefront) per instance; grid = batch
// =====================================================================================================================
__device__ __forceinline__ void qp_boxadmm_one(int b, int n, int m, const double* __restrict__ H, const double* __restrict__ h, const double* __restrict__ A,
                                               const double* __restrict__ Alb, const double* __restrict__ Aub, const double* __restrict__ xlb,
                                               const double* __restrict__ xub, const double* __restrict__ x0, const double* __restrict__ y0,
                                               const pmpc_qp_settings& s, double* __restrict__ x, double* __restrict__ y, pmpc_qp_info* __restrict__ info,
                                               double* smem, int extra_flags) {
    QpLds w;
    double* p = w.carve(smem, n, m);
    // stage the vectors the ADMM loop touches every iteration: h, Alb, Aub, xlb, xub
    double* hL = p; p += n; double* albL = p; p += m; double* aubL = p; p += m; double* xlbL = p; p += n; double* xubL = p; p += n;
    const int ln = lane_id();
    for (int i = ln; i < n; i += WAVE) { hL[i] = h[(size_t)b * n + i]; xlbL[i] = xlb[(size_t)b * n + i]; xubL[i] = xub[(size_t)b * n + i]; }
    for (int i = ln; i < m; i += WAVE) { albL[i] = Alb[(size_t)b * m + i]; aubL[i] = Aub[(size_t)b * m + i]; }
    wsync();
    pmpc_qp_info qi;
    boxadmm_solve(w, n, m, H + (size_t)b * n * n, n, hL, A + (size_t)b * m * n, m, albL, aubL, xlbL, xubL,
                  x0 ? x0 + (size_t)b * n : nullptr, y0 ? y0 + (size_t)b * (n + m) : nullptr, s, qi);
    for (int i = ln; i < n; i += WAVE) x[(size_t)b * n + i] = w.x[i];
    for (int i = ln; i < n + m; i += WAVE) y[(size_t)b * (n + m) + i] = w.y[i];
    qi.flags |= extra_flags;
    if (ln == 0) info[b] = qi;
    wsync();
}
__global__ __launch_bounds__(64) void qp_boxadmm_kernel(int B, int n, int m, const double* __restrict__ H,
                                                        const double* __restrict__ h, const double* __restrict__ A,
                                                        const double* __restrict__ Alb, const double* __restrict__ Aub,
                                                        const double* __restrict__ xlb, const double* __restrict__ xub,
                                                        const double* __restrict__ x0, const double* __restrict__ y0,
                                                        pmpc_qp_settings s, double* __restrict__ x, double* __restrict__ y,
                                                        pmpc_qp_info* __restrict__ info, int redo) {
    extern __shared__ double smem[];
    if (redo) {
        // redo launch behind a one-row-per-lane register kernel (grid = ceil(B / 64)): this workgroup looks at 64 QPs at once — one word each — and solves,
        // one after the other, those that gave up at their conditioning gate (PMPC_FLAG_ILLCOND; normally none: 256 workgroups that read a word and exit
        // behind 16 384 QPs instead of 16 384 of them — the QP entry point's batches are flat and large, and a workgroup launch is not free)
        const int base = (int)blockIdx.x * WAVE, ln = lane_id();
        const int fl = (base + ln < B) ? info[base + ln].flags : 0;
        unsigned long long todo = __builtin_amdgcn_ballot_w64((fl & PMPC_FLAG_ILLCOND) != 0);
        while (todo) {
            const int i = __builtin_ctzll(todo);
            todo &= todo - 1;
            qp_boxadmm_one(base + i, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0, s, x, y, info, smem, PMPC_FLAG_ILLCOND);   // (the flag stays: this QP took the full KKT form)
        }
        return;
    }
    const int b = blockIdx.x;
    if (b >= B) return;
    qp_boxadmm_one(b, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0, s, x, y, info, smem, 0);
}
// register-resident specialisation for compile-time (NN, MM), NN+MM <= 64
template <int NN, int MM>
__global__ __launch_bounds__(64, 2) void qp_boxadmm_reg_kernel(int B, const double* __restrict__ H, const double* __restrict__ h,
                                                            const double* __restrict__ A, const double* __restrict__ Alb,
                                                            const double* __restrict__ Aub, const double* __restrict__ xlb,
                                                            const double* __restrict__ xub, const double* __restrict__ x0,
                                                            const double* __restrict__ y0, pmpc_qp_settings s,
                                                            double* __restrict__ x, double* __restrict__ y, pmpc_qp_info* __restrict__ info) {
    __shared__ double tr[RegKkt<NN + MM>::TRI];
    const int b = blockIdx.x;
    if (b >= B) return;
    pmpc_qp_info qi;
    boxadmm_solve_reg<NN, MM, false, true, true>(H + (size_t)b * NN * NN, h + (size_t)b * NN, A + (size_t)b * MM * NN, Alb + (size_t)b * MM, Aub + (size_t)b * MM,
                              xlb + (size_t)b * NN, xub + (size_t)b * NN, x0 ? x0 + (size_t)b * NN : nullptr,
                              y0 ? y0 + (size_t)b * (NN + MM) : nullptr, s, qi, x + (size_t)b * NN, y + (size_t)b * (NN + MM), tr);
    if (lane_id() == 0) info[b] = qi;
}
static size_t qp_kernel_lds_bytes(int n, int m) { return (QpLds::doubles(n, m) + 3 * (size_t)n + 2 * (size_t)m) * sizeof(double); }

extern "C" size_t pmpc_internal_qp_big_ws_doubles(int n, int m);
extern "C" size_t pmpc_internal_qp_big_lds_bytes(int n, int m);
extern "C" int pmpc_internal_qp_big_launch(void* stream, double* Kws, int B, int n, int m, const double* H, const double* h, const double* A,
                                           const double* Alb, const double* Aub, const double* xlb, const double* xub, const double* x0,
                                           const double* y0, const pmpc_qp_settings* s, double* x, double* y, pmpc_qp_info* info);
constexpr int PMPC_QP_BIG_MIN_ROWS = 112;   // measured on 4096 random QPs, 51 iterations (HBM factor vs LDS triangle): 96 rows 4.8 vs 3.8 ms, 128 rows 6.8 vs 7.4, 168 rows 12.5 vs 73.1
                                             // (the fused SQP kernel switches at 96: its LDS-resident variant carries the SQP vectors too, pmpc_launch.hpp)

extern "C" pmpc_status pmpc_internal_services(pmpc_context* ctx, int P, int S, double t0, double tf, size_t ws_bytes, const void** cheb,
                                               double** ws, void** stream, size_t* lds_limit, unsigned long long** phase_cycles, int* force_lds) {
    if (!ctx) return PMPC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(ctx->device));
    const ChebData* cd = nullptr;
    pmpc_status st = get_cheb(ctx, P, S, t0, tf, &cd);
    if (st != PMPC_OK) return st;
    st = ensure_ws(ctx, ws_bytes);
    if (st != PMPC_OK) return st;
    PMPC_POISON_DEVICE(ctx);   // (developer harness: every fused SQP launch asks for its services first)
    *cheb = cd; *ws = ctx->ws; *stream = (void*)ctx->stream; *lds_limit = ctx->lds_limit; *phase_cycles = ctx->phase_cycles;
    *force_lds = ctx->force_lds_path ? 1 : 0;
    return PMPC_OK;
}

extern "C" int pmpc_internal_sqp_slice(pmpc_context* ctx) { return ctx ? ctx->sqp_slice : 0; }
extern "C" int pmpc_internal_sqp_rr(pmpc_context* ctx) { return ctx ? ctx->sqp_rr : 0; }
extern "C" int pmpc_internal_simd_count(pmpc_context* ctx) { return ctx ? ctx->simd_count : 1024; }
extern "C" int pmpc_internal_switch(pmpc_context* ctx, int which) { return ctx ? (int)((ctx->dev_switches >> which) & 1u) : 0; }
extern "C" void pmpc_internal_set_route(pmpc_context* ctx, int route) { if (ctx) ctx->last_route = route; }
extern "C" int pmpc_internal_last_route(pmpc_context* ctx) { return ctx ? ctx->last_route : 0; }


// =====================================================================================================================
// C ABI
// =====================================================================================================================
extern "C" {

#define PMPC_STR2(x) #x
#define PMPC_STR(x) PMPC_STR2(x)
const char* pmpc_version(void) { return "polympc_amd 0.3 (gfx950, abi " PMPC_STR(PMPC_ABI_VERSION) ")"; }
int pmpc_abi_version(void) { return PMPC_ABI_VERSION; }
unsigned long pmpc_struct_size(int which) {
    switch (which) {
        case 0: return (unsigned long)sizeof(pmpc_qp_settings);
        case 1: return (unsigned long)sizeof(pmpc_qp_info);
        case 2: return (unsigned long)sizeof(pmpc_sqp_settings);
        case 3: return (unsigned long)sizeof(pmpc_sqp_info);
    }
    return 0;
}
int pmpc_sqp_last_route(pmpc_context* ctx) { return ctx ? ctx->last_route : PMPC_ROUTE_NONE; }
const char* pmpc_status_string(pmpc_status s) {
    switch (s) {
        case PMPC_OK: return "ok";
        case PMPC_ERR_INVALID_ARGUMENT: return "invalid argument";
        case PMPC_ERR_NO_DEVICE: return "no HIP device (there is no CPU fallback)";
        case PMPC_ERR_HIP: return "HIP runtime error";
        case PMPC_ERR_UNSUPPORTED_SIZE: return "problem size not supported by the LDS-resident kernels";
        case PMPC_ERR_UNKNOWN_MODEL: return "unknown model id";
        case PMPC_ERR_ABI_MISMATCH: return "library and header / binding come from different ABI versions";
    }
    return "?";
}

static pmpc_status create_impl(int device, void* stream, pmpc_context* ctx);
pmpc_status pmpc_create(int device, void* stream, pmpc_context** out) {
    if (!out) return PMPC_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return PMPC_ERR_NO_DEVICE;
    HIPCHK(hipSetDevice(device));
    pmpc_context* ctx = new pmpc_context();
    const pmpc_status st = create_impl(device, stream, ctx);
    if (st != PMPC_OK) {   // nothing of a half-built context outlives the failed call
        if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
        if (ctx->phase_cycles) (void)hipFree(ctx->phase_cycles);
        delete ctx;
        return st;
    }
    *out = ctx;
    return PMPC_OK;
}
static pmpc_status create_impl(int device, void* stream, pmpc_context* ctx) {
    ctx->device = device;
    if (stream) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
    else { HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    ctx->simd_count = 4 * (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
    ctx->lds_limit = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 64 * 1024;
    if (prop.sharedMemPerBlockOptin && (size_t)prop.sharedMemPerBlockOptin < ctx->lds_limit) ctx->lds_limit = prop.sharedMemPerBlockOptin;
    ctx->lds_limit_device = ctx->lds_limit;
    { const char* e = getenv("PMPC_POISON"); ctx->poison = (e && e[0] && e[0] != '0') ? 1 : 0; }   // developer harness (pmpc_poison.hip)
    { const char* e = getenv("PMPC_LDS_LIMIT"); if (e && e[0] && atol(e) > 0 && (size_t)atol(e) < ctx->lds_limit) ctx->lds_limit = (size_t)atol(e); }   // developer switch: a smaller LDS budget (moves mid-size instances to the HBM-factor kernel)
    { const char* e = getenv("PMPC_FORCE_LDS_PATH"); ctx->force_lds_path = (e && e[0] == '1'); }
    { const char* e = getenv("PMPC_SQP_SLICE"); if (e && e[0]) ctx->sqp_slice = atoi(e) < 0 ? 0 : atoi(e); }
    { const char* e = getenv("PMPC_SQP_RR"); if (e && e[0]) ctx->sqp_rr = atoi(e) != 0 ? 1 : 0; }
    {   // the launcher's developer switches (pmpc_launch.hpp: pmpc_dev_switch), read once per context
        ctx->dev_switches = 0;
        if (getenv("PMPC_NO_REDO_LAUNCH")) ctx->dev_switches |= 1u << PMPC_SW_NO_REDO_LAUNCH;
        if (getenv("PMPC_NO_CONDREG")) ctx->dev_switches |= 1u << PMPC_SW_NO_CONDREG;
        if (getenv("PMPC_NO_SCHUR")) ctx->dev_switches |= 1u << PMPC_SW_NO_SCHUR;
        if (getenv("PMPC_NO_CONDREG_RUIZ")) ctx->dev_switches |= 1u << PMPC_SW_NO_CONDREG_RUIZ;   // (A/B timing: preconditioner = 1 on the full two-rows-per-lane inverse as before round 6)
        if (getenv("PMPC_SCHUR_SMALL")) ctx->dev_switches |= 1u << PMPC_SW_SCHUR_SMALL;
        const char* e = getenv("PMPC_BIG_WG4");
        if (e && e[0]) ctx->dev_switches |= (e[0] != '0') ? (1u << PMPC_SW_BIG_WG4_ON) : (1u << PMPC_SW_BIG_WG4_OFF);
    }
    { const char* e = getenv("PMPC_PHASE_PROFILE");
      if (e && e[0] == '1') { HIPCHK(hipMalloc((void**)&ctx->phase_cycles, 24 * sizeof(unsigned long long))); HIPCHK(hipMemset(ctx->phase_cycles, 0, 24 * sizeof(unsigned long long))); } }
    return PMPC_OK;
}
pmpc_status pmpc_destroy(pmpc_context* ctx) {
    if (!ctx) return PMPC_ERR_INVALID_ARGUMENT;
    delete ctx->shard_worker;   // joins the shard thread (idle unless a sharded call is in flight, which the caller must not destroy under)
    ctx->shard_worker = nullptr;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->cheb_cache) (void)hipFree(kv.second);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->phase_cycles) (void)hipFree(ctx->phase_cycles);
    for (int i = 0; i < 24; ++i) if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PMPC_OK;
}
pmpc_status pmpc_debug_phase_cycles(pmpc_context* ctx, unsigned long long* out24, int reset) {
    if (!ctx || !out24) return PMPC_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < 24; ++i) out24[i] = 0;
    if (!ctx->phase_cycles) return PMPC_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(out24, ctx->phase_cycles, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(hipMemset(ctx->phase_cycles, 0, 24 * sizeof(unsigned long long)));
    return PMPC_OK;
}
pmpc_status pmpc_debug_set_poison(pmpc_context* ctx, int on) {
    if (!ctx) return PMPC_ERR_INVALID_ARGUMENT;
    ctx->poison = on ? 1 : 0;
    return PMPC_OK;
}
pmpc_status pmpc_synchronize(pmpc_context* ctx) {
    if (!ctx) return PMPC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PMPC_OK;
}

pmpc_status pmpc_filter_state_create(pmpc_context* ctx, int B, double** filter_state) {
    if (!ctx || B < 1 || !filter_state) return PMPC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(ctx->device));
    double* p = nullptr;
    const size_t bytes = (size_t)B * PMPC_FILTER_STATE_DOUBLES * sizeof(double);
    HIPCHK(hipMalloc((void**)&p, bytes));
    HIPCHK(hipMemsetAsync(p, 0, bytes, ctx->stream));
    *filter_state = p;
    return PMPC_OK;
}
pmpc_status pmpc_filter_state_clear(pmpc_context* ctx, int B, double* filter_state) {
    if (!ctx || B < 1 || !filter_state) return PMPC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipMemsetAsync(filter_state, 0, (size_t)B * PMPC_FILTER_STATE_DOUBLES * sizeof(double), ctx->stream));
    return PMPC_OK;
}
pmpc_status pmpc_filter_state_download(pmpc_context* ctx, int B, const double* filter_state, double* host_out) {
    if (!ctx || B < 1 || !filter_state || !host_out) return PMPC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipMemcpyAsync(host_out, filter_state, (size_t)B * PMPC_FILTER_STATE_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PMPC_OK;
}
pmpc_status pmpc_filter_state_destroy(pmpc_context* ctx, double* filter_state) {
    if (!ctx) return PMPC_ERR_INVALID_ARGUMENT;
    if (filter_state) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(filter_state)); }
    return PMPC_OK;
}

pmpc_status pmpc_iteration_trace_create(pmpc_context* ctx, int B, int capacity, double** trace) {
    if (!ctx || B < 1 || capacity < 1 || !trace) return PMPC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(ctx->device));
    double* p = nullptr;
    const size_t bytes = (size_t)B * capacity * PMPC_TRACE_DOUBLES * sizeof(double);
    HIPCHK(hipMalloc(&p, bytes));
    if (hipMemsetAsync(p, 0, bytes, ctx->stream) != hipSuccess) { (void)hipFree(p); return PMPC_ERR_HIP; }
    *trace = p;
    return PMPC_OK;
}
pmpc_status pmpc_iteration_trace_clear(pmpc_context* ctx, int B, int capacity, double* trace) {
    if (!ctx || B < 1 || capacity < 1 || !trace) return PMPC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipMemsetAsync(trace, 0, (size_t)B * capacity * PMPC_TRACE_DOUBLES * sizeof(double), ctx->stream));
    return PMPC_OK;
}
pmpc_status pmpc_iteration_trace_download(pmpc_context* ctx, int B, int capacity, const double* trace, double* host_out) {
    if (!ctx || B < 1 || capacity < 1 || !trace || !host_out) return PMPC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipMemcpyAsync(host_out, trace, (size_t)B * capacity * PMPC_TRACE_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PMPC_OK;
}
pmpc_status pmpc_iteration_trace_destroy(pmpc_context* ctx, double* trace) {
    if (!ctx) return PMPC_ERR_INVALID_ARGUMENT;
    if (trace) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(trace)); }
    return PMPC_OK;
}

void pmpc_qp_settings_default(pmpc_qp_settings* s) {
    s->eps_rel = 1e-3; s->eps_abs = 1e-3; s->max_iter = 1000; s->rho = 1e-1; s->sigma = 1e-6; s->alpha = 1.0;
    s->check_termination = 25; s->adaptive_rho = 0; s->adaptive_rho_tolerance = 5; s->adaptive_rho_interval = 25; s->linear_solver = 0;
}
void pmpc_qp_settings_sqp_default(pmpc_qp_settings* s) {
    pmpc_qp_settings_default(s);
    s->check_termination = 10; s->eps_abs = 1e-4; s->eps_rel = 1e-4; s->max_iter = 100; s->adaptive_rho = 1;
    s->adaptive_rho_interval = 50; s->alpha = 1.0;
}
void pmpc_sqp_settings_default(pmpc_sqp_settings* s) {
    s->tau = 0.5; s->eta = 0.25; s->rho = 0.5; s->eps_prim = 1e-3; s->eps_dual = 1e-3; s->max_iter = 100;
    s->line_search_max_iter = 100; s->regularisation = 0; s->exact_hessian_every_iter = 0; s->preconditioner = 0; s->hessian_update = 0; s->qp_solver = 0;
    s->line_search = 0; s->filter_max_depth = PMPC_FILTER_MAX_DEPTH; s->filter_beta = 1e-5; s->filter_state = nullptr;
    s->iteration_trace = nullptr; s->iteration_trace_capacity = 0; s->kkt_form = 0;
}

pmpc_status pmpc_chebyshev(int P, double* nodes, double* weights, double* D) {
    if (P < 1 || !nodes || !weights || !D) return PMPC_ERR_INVALID_ARGUMENT;
    cheb_nodes(P, nodes); cheb_weights(P, weights); cheb_diff_matrix(P, D);
    return PMPC_OK;
}

pmpc_status pmpc_qp_boxadmm_solve_batch_dev(pmpc_context* ctx, int B, int n, int m, const double* H, const double* h,
                                            const double* A, const double* Alb, const double* Aub, const double* xlb,
                                            const double* xub, const double* x0, const double* y0,
                                            const pmpc_qp_settings* settings, double* x, double* y, pmpc_qp_info* info) {
    if (!ctx || B < 0 || n < 1 || m < 0 || !H || !h || !xlb || !xub || !settings || !x || !y || !info) return PMPC_ERR_INVALID_ARGUMENT;
    if (m > 0 && (!A || !Alb || !Aub)) return PMPC_ERR_INVALID_ARGUMENT;
    if ((x0 == nullptr) != (y0 == nullptr)) return PMPC_ERR_INVALID_ARGUMENT;
    if (B == 0) return PMPC_OK;
    HIPCHK(hipSetDevice(ctx->device));
    if (settings->linear_solver != 0 && settings->linear_solver != 1) return PMPC_ERR_INVALID_ARGUMENT;
    PMPC_POISON_DEVICE(ctx);
    const bool static_order = settings->linear_solver == 0 && !ctx->force_lds_path;   // the register-resident specialisations factorise in a static order
    // register-resident specialisations (one KKT row per lane): config A's QP and the QPs of the robot / CSTR grids of 4 to 8 nodes
#define PMPC_REG1_CASE(NN_, MM_)                                                                                                             \
    if (n == NN_ && m == MM_ && static_order) {                                                                                              \
        hipLaunchKernelGGL((qp_boxadmm_reg_kernel<NN_, MM_>), dim3(B), dim3(WAVE), 0, ctx->stream, B, H, h, A, Alb, Aub, xlb, xub, x0, y0,   \
                           *settings, x, y, info);                                                                                           \
        /* redo launch: the QPs that gave up at the conditioning gate of the constraint-first sweep, on the LDS-resident static LDL^T */          \
        const size_t ldsg_ = qp_kernel_lds_bytes(n, m);                                                                                      \
        if (ldsg_ <= ctx->lds_limit && !pmpc_internal_switch(ctx, PMPC_SW_NO_REDO_LAUNCH)) {                                                                 \
            HIPCHK(hipFuncSetAttribute((const void*)qp_boxadmm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsg_));             \
            hipLaunchKernelGGL(qp_boxadmm_kernel, dim3((B + WAVE - 1) / WAVE), dim3(WAVE), ldsg_, ctx->stream, B, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0, \
                               *settings, x, y, info, 1);                                                                                    \
        }                                                                                                                                    \
        HIPCHK(hipGetLastError());                                                                                                           \
        return PMPC_OK;                                                                                                                      \
    }
    PMPC_REG1_CASE(35, 21)
    PMPC_REG1_CASE(20, 12)
    PMPC_REG1_CASE(25, 15)
    PMPC_REG1_CASE(30, 18)
    PMPC_REG1_CASE(40, 24)
    PMPC_REG1_CASE(24, 16)
    PMPC_REG1_CASE(30, 20)
    PMPC_REG1_CASE(36, 24)
#undef PMPC_REG1_CASE
    if (static_order) {   // 65..112 KKT rows with a two-rows-per-lane register specialisation (pmpc_qp_reg2.hip)
        const int r2 = pmpc_internal_qp_reg2_launch((void*)ctx->stream, B, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0, settings, x, y, info);
        if (r2 < 0) return PMPC_ERR_HIP;
        if (r2 > 0) return PMPC_OK;
    }
    const size_t lds = qp_kernel_lds_bytes(n, m);
    // From BIG_KKT_MIN_ROWS rows on (and whenever the packed triangle does not fit LDS: the reference's kite size, 464 rows) the factor lives in HBM as
    // tiles (pmpc_qp_big.hip: blocked LDL^T with MFMA updates, one QP per SIMD instead of one or two per CU). The pivoted factorisation exists in LDS only.
    if (static_order && n + m >= 16 && (lds > ctx->lds_limit || n + m >= PMPC_QP_BIG_MIN_ROWS)) {
        if (pmpc_internal_qp_big_lds_bytes(n, m) > ctx->lds_limit) return PMPC_ERR_UNSUPPORTED_SIZE;
        const pmpc_status ws = ensure_ws(ctx, (size_t)B * pmpc_internal_qp_big_ws_doubles(n, m) * sizeof(double));
        if (ws != PMPC_OK) return ws;
        if (pmpc_internal_qp_big_launch((void*)ctx->stream, ctx->ws, B, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0, settings, x, y, info) != 0) return PMPC_ERR_HIP;
        return PMPC_OK;
    }
    if (lds > ctx->lds_limit) return PMPC_ERR_UNSUPPORTED_SIZE;
    HIPCHK(hipFuncSetAttribute((const void*)qp_boxadmm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(qp_boxadmm_kernel, dim3(B), dim3(WAVE), lds, ctx->stream, B, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0,
                       *settings, x, y, info, 0);
    HIPCHK(hipGetLastError());
    return PMPC_OK;
}


pmpc_status pmpc_qp_boxadmm_solve_batch(pmpc_context* ctx, int B, int n, int m, const double* H, const double* h,
                                        const double* A, const double* Alb, const double* Aub, const double* xlb,
                                        const double* xub, const double* x0, const double* y0,
                                        const pmpc_qp_settings* settings, double* x, double* y, pmpc_qp_info* info) {
    if (!ctx || B < 0 || n < 1 || m < 0 || !H || !h || !xlb || !xub || !settings || !x || !y || !info) return PMPC_ERR_INVALID_ARGUMENT;
    if (B == 0) return PMPC_OK;
    HIPCHK(hipSetDevice(ctx->device));
    double *dH, *dh, *dA, *dAlb, *dAub, *dxlb, *dxub, *dx0, *dy0, *dx, *dy; pmpc_qp_info* dinfo;
    const size_t Bn = (size_t)B * n, Bm = (size_t)B * m;
    H2D(0, H, Bn * n, dH); H2D(1, h, Bn, dh); H2D(2, (m ? A : nullptr), Bm * n, dA); H2D(3, (m ? Alb : nullptr), Bm, dAlb);
    H2D(4, (m ? Aub : nullptr), Bm, dAub); H2D(5, xlb, Bn, dxlb); H2D(6, xub, Bn, dxub); H2D(7, x0, Bn, dx0); H2D(8, y0, Bn + Bm, dy0);
    DEVOUT(9, Bn * sizeof(double), dx); DEVOUT(10, (Bn + Bm) * sizeof(double), dy); DEVOUT(11, (size_t)B * sizeof(pmpc_qp_info), dinfo);
    if (m == 0) { DEVOUT(2, 8, dA); DEVOUT(3, 8, dAlb); DEVOUT(4, 8, dAub); }
    pmpc_status st = pmpc_qp_boxadmm_solve_batch_dev(ctx, B, n, m, dH, dh, dA, dAlb, dAub, dxlb, dxub, dx0, dy0, settings, dx, dy, dinfo);
    if (st != PMPC_OK) return st;
    HIPCHK(hipMemcpyAsync(x, dx, Bn * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(y, dy, (Bn + Bm) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(info, dinfo, (size_t)B * sizeof(pmpc_qp_info), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PMPC_OK;
}

pmpc_status pmpc_qp_admm_solve_batch_dev(pmpc_context* ctx, int B, int n, int m, const double* H, const double* h, const double* A,
                                         const double* Alb, const double* Aub, const double* xlb, const double* xub, const double* x0,
                                         const double* y0, const pmpc_qp_settings* settings, double* x, double* y, pmpc_qp_info* info) {
    if (!ctx || B < 0 || n < 1 || m < 0 || !H || !h || !xlb || !xub || !settings || !x || !y || !info) return PMPC_ERR_INVALID_ARGUMENT;
    if (m > 0 && (!A || !Alb || !Aub)) return PMPC_ERR_INVALID_ARGUMENT;
    if ((x0 == nullptr) != (y0 == nullptr)) return PMPC_ERR_INVALID_ARGUMENT;
    if (B == 0) return PMPC_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t lds = QpLds::doubles(n, m + n) * sizeof(double);   // the (2n+m)-row KKT factor + vectors of the stacked system
    if (lds > ctx->lds_limit) return PMPC_ERR_UNSUPPORTED_SIZE;
    HIPCHK(hipFuncSetAttribute((const void*)qp_admm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PMPC_POISON_DEVICE(ctx);
    hipLaunchKernelGGL(qp_admm_kernel, dim3(B), dim3(WAVE), lds, ctx->stream, B, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0, *settings, x, y, info);
    HIPCHK(hipGetLastError());
    return PMPC_OK;
}
pmpc_status pmpc_qp_admm_solve_batch(pmpc_context* ctx, int B, int n, int m, const double* H, const double* h, const double* A,
                                     const double* Alb, const double* Aub, const double* xlb, const double* xub, const double* x0,
                                     const double* y0, const pmpc_qp_settings* settings, double* x, double* y, pmpc_qp_info* info) {
    if (!ctx || B < 0 || n < 1 || m < 0 || !H || !h || !xlb || !xub || !settings || !x || !y || !info) return PMPC_ERR_INVALID_ARGUMENT;
    if (B == 0) return PMPC_OK;
    HIPCHK(hipSetDevice(ctx->device));
    double *dH, *dh, *dA, *dAlb, *dAub, *dxlb, *dxub, *dx0, *dy0, *dx, *dy; pmpc_qp_info* dinfo;
    const size_t Bn = (size_t)B * n, Bm = (size_t)B * m;
    H2D(0, H, Bn * n, dH); H2D(1, h, Bn, dh); H2D(2, (m ? A : nullptr), Bm * n, dA); H2D(3, (m ? Alb : nullptr), Bm, dAlb);
    H2D(4, (m ? Aub : nullptr), Bm, dAub); H2D(5, xlb, Bn, dxlb); H2D(6, xub, Bn, dxub); H2D(7, x0, Bn, dx0); H2D(8, y0, Bn + Bm, dy0);
    DEVOUT(9, Bn * sizeof(double), dx); DEVOUT(10, (Bn + Bm) * sizeof(double), dy); DEVOUT(11, (size_t)B * sizeof(pmpc_qp_info), dinfo);
    if (m == 0) { DEVOUT(2, 8, dA); DEVOUT(3, 8, dAlb); DEVOUT(4, 8, dAub); }
    pmpc_status st = pmpc_qp_admm_solve_batch_dev(ctx, B, n, m, dH, dh, dA, dAlb, dAub, dxlb, dxub, dx0, dy0, settings, dx, dy, dinfo);
    if (st != PMPC_OK) return st;
    HIPCHK(hipMemcpyAsync(x, dx, Bn * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(y, dy, (Bn + Bm) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(info, dinfo, (size_t)B * sizeof(pmpc_qp_info), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PMPC_OK;
}

pmpc_status pmpc_qp_ruiz_compute_batch_dev(pmpc_context* ctx, int B, int n, int m, double* H, double* h, double* A, double* Alb,
                                           double* Aub, double* xlb, double* xub, double* D, double* E, double* c) {
    if (!ctx || B < 0 || n < 1 || m < 0 || !H || !h || !xlb || !xub || !D || !c) return PMPC_ERR_INVALID_ARGUMENT;
    if (m > 0 && (!A || !Alb || !Aub || !E)) return PMPC_ERR_INVALID_ARGUMENT;
    if (B == 0) return PMPC_OK;
    HIPCHK(hipSetDevice(ctx->device));
    double* scratch = nullptr;
    DEVOUT(23, (size_t)B * (n + m) * sizeof(double), scratch);
    PMPC_POISON_DEVICE(ctx);
    hipLaunchKernelGGL(ruiz_compute_kernel, dim3(B), dim3(WAVE), 0, ctx->stream, B, n, m, H, h, A, Alb, Aub, xlb, xub, D, E, c, scratch);
    HIPCHK(hipGetLastError());
    return PMPC_OK;
}
pmpc_status pmpc_qp_ruiz_unscale_batch_dev(pmpc_context* ctx, int B, int n, int m, const double* D, const double* E, const double* c,
                                           double* x, double* y) {
    if (!ctx || B < 0 || n < 1 || m < 0 || !D || !c || !x || !y || (m > 0 && !E)) return PMPC_ERR_INVALID_ARGUMENT;
    if (B == 0) return PMPC_OK;
    HIPCHK(hipSetDevice(ctx->device));
    PMPC_POISON_DEVICE(ctx);
    hipLaunchKernelGGL(ruiz_unscale_solution_kernel, dim3(B), dim3(WAVE), 0, ctx->stream, B, n, m, D, E, c, x, y);
    HIPCHK(hipGetLastError());
    return PMPC_OK;
}
pmpc_status pmpc_qp_ruiz_compute_batch(pmpc_context* ctx, int B, int n, int m, double* H, double* h, double* A, double* Alb,
                                       double* Aub, double* xlb, double* xub, double* D, double* E, double* c) {
    if (!ctx || B < 0 || n < 1 || m < 0 || !H || !h || !xlb || !xub || !D || !c) return PMPC_ERR_INVALID_ARGUMENT;
    if (m > 0 && (!A || !Alb || !Aub || !E)) return PMPC_ERR_INVALID_ARGUMENT;
    if (B == 0) return PMPC_OK;
    HIPCHK(hipSetDevice(ctx->device));
    double *dH, *dh, *dA, *dAlb, *dAub, *dxlb, *dxub, *dD, *dE, *dc;
    const size_t Bn = (size_t)B * n, Bm = (size_t)B * m;
    H2D(0, H, Bn * n, dH); H2D(1, h, Bn, dh); H2D(2, (m ? A : nullptr), Bm * n, dA); H2D(3, (m ? Alb : nullptr), Bm, dAlb);
    H2D(4, (m ? Aub : nullptr), Bm, dAub); H2D(5, xlb, Bn, dxlb); H2D(6, xub, Bn, dxub);
    DEVOUT(7, Bn * sizeof(double), dD); DEVOUT(8, (Bm + 1) * sizeof(double), dE); DEVOUT(9, (size_t)B * sizeof(double), dc);
    if (m == 0) { DEVOUT(2, 8, dA); DEVOUT(3, 8, dAlb); DEVOUT(4, 8, dAub); }
    pmpc_status st = pmpc_qp_ruiz_compute_batch_dev(ctx, B, n, m, dH, dh, dA, dAlb, dAub, dxlb, dxub, dD, dE, dc);
    if (st != PMPC_OK) return st;
#define D2H_(host, dev, count) HIPCHK(hipMemcpyAsync(host, dev, (size_t)(count) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream))
    D2H_(H, dH, Bn * n); D2H_(h, dh, Bn); D2H_(xlb, dxlb, Bn); D2H_(xub, dxub, Bn); D2H_(D, dD, Bn); D2H_(c, dc, B);
    if (m > 0) { D2H_(A, dA, Bm * n); D2H_(Alb, dAlb, Bm); D2H_(Aub, dAub, Bm); D2H_(E, dE, Bm); }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PMPC_OK;
}
pmpc_status pmpc_qp_ruiz_unscale_batch(pmpc_context* ctx, int B, int n, int m, const double* D, const double* E, const double* c,
                                       double* x, double* y) {
    if (!ctx || B < 0 || n < 1 || m < 0 || !D || !c || !x || !y || (m > 0 && !E)) return PMPC_ERR_INVALID_ARGUMENT;
    if (B == 0) return PMPC_OK;
    HIPCHK(hipSetDevice(ctx->device));
    double *dD, *dE, *dc, *dx, *dy;
    const size_t Bn = (size_t)B * n, Bm = (size_t)B * m;
    H2D(0, D, Bn, dD); H2D(1, (m ? E : nullptr), Bm, dE); H2D(2, c, B, dc); H2D(3, x, Bn, dx); H2D(4, y, Bn + Bm, dy);
    if (m == 0) DEVOUT(1, 8, dE);
    pmpc_status st = pmpc_qp_ruiz_unscale_batch_dev(ctx, B, n, m, dD, dE, dc, dx, dy);
    if (st != PMPC_OK) return st;
    D2H_(x, dx, Bn); D2H_(y, dy, Bn + Bm);
#undef D2H_
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PMPC_OK;
}

#define DISPATCH_MODEL(model, F, ...)                                            \
    switch (model) {                                                             \
        case PMPC_MODEL_ROBOT: return F<RobotOCP>(__VA_ARGS__);                  \
        case PMPC_MODEL_CSTR: return F<CstrOCP>(__VA_ARGS__);                    \
        case PMPC_MODEL_PARKING: return F<ParkingOCP>(__VA_ARGS__);              \
        case PMPC_MODEL_ROBOT_NG: return F<RobotNGOCP>(__VA_ARGS__);             \
        case PMPC_MODEL_KITE_STANDIN: return F<KiteStandInOCP>(__VA_ARGS__);     \
        case PMPC_MODEL_PARKING_NG: return F<ParkingNGOCP>(__VA_ARGS__);         \
        default: return PMPC_ERR_UNKNOWN_MODEL;                                  \
    }

}  // extern "C"

template <class Model>
static pmpc_status dims_impl(int P, int S, int* nx, int* nu, int* np, int* nd, int* ng, int* n, int* me, int* mi) {
    if (P < 1 || P > MAX_P || S < 1 || P * S + 1 > MAX_NODES) return PMPC_ERR_UNSUPPORTED_SIZE;
    OcpDims<Model> dm(P, S);
    if (nx) *nx = Model::NX; if (nu) *nu = Model::NU; if (np) *np = Model::NP; if (nd) *nd = Model::ND; if (ng) *ng = Model::NG;
    if (n) *n = dm.n; if (me) *me = dm.me; if (mi) *mi = dm.mi;
    return PMPC_OK;
}

extern "C" {

pmpc_status pmpc_ocp_dims(int model, int P, int S, int* nx, int* nu, int* np, int* nd, int* ng, int* var_size, int* num_eq, int* num_ineq) {
    DISPATCH_MODEL(model, dims_impl, P, S, nx, nu, np, nd, ng, var_size, num_eq, num_ineq);
}

pmpc_status pmpc_ocp_linearise_batch(pmpc_context* ctx, int model, int P, int S, double t0, double tf, const double* mparams,
                                     int n_mparams, int B, const double* var, const double* d, const double* lam, double* cost,
                                     double* constr, double* jac, double* cost_grad, double* lag_grad, double* lag_hess) {
    if (!ctx || B < 1 || !var) return PMPC_ERR_INVALID_ARGUMENT;
    { int nd = 0; const pmpc_status ds = pmpc_ocp_dims(model, P, S, nullptr, nullptr, nullptr, &nd, nullptr, nullptr, nullptr, nullptr);
      if (ds != PMPC_OK) return ds;
      if (nd > 0 && !d) return PMPC_ERR_INVALID_ARGUMENT; }
    HIPCHK(hipSetDevice(ctx->device));
    DISPATCH_MODEL(model, linearise_impl, ctx, P, S, t0, tf, mparams, n_mparams, B, var, d, lam, cost, constr, jac, cost_grad, lag_grad, lag_hess);
}

/* arguments every SQP entry point shares: the static parameters `d` are mandatory for models that have them (ND > 0), and at least one
 * SQP iteration must be allowed (max_iter <= 0 would launch nothing and leave the outputs unwritten) */
static pmpc_status check_sqp_args(int model, int P, int S, const double* d, const pmpc_sqp_settings* ss) {
    int nd = 0;
    const pmpc_status st = pmpc_ocp_dims(model, P, S, nullptr, nullptr, nullptr, &nd, nullptr, nullptr, nullptr, nullptr);
    if (st != PMPC_OK) return st;
    if (nd > 0 && !d) return PMPC_ERR_INVALID_ARGUMENT;
    if (ss && ss->max_iter < 1) return PMPC_ERR_INVALID_ARGUMENT;
    if (ss && ss->iteration_trace && ss->iteration_trace_capacity < 1) return PMPC_ERR_INVALID_ARGUMENT;
    if (ss && (ss->kkt_form < 0 || ss->kkt_form > 2)) return PMPC_ERR_INVALID_ARGUMENT;
    return PMPC_OK;
}

pmpc_status pmpc_sqp_solve_batch_dev(pmpc_context* ctx, int model, int P, int S, double t0, double tf, const double* mparams,
                                     int n_mparams, int B, const double* x_guess, const double* lam_guess, const double* d,
                                     const double* lbx, const double* ubx, const double* lbg, const double* ubg,
                                     const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* x, double* lam,
                                     pmpc_sqp_info* info) {
    if (!ctx || B < 0 || !lbx || !ubx || !ss || !qs || !x || !lam || !info) return PMPC_ERR_INVALID_ARGUMENT;
    if (ss->regularisation < 0 || ss->regularisation > 2) return PMPC_ERR_INVALID_ARGUMENT;
    { const pmpc_status ca = check_sqp_args(model, P, S, d, ss); if (ca != PMPC_OK) return ca; }
    if (B == 0) return PMPC_OK;
    HIPCHK(hipSetDevice(ctx->device));
    DISPATCH_MODEL(model, sqp_builtin_dev, ctx, P, S, t0, tf, mparams, n_mparams, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss, qs, x, lam, info);
}

/* MPC façade, batched (mpc_wrapper.hpp:89-93 initial_conditions, :298 solve, :241-244 solution_u_at): one receding-horizon step */
__global__ void mpc_pin_initial_state_kernel(int B, int n, int varx, int nx, const double* __restrict__ x0, double* __restrict__ lbx,
                                             double* __restrict__ ubx) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * nx) return;
    const int b = idx / nx, q = idx - b * nx;
    const size_t e = (size_t)b * n + varx - nx + q;   // the LAST nx entries of the x block are the state at t_start
    lbx[e] = x0[idx]; ubx[e] = x0[idx];
}
__global__ void mpc_first_control_kernel(int B, int n, int varx, int nu, int nn, const double* __restrict__ x, double* __restrict__ u0) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * nu) return;
    const int b = idx / nu, i = idx - b * nu;
    u0[idx] = x[(size_t)b * n + varx + (size_t)(nn - 1) * nu + i];
}
pmpc_status pmpc_mpc_step_batch_dev(pmpc_context* ctx, int model, int P, int S, double t0, double tf, const double* mparams, int n_mparams,
                                    int B, const double* x0, const double* d, double* lbx, double* ubx, const double* lbg,
                                    const double* ubg, const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* x, double* lam,
                                    pmpc_sqp_info* info, double* u0) {
    if (!ctx || B < 0 || !x0 || !lbx || !ubx || !ss || !qs || !x || !lam || !info) return PMPC_ERR_INVALID_ARGUMENT;
    if (B == 0) return PMPC_OK;
    int nx, nu, np, nd, ng, n, me, mi;
    pmpc_status st = pmpc_ocp_dims(model, P, S, &nx, &nu, &np, &nd, &ng, &n, &me, &mi);
    if (st != PMPC_OK) return st;
    if ((nd > 0 && !d) || ss->max_iter < 1) return PMPC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(ctx->device));
    const int m = me + mi, nn = P * S + 1, varx = nx * nn;
    hipLaunchKernelGGL(mpc_pin_initial_state_kernel, dim3((B * nx + 255) / 256), dim3(256), 0, ctx->stream, B, n, varx, nx, x0, lbx, ubx);
    // warm start: the previous solution is the guess (SQPBase::solve() starts from m_x / m_lam, sqp_base.hpp:569-581); the
    // kernel's guess and result pointers must not alias, so the guess is a device-to-device copy
    double *xg, *lg;
    DEVOUT(12, (size_t)B * n * sizeof(double), xg); DEVOUT(13, (size_t)B * (m + n) * sizeof(double), lg);
    HIPCHK(hipMemcpyAsync(xg, x, (size_t)B * n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(lg, lam, (size_t)B * (m + n) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    st = pmpc_sqp_solve_batch_dev(ctx, model, P, S, t0, tf, mparams, n_mparams, B, xg, lg, d, lbx, ubx, lbg, ubg, ss, qs, x, lam, info);
    if (st != PMPC_OK) return st;
    if (u0) hipLaunchKernelGGL(mpc_first_control_kernel, dim3((B * nu + 255) / 256), dim3(256), 0, ctx->stream, B, n, varx, nu, nn, x, u0);
    HIPCHK(hipGetLastError());
    return PMPC_OK;
}

/* ---- a batch of MPC controllers whose state lives on the device between steps (host-side callers without HIP) ---- */
struct pmpc_mpc_batch {
    pmpc_context* ctx; int model, P, S, B, nx, nu, nd, n, m, mi; double t0, tf; std::vector<double> mparams;
    double *d = nullptr, *lbx = nullptr, *ubx = nullptr, *lbg = nullptr, *ubg = nullptr, *x = nullptr, *lam = nullptr, *x0 = nullptr, *u0 = nullptr;
    pmpc_sqp_info* info = nullptr;
};
pmpc_status pmpc_mpc_batch_destroy(pmpc_mpc_batch* h) {
    if (!h) return PMPC_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    for (void* q : {(void*)h->d, (void*)h->lbx, (void*)h->ubx, (void*)h->lbg, (void*)h->ubg, (void*)h->x, (void*)h->lam, (void*)h->x0, (void*)h->u0, (void*)h->info})
        if (q) (void)hipFree(q);
    delete h;
    return PMPC_OK;
}
pmpc_status pmpc_mpc_batch_create(pmpc_context* ctx, int model, int P, int S, double t0, double tf, const double* mparams, int n_mparams, int B,
                                  const double* d, const double* lbx, const double* ubx, const double* lbg, const double* ubg,
                                  const double* x_guess, const double* lam_guess, pmpc_mpc_batch** out) {
    if (!ctx || !out || B < 1 || !lbx || !ubx) return PMPC_ERR_INVALID_ARGUMENT;
    int nx, nu, np, nd, ng, n, me, mi;
    pmpc_status st = pmpc_ocp_dims(model, P, S, &nx, &nu, &np, &nd, &ng, &n, &me, &mi);
    if (st != PMPC_OK) return st;
    if ((nd > 0 && !d) || (mi > 0 && (!lbg || !ubg))) return PMPC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(ctx->device));
    pmpc_mpc_batch* h = new pmpc_mpc_batch{ctx, model, P, S, B, nx, nu, nd, n, me + mi, mi, t0, tf, std::vector<double>(mparams ? mparams : nullptr, mparams ? mparams + n_mparams : nullptr)};
    const size_t Bn = (size_t)B * n, Bd = (size_t)B * (n + h->m);
    auto up = [&](double** dst, const double* src, size_t count, bool zero) -> bool {
        if (hipMalloc((void**)dst, (count ? count : 1) * sizeof(double)) != hipSuccess) return false;
        if (src) return hipMemcpyAsync(*dst, src, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
        return !zero || hipMemsetAsync(*dst, 0, (count ? count : 1) * sizeof(double), ctx->stream) == hipSuccess;
    };
    bool ok = up(&h->d, nd ? d : nullptr, (size_t)B * nd, true) && up(&h->lbx, lbx, Bn, false) && up(&h->ubx, ubx, Bn, false) &&
              up(&h->x, x_guess, Bn, true) && up(&h->lam, lam_guess, Bd, true) && up(&h->x0, nullptr, (size_t)B * nx, true) &&
              up(&h->u0, nullptr, (size_t)B * nu, true) && hipMalloc((void**)&h->info, (size_t)B * sizeof(pmpc_sqp_info)) == hipSuccess;
    if (ok && mi > 0) ok = up(&h->lbg, lbg, (size_t)B * mi, false) && up(&h->ubg, ubg, (size_t)B * mi, false);
    if (ok) ok = hipStreamSynchronize(ctx->stream) == hipSuccess;
    if (!ok) { (void)pmpc_mpc_batch_destroy(h); return PMPC_ERR_HIP; }
    *out = h;
    return PMPC_OK;
}
pmpc_status pmpc_mpc_batch_step(pmpc_mpc_batch* h, const double* x0, const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* u0,
                                pmpc_sqp_info* info) {
    if (!h || !x0 || !ss || !qs) return PMPC_ERR_INVALID_ARGUMENT;
    pmpc_context* ctx = h->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(h->x0, x0, (size_t)h->B * h->nx * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    pmpc_status st = pmpc_mpc_step_batch_dev(ctx, h->model, h->P, h->S, h->t0, h->tf, h->mparams.empty() ? nullptr : h->mparams.data(), (int)h->mparams.size(),
                                             h->B, h->x0, h->d, h->lbx, h->ubx, h->lbg, h->ubg, ss, qs, h->x, h->lam, h->info, h->u0);
    if (st != PMPC_OK) return st;
    if (u0) HIPCHK(hipMemcpyAsync(u0, h->u0, (size_t)h->B * h->nu * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (info) HIPCHK(hipMemcpyAsync(info, h->info, (size_t)h->B * sizeof(pmpc_sqp_info), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PMPC_OK;
}
pmpc_status pmpc_mpc_batch_solution(pmpc_mpc_batch* h, double* x, double* lam) {
    if (!h) return PMPC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(h->ctx->device));
    if (x) HIPCHK(hipMemcpyAsync(x, h->x, (size_t)h->B * h->n * sizeof(double), hipMemcpyDeviceToHost, h->ctx->stream));
    if (lam) HIPCHK(hipMemcpyAsync(lam, h->lam, (size_t)h->B * (h->n + h->m) * sizeof(double), hipMemcpyDeviceToHost, h->ctx->stream));
    HIPCHK(hipStreamSynchronize(h->ctx->stream));
    return PMPC_OK;
}

pmpc_status pmpc_sqp_solve_batch(pmpc_context* ctx, int model, int P, int S, double t0, double tf, const double* mparams,
                                 int n_mparams, int B, const double* x_guess, const double* lam_guess, const double* d,
                                 const double* lbx, const double* ubx, const double* lbg, const double* ubg,
                                 const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* x, double* lam, pmpc_sqp_info* info) {
    if (!ctx || B < 0 || !lbx || !ubx || !ss || !qs || !x || !lam || !info) return PMPC_ERR_INVALID_ARGUMENT;
    if (B == 0) return PMPC_OK;
    int nx, nu, np, nd, ng, n, me, mi;
    pmpc_status st = pmpc_ocp_dims(model, P, S, &nx, &nu, &np, &nd, &ng, &n, &me, &mi);
    if (st != PMPC_OK) return st;
    if (nd > 0 && !d) return PMPC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(ctx->device));
    const int m = me + mi;
    double *dxg, *dlg, *dd, *dlbx, *dubx, *dlbg, *dubg, *dx, *dlam; pmpc_sqp_info* dinfo;
    H2D(12, x_guess, (size_t)B * n, dxg); H2D(13, lam_guess, (size_t)B * (m + n), dlg); H2D(14, (nd ? d : nullptr), (size_t)B * nd, dd);
    if (!nd) DEVOUT(14, 8, dd);
    H2D(15, lbx, (size_t)B * n, dlbx); H2D(16, ubx, (size_t)B * n, dubx);
    H2D(17, (mi ? lbg : nullptr), (size_t)B * mi, dlbg); H2D(18, (mi ? ubg : nullptr), (size_t)B * mi, dubg);
    DEVOUT(19, (size_t)B * n * sizeof(double), dx); DEVOUT(20, (size_t)B * (m + n) * sizeof(double), dlam);
    DEVOUT(21, (size_t)B * sizeof(pmpc_sqp_info), dinfo);
    st = pmpc_sqp_solve_batch_dev(ctx, model, P, S, t0, tf, mparams, n_mparams, B, dxg, dlg, dd, dlbx, dubx, dlbg, dubg, ss, qs, dx, dlam, dinfo);
    if (st != PMPC_OK) return st;
    HIPCHK(hipMemcpyAsync(x, dx, (size_t)B * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(lam, dlam, (size_t)B * (m + n) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(info, dinfo, (size_t)B * sizeof(pmpc_sqp_info), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PMPC_OK;
}


/* contiguous shards over n_ctx contexts, one PERSISTENT host thread per context (started on the context's first sharded call, joined
   by pmpc_destroy), no collective. The contexts must be distinct objects (two contexts on one device are fine; one context twice is not: its
   stream, workspace and staging buffers serve one call at a time). */
pmpc_status pmpc_sqp_solve_batch_multi(pmpc_context* const* ctxs, int n_ctx, int model, int P, int S, double t0, double tf, const double* mparams,
                                       int n_mparams, int B, const double* x_guess, const double* lam_guess, const double* d, const double* lbx,
                                       const double* ubx, const double* lbg, const double* ubg, const pmpc_sqp_settings* ss,
                                       const pmpc_qp_settings* qs, double* x, double* lam, pmpc_sqp_info* info) {
    if (!ctxs || n_ctx < 1 || B < 0 || !lbx || !ubx || !ss || !qs || !x || !lam || !info) return PMPC_ERR_INVALID_ARGUMENT;
    for (int k = 0; k < n_ctx; ++k) {
        if (!ctxs[k]) return PMPC_ERR_INVALID_ARGUMENT;
        for (int j = 0; j < k; ++j) if (ctxs[j] == ctxs[k]) return PMPC_ERR_INVALID_ARGUMENT;
    }
    if (ss->filter_state || ss->iteration_trace) return PMPC_ERR_INVALID_ARGUMENT;   // device buffers of one context
    int nx, nu, np, nd, ng, n, me, mi;
    const pmpc_status ds = pmpc_ocp_dims(model, P, S, &nx, &nu, &np, &nd, &ng, &n, &me, &mi);
    if (ds != PMPC_OK) return ds;
    if (B == 0) return PMPC_OK;
    const int m = me + mi;
    // The status slots and the list of posted shards live OUTSIDE the try block and are sized before anything is posted: a worker writes its slot after
    // this function's catch handler would have run, and whatever fails later (a worker that cannot be created, the allocation inside post()), every
    // shard already posted is waited for before the function returns — its thread writes x / lam / info of the caller and its status slot.
    std::vector<pmpc_status> st;
    std::vector<int> posted;
    try { st.assign((size_t)n_ctx, PMPC_OK); posted.reserve((size_t)n_ctx); } catch (...) { return PMPC_ERR_HIP; }
    pmpc_status failed = PMPC_OK;
    try {   // nothing may leave an extern "C" function by exception (std::bad_alloc, std::system_error from a thread that cannot be created)
        for (int k = 0; k < n_ctx; ++k) {
            if ((long long)B * (k + 1) / n_ctx <= (long long)B * k / n_ctx) continue;
            if (!ctxs[k]->shard_worker) ctxs[k]->shard_worker = new ShardWorker();
            if (!ctxs[k]->shard_worker->start()) { failed = PMPC_ERR_HIP; break; }
            const long long b0 = (long long)B * k / n_ctx, b1 = (long long)B * (k + 1) / n_ctx;
            pmpc_status* out = &st[k];
            pmpc_context* ctx = ctxs[k];
            ctx->shard_worker->post([=]() {
                auto at = [&](const double* p, size_t per) { return p ? p + (size_t)b0 * per : nullptr; };
                *out = pmpc_sqp_solve_batch(ctx, model, P, S, t0, tf, mparams, n_mparams, (int)(b1 - b0), at(x_guess, n), at(lam_guess, m + n),
                                            at(d, nd), at(lbx, n), at(ubx, n), at(lbg, mi), at(ubg, mi), ss, qs, x + (size_t)b0 * n,
                                            lam + (size_t)b0 * (m + n), info + b0);
            });
            posted.push_back(k);   // (capacity reserved above: cannot throw)
        }
    } catch (...) {
        failed = PMPC_ERR_HIP;
    }
    for (int k : posted) ctxs[k]->shard_worker->wait();
    if (failed != PMPC_OK) return failed;
    for (int k = 0; k < n_ctx; ++k) if (st[k] != PMPC_OK) return st[k];
    return PMPC_OK;
}

/* Host-buffer wrapper around a user-registered OCP's device entry (PMPC_REGISTER_OCP): stage in, launch, stage out. */
pmpc_status pmpc_sqp_solve_batch_user(pmpc_context* ctx, pmpc_sqp_dev_fn fn, const void* model, int nx, int nu, int np, int nd, int ng,
                                      int P, int S, double t0, double tf, int B, const double* x_guess, const double* lam_guess,
                                      const double* d, const double* lbx, const double* ubx, const double* lbg, const double* ubg,
                                      const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* x, double* lam, pmpc_sqp_info* info) {
    if (!ctx || !fn || !model || B < 0 || !lbx || !ubx || !ss || !qs || !x || !lam || !info) return PMPC_ERR_INVALID_ARGUMENT;
    if (B == 0) return PMPC_OK;
    if (nd > 0 && !d) return PMPC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(ctx->device));
    const int nn = P * S + 1, n = (nx + nu) * nn + np, me = nx * nn, mi = ng * nn, m = me + mi;
    double *dxg, *dlg, *dd, *dlbx, *dubx, *dlbg, *dubg, *dx, *dlam; pmpc_sqp_info* dinfo;
    H2D(12, x_guess, (size_t)B * n, dxg); H2D(13, lam_guess, (size_t)B * (m + n), dlg); H2D(14, (nd ? d : nullptr), (size_t)B * nd, dd);
    if (!nd) DEVOUT(14, 8, dd);
    H2D(15, lbx, (size_t)B * n, dlbx); H2D(16, ubx, (size_t)B * n, dubx);
    H2D(17, (mi ? lbg : nullptr), (size_t)B * mi, dlbg); H2D(18, (mi ? ubg : nullptr), (size_t)B * mi, dubg);
    DEVOUT(19, (size_t)B * n * sizeof(double), dx); DEVOUT(20, (size_t)B * (m + n) * sizeof(double), dlam);
    DEVOUT(21, (size_t)B * sizeof(pmpc_sqp_info), dinfo);
    pmpc_status st = fn(ctx, model, P, S, t0, tf, B, dxg, dlg, dd, dlbx, dubx, dlbg, dubg, ss, qs, dx, dlam, dinfo);
    if (st != PMPC_OK) return st;
    HIPCHK(hipMemcpyAsync(x, dx, (size_t)B * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(lam, dlam, (size_t)B * (m + n) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(info, dinfo, (size_t)B * sizeof(pmpc_sqp_info), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PMPC_OK;
}

}  // extern "C"
