// Developer probe (not a test): how many bytes per cycle ONE wavefront streams from its own HBM region, by load width and by the number of loads it
// keeps in flight — the bound on the triangular passes of the large-instance kernel (pmpc_qp_big.hpp: one wavefront per QP walks a 328 KB factor
// twice per ADMM iteration). Build: hipcc --offload-arch=gfx950 -O3 -o wave_stream_probe wave_stream_probe.hip ; run: ./wave_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using d2 = double __attribute__((ext_vector_type(2)));

// W = 8: 16 loads of 8 B per lane per unit (8 KB); W = 16: 8 loads of 16 B per lane per unit (8 KB). DEPTH units in flight.
template <int W, int DEPTH>
__global__ __launch_bounds__(64) void stream(const double* base, size_t region_doubles, int units, int reps, double* sink, long long* cyc) {
    const int ln = threadIdx.x;
    const double* p = base + (size_t)blockIdx.x * region_doubles;
    constexpr int NL = W == 8 ? 16 : 8;
    double acc = 0.0;
    long long t0 = clock64();
    for (int rep = 0; rep < reps; ++rep) {
        double buf[DEPTH][16];
        auto load = [&](int u, double (&b)[16]) {
            const double* q = p + ((size_t)rep * units + u) * 1024;   // (every unit is touched once: nothing comes back from L2 / the Infinity Cache)
            if constexpr (W == 8) {
#pragma unroll
                for (int k = 0; k < 16; ++k) b[k] = q[k * 64 + ln];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const d2 v = reinterpret_cast<const d2*>(q)[k * 64 + ln]; b[2 * k] = v[0]; b[2 * k + 1] = v[1]; }
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) load(d, buf[d]);
        for (int u0 = 0; u0 < units; u0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = fma(buf[d][k], 1.0000001, acc);
                const int un = u0 + d + DEPTH;
                load(un < units ? un : 0, buf[d]);
            }
        }
    }
    long long t1 = clock64();
    if (ln == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + ln] = acc;
}

template <int W, int DEPTH>
void run(const double* d_base, size_t region, int units, int grid, double* d_sink, long long* d_cyc) {
    const int reps = 20;
    hipLaunchKernelGGL((stream<W, DEPTH>), dim3(grid), dim3(64), 0, 0, d_base, region, units, reps, d_sink, d_cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<W, DEPTH>), dim3(grid), dim3(64), 0, 0, d_base, region, units, reps, d_sink, d_cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(grid);
    hipMemcpy(c.data(), d_cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : c) mean += (double)v; mean /= grid;
    const double bytes = (double)units * 8192.0 * reps;
    printf("width %2d B/lane depth %d grid %5d: %8.0f cycles per 8 KB unit, %6.2f B/cycle/wave, %7.1f GB/s aggregate (%.3f ms)\n", W, DEPTH, grid,
           mean / ((double)units * reps), bytes / mean, bytes * grid / (ms * 1e6), ms);
}

int main() {
    const int units = 40;                       // 320 KB per wavefront, as one triangular pass of config C
    const size_t region = 1u << 20;             // 8 MB per wavefront: 20 passes over distinct memory
    const int maxgrid = 2048;
    double* d_base; double* d_sink; long long* d_cyc;
    hipMalloc(&d_base, region * maxgrid * sizeof(double));
    hipMemset(d_base, 0, region * maxgrid * sizeof(double));
    hipMalloc(&d_sink, maxgrid * 64 * sizeof(double));
    hipMalloc(&d_cyc, maxgrid * sizeof(long long));
    for (int grid : {1, 128, 256, 1024, 2048}) {
        run<8, 1>(d_base, region, units, grid, d_sink, d_cyc);
        run<8, 2>(d_base, region, units, grid, d_sink, d_cyc);
        run<8, 3>(d_base, region, units, grid, d_sink, d_cyc);
        run<16, 1>(d_base, region, units, grid, d_sink, d_cyc);
        run<16, 2>(d_base, region, units, grid, d_sink, d_cyc);
        run<16, 4>(d_base, region, units, grid, d_sink, d_cyc);
        run<16, 6>(d_base, region, units, grid, d_sink, d_cyc);
    }
    return 0;
}
