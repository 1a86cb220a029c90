// Calibrates rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for THIS project's access pattern: 8-byte-per-lane coalesced
// loads/stores (global_load_dwordx2 / global_store_dwordx2), as MI355X_MICROARCH.md §HBM prescribes. Reads 1 GiB, writes
// 0.5 GiB of fresh (never cached) memory; compare the counters with these known byte counts.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void stream_rw(const double* __restrict__ in, double* __restrict__ out, size_t n_in, size_t n_out) {
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nth = gridDim.x * (size_t)blockDim.x;
    double acc = 0.0;
    for (size_t i = tid; i < n_in; i += nth) acc += in[i];
    for (size_t i = tid; i < n_out; i += nth) out[i] = acc + (double)i;
}
int main() {
    const size_t n_in = (1ull << 30) / 8, n_out = (1ull << 29) / 8;
    double *in, *out;
    if (hipMalloc(&in, n_in * 8) != hipSuccess || hipMalloc(&out, n_out * 8) != hipSuccess) return 1;
    (void)hipMemset(in, 0, n_in * 8);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(stream_rw, dim3(4096), dim3(256), 0, 0, in, out, n_in, n_out);
    (void)hipDeviceSynchronize();
    printf("stream_rw: read %zu bytes, wrote %zu bytes\n", n_in * 8, n_out * 8);
    return 0;
}
