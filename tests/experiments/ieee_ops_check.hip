// Developer experiment (run on an MI355X): are the device's fp64 sqrt, division and reciprocal the correctly rounded IEEE results the
// host (x86-64 SSE2) computes? The bit-for-bit GPU-vs-CPU parity of the engine rests on add / mul / fma / div / sqrt being identical on
// both sides. 2^22 seeded operands per operation, results computed on the device, compared on the host. Expected: 0 mismatches each.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
static uint64_t splitmix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__global__ void ops(int n, const double* a, const double* b, double* sq, double* dv, double* rc, double* fm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sq[i] = ::sqrt(fabs(a[i]));
    dv[i] = a[i] / b[i];
    rc[i] = 1.0 / b[i];
    fm[i] = fma(a[i], b[i], sq[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> a(n), b(n), sq(n), dv(n), rc(n), fm(n);
    for (int i = 0; i < n; ++i) {
        const uint64_t r = splitmix(2 * (uint64_t)i), r2 = splitmix(2 * (uint64_t)i + 1);
        const int ea = (int)(splitmix(r) % 201) - 100, eb = (int)(splitmix(r2) % 201) - 100;
        uint64_t ba = (r & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(ea + 1023) << 52), bb = (r2 & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(eb + 1023) << 52);
        memcpy(&a[i], &ba, 8); memcpy(&b[i], &bb, 8);
    }
    double *da, *db, *d1, *d2, *d3, *d4;
    (void)hipMalloc(&da, n * 8); (void)hipMalloc(&db, n * 8); (void)hipMalloc(&d1, n * 8); (void)hipMalloc(&d2, n * 8); (void)hipMalloc(&d3, n * 8); (void)hipMalloc(&d4, n * 8);
    (void)hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(ops, dim3(n / 256), dim3(256), 0, 0, n, da, db, d1, d2, d3, d4);
    (void)hipMemcpy(sq.data(), d1, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(dv.data(), d2, n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(rc.data(), d3, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(fm.data(), d4, n * 8, hipMemcpyDeviceToHost);
    long ms = 0, md = 0, mr = 0, mf = 0;
    for (int i = 0; i < n; ++i) {
        const double hs = std::sqrt(std::fabs(a[i])), hd = a[i] / b[i], hr = 1.0 / b[i], hf = std::fma(a[i], b[i], hs);
        ms += memcmp(&hs, &sq[i], 8) != 0; md += memcmp(&hd, &dv[i], 8) != 0; mr += memcmp(&hr, &rc[i], 8) != 0; mf += memcmp(&hf, &fm[i], 8) != 0;
    }
    printf("device vs host on %d operand pairs: sqrt %ld, a/b %ld, 1/b %ld, fma %ld mismatches\n", n, ms, md, mr, mf);
    return (ms || md || mr || mf) ? 1 : 0;
}
