// Probe: layout and rounding order of v_mfma_f64_16x16x4_f64 on gfx950 (developer experiment, not part of the product).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, const double* C, double* D) {
    const int l = threadIdx.x;
    // A is 16x4 row-major (A[i*4+k]), B is 4x16 row-major (B[k*16+j]), C/D 16x16 row-major
    const double a = A[(l & 15) * 4 + (l >> 4)];
    const double b = B[(l >> 4) * 16 + (l & 15)];
    double4_t c;
    for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];
}
int main() {
    std::vector<double> A(64), B(64), C(256), D(256);
    srand(1);
    auto rnd = []() { return (rand() / (double)RAND_MAX - 0.5) * 4.0; };
    for (auto& v : A) v = rnd(); for (auto& v : B) v = rnd(); for (auto& v : C) v = rnd() * 1e-3;
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    int exact_fwd = 0, exact_rev = 0, close = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double f = C[i * 16 + j]; for (int kk = 0; kk < 4; ++kk) f = fma(A[i * 4 + kk], B[kk * 16 + j], f);
        double r = C[i * 16 + j]; for (int kk = 3; kk >= 0; --kk) r = fma(A[i * 4 + kk], B[kk * 16 + j], r);
        exact_fwd += (D[i * 16 + j] == f); exact_rev += (D[i * 16 + j] == r); close += (fabs(D[i * 16 + j] - f) < 1e-12);
    }
    printf("mfma_f64_16x16x4: bitwise == k-ascending fma chain: %d/256, == k-descending: %d/256, within 1e-12: %d/256\n", exact_fwd, exact_rev, close);
    return 0;
}
