// CPU check of the corrected quotient used by the BFGS update (polympc_amd/csrc/pmpc_qp.hpp, UniformDiv): with y = RN(1/b),
//   q0 = RN(a*y), q1 = fma(fma(-q0, b, a), y, q0), q = fma(fma(-q1, b, a), y, q1)
// must equal the IEEE quotient a / b bit for bit. Random significands, exponents within +-60 of each other, plus
// structured hard cases (significands near 1 and near 2). Built without contraction; fma is explicit.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
static inline double corrected(double a, double b) {
    const double y = 1.0 / b;
    const double q0 = a * y;
    const double q1 = std::fma(std::fma(-q0, b, a), y, q0);
    return std::fma(std::fma(-q1, b, a), y, q1);
}
static inline double mk(uint64_t frac, int e, bool neg) {
    uint64_t u = (frac & 0x000FFFFFFFFFFFFFull) | ((uint64_t)(1023 + e) << 52) | ((uint64_t)neg << 63);
    double d; memcpy(&d, &u, 8); return d;
}
int main(int argc, char** argv) {
    const long N = argc > 1 ? atol(argv[1]) : 20000000L;
    std::mt19937_64 g(20260929);
    long bad = 0;
    for (long t = 0; t < N; ++t) {
        const uint64_t ra = g(), rb = g(), rc = g();
        uint64_t fa = ra, fb = rb;
        if ((rc & 7) == 0) fa = (ra & 0xFFF);                       // significand just above 1
        if ((rc & 7) == 1) fb = 0x000FFFFFFFFFFFFFull - (rb & 0xFFF); // significand just below 2
        if ((rc & 7) == 2) { fa = 0x000FFFFFFFFFFFFFull - (ra & 0xFFF); fb = (rb & 0xFFF); }
        const double a = mk(fa, (int)((rc >> 8) % 121) - 60, (rc >> 20) & 1), b = mk(fb, (int)((rc >> 32) % 121) - 60, (rc >> 44) & 1);
        if (corrected(a, b) != a / b) ++bad;
    }
    printf("uniform_div_check: %ld trials, %ld mismatches\n", N, bad);
    return bad ? 1 : 0;
}
