// Host build of odam_amd/csrc/sq_math.h compared against the host libm (test infrastructure).
// g++ -O2 -ffp-contract=off -fopenmp -shared -fPIC
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "../../odam_amd/csrc/sq_math.h"
using namespace odam_math;

static inline bool same(float a, float b) { return f2u(a) == f2u(b) || (a != a && b != b); }

extern "C" {
// sweep every `stride`-th float bit pattern in [lo_bits, hi_bits] (both signs); which: 0 sin 1 cos 2 exp 3 log
long sq_math_sweep1(int which, uint32_t lo_bits, uint32_t hi_bits, uint32_t stride, uint32_t* first_bad) {
    long bad = 0;
    uint32_t fb = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t b = lo_bits; b <= (int64_t)hi_bits; b += stride) {
        for (int sgn = 0; sgn < 2; sgn++) {
            float x = u2f((uint32_t)b | (sgn ? 0x80000000u : 0u));
            float mine, ref;
            switch (which) {
                case 0: mine = sinf_(x); ref = sinf(x); break;
                case 1: mine = cosf_(x); ref = cosf(x); break;
                case 2: mine = expf_(x); ref = expf(x); break;
                default: if (sgn) continue; mine = logf_pos(x); ref = logf(x); break;
            }
            if (!same(mine, ref)) { bad++; fb = f2u(x); }
        }
    }
    if (first_bad) *first_bad = fb;
    return bad;
}
// powf: every `stride`-th base bit pattern in [lo,hi] against n_y exponents
long sq_math_sweep_pow(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride, const float* ys, int n_y) {
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t b = lo_bits; b <= (int64_t)hi_bits; b += stride) {
        float x = u2f((uint32_t)b);
        for (int j = 0; j < n_y; j++)
            if (!same(powf_pos(x, ys[j]), powf(x, ys[j]))) bad++;
    }
    return bad;
}
void sq_math_eval(int which, const float* x, const float* y, float* out, int n) {
    for (int i = 0; i < n; i++) switch (which) {
        case 0: out[i] = sinf_(x[i]); break;
        case 1: out[i] = cosf_(x[i]); break;
        case 2: out[i] = expf_(x[i]); break;
        case 3: out[i] = logf_pos(x[i]); break;
        default: out[i] = powf_pos(x[i], y[i]); break;
    }
}
}
