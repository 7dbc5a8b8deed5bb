// tools/fir_int_check.cpp -- TEST TOOL (CPU): the integer forms of the two FIR stages (fir_h_int / fir_v_int,
// h2y_math.h) against the float forms that restate Subsample444to420_FIR (convert.cpp:305-317, :365-374), for every
// bit depth the integer form is used at (8..14): random samples, all-equal samples at every code value, extremes
// (alternating 0 / maxCV in every tap pattern).  Prints the number of mismatches; exit code 1 if any.
//   g++ -O2 -ffp-contract=off -I hdr2yuv_amd/csrc tools/fir_int_check.cpp -o build/fir_int_check
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "h2y_math.h"
using namespace h2y;

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    long bad = 0, total = 0;
    for (int depth = 8; depth <= H2Y_FIR_INT_MAX_DEPTH; depth++) {
        const int maxcv = (1 << depth) - 1;
        const float fmax = (float)maxcv;
        int v[12];
        auto check = [&]() {
            const uint32_t a = fir_h((float)v[0], (float)v[1], (float)v[2], (float)v[3], (float)v[4], (float)v[5], (float)v[6], fmax);
            const uint32_t b = fir_h_int(v[0], v[1], v[2], v[3], v[4], v[5], v[6], maxcv);
            const uint32_t c = fir_v((float)v[0], (float)v[1], (float)v[2], (float)v[3], (float)v[4], (float)v[5], (float)v[6], (float)v[7],
                                     (float)v[8], (float)v[9], (float)v[10], (float)v[11], fmax);
            const uint32_t d = fir_v_int(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], 0, maxcv);
            total += 2;
            if (a != b) bad++;
            if (c != d) bad++;
        };
        for (long i = 0; i < n; i++) {
            const int mode = (int)(rnd() % 4u);
            for (int k = 0; k < 12; k++) {
                const uint32_t r = rnd();
                v[k] = mode == 0 ? (int)(r % (uint32_t)(maxcv + 1)) : mode == 1 ? ((r & 1u) ? maxcv : 0) : mode == 2 ? maxcv - (int)(r % 8u) : (int)(r % 8u);
            }
            check();
        }
        for (int pat = 0; pat < 4096; pat++) { // every 0 / maxCV pattern over the twelve taps
            for (int k = 0; k < 12; k++) v[k] = ((pat >> k) & 1) ? maxcv : 0;
            check();
        }
        for (int cv = 0; cv <= maxcv; cv++) { // flat areas
            for (int k = 0; k < 12; k++) v[k] = cv;
            check();
        }
    }
    printf("%ld comparisons, %ld mismatches\n", total, bad);
    return bad ? 1 : 0;
}
