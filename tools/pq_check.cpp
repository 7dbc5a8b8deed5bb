/*
 * tools/pq_check.cpp -- TEST TOOL (host build of hdr2yuv_amd/csrc/h2y_math.h).
 *
 * Checks the two PQ tiers of the device arithmetic against the reference
 * formula evaluated with this machine's libm (what convert.cpp:56-63 does):
 *
 *   pq_check pow   N            random pow_dd() vs libm pow(), both exponents
 *   pq_check range LO HI [T]    every float with bit pattern in [LO,HI):
 *                               fast tier (+ slow when flagged) vs libm PQ;
 *                               reports mismatches, slow-tier rate and the
 *                               largest fast-tier error in double ulps
 *   pq_check slow  LO HI [T]    same range, slow tier only
 *
 * Build: g++ -O2 -ffp-contract=off -mfma -std=c++17 -pthread
 */
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../hdr2yuv_amd/csrc/h2y_math.h"

using namespace h2y;

static inline double ref_chain(float L, float *vf)
{
    double Ln = pow((double)L, 0.1593017578);
    double V = pow((0.8359375 + 18.8515625 * Ln) / (1 + 18.6875 * Ln), 78.84375);
    *vf = (float)V;
    return V;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    if (!strcmp(argv[1], "pow")) {
        long n = argc > 2 ? atol(argv[2]) : 10000000;
        std::mt19937_64 rng(1);
        long bad1 = 0, bad2 = 0, big = 0;
        for (long i = 0; i < n; i++) {
            uint32_t b = (uint32_t)(rng() % 0x7F800000u);
            if (b == 0) b = 1;
            float xf = bits2f(b);
            double a = pow((double)xf, H2Y_PQ_M1), m = pow_dd((double)xf, H2Y_PQ_M1);
            if (a != m) {
                bad1++;
                if (fabs((double)(int64_t)(d2bits(a) - d2bits(m))) > 1) big++;
            }
            double base = 0.83 + (rng() >> 11) * 0x1p-53 * 0.18;
            a = pow(base, H2Y_PQ_M2);
            m = pow_dd(base, H2Y_PQ_M2);
            if (a != m) {
                bad2++;
                if (fabs((double)(int64_t)(d2bits(a) - d2bits(m))) > 1) big++;
            }
        }
        printf("pow_dd vs libm: m1 differ %ld / %ld (%.4f%%), m2 differ %ld (%.4f%%), >1ulp: %ld\n", bad1, n,
               100.0 * bad1 / n, bad2, 100.0 * bad2 / n, big);
        return 0;
    }
    if (!strcmp(argv[1], "t1")) {
        /* first tier: every float in [LO,HI): unflagged results must equal the reference float */
        uint32_t lo = (uint32_t)strtoul(argv[2], 0, 0), hi = (uint32_t)strtoul(argv[3], 0, 0);
        int T = argc > 4 ? atoi(argv[4]) : 8;
        std::vector<pq_rec1> T1(H2Y_T1_NREC);
        pq_build_table1(T1.data());
        std::atomic<uint64_t> wrong{0}, flagged{0};
        std::vector<double> worst(T, 0.0);
        std::vector<std::thread> th;
        uint64_t span = (uint64_t)hi - lo;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                uint64_t a = lo + span * t / T, b = lo + span * (t + 1) / T, wr = 0, fl = 0;
                double w = 0;
                for (uint64_t u = a; u < b; u++) {
                    float x = bits2f((uint32_t)u), want;
                    double vref = ref_chain(x, &want);
                    bool uns;
                    float got = pq_t1(x, T1.data(), &uns);
                    if (uns) fl++;
                    else if (f2bits(got) != f2bits(want)) {
                        if (wr++ < 3) fprintf(stderr, "T1 WRONG x=%a got %a want %a\n", x, got, want);
                    }
                    uint32_t off = pq_t1_offset(x) >> 4;
                    if (off >= 1 && off <= H2Y_T1_NSEG && T1[off].cb == T1[off].cb) { /* not a sentinel */
                        /* t as computed against t of the reference's double, in ulps of the output */
                        const float c = bits2f(f2bits(T1[off].cb) - 1u);
                        float vlo; /* the output's binade: that of the segment's first value, as the table's builder takes it */
                        ref_chain(bits2f((uint32_t)u & ~((1u << H2Y_T1_LOW_BITS) - 1u)), &vlo);
                        const double U = ldexp(1.0, (int)(f2bits(vlo) >> 23) - 127 - 23);
                        const double t_ref = (vref - (double)c) / U - 0.5 + (double)H2Y_T1_DELTA;
                        double e = fabs((double)pq_t1_t((uint32_t)u, T1[off]) - t_ref);
                        if (e > w) w = e;
                    }
                }
                wrong += wr; flagged += fl; worst[t] = w;
            });
        for (auto &x : th) x.join();
        double w = 0; for (double x : worst) w = fmax(w, x);
        printf("T1 over [0x%08x,0x%08x): wrong %llu, flagged %llu of %llu (%.3f%%), max |t - t of the reference| = %.5f ulps (H2Y_T1_DELTA %.5f)\n", lo, hi,
               (unsigned long long)wrong.load(), (unsigned long long)flagged.load(), (unsigned long long)span, 100.0 * flagged.load() / span, w, (double)H2Y_T1_DELTA);
        return wrong.load() || w > (double)H2Y_T1_DELTA ? 1 : 0;
    }
    if (!strcmp(argv[1], "tf")) {
        /* the other transfer functions (careful tier only) against libm, random inputs in [0, 1.25) */
        long n = argc > 2 ? atol(argv[2]) : 2000000;
        std::mt19937_64 rng(7);
        long bad[6] = {0, 0, 0, 0, 0, 0};
        for (long i = 0; i < n; i++) {
            float x = (float)(rng() >> 40) * (1.25f / 16777216.0f);
            if (i < 4) x = i == 0 ? 0.0f : i == 1 ? 1.0f : i == 2 ? 0x1p-30f : 1.2f;
            /* reference formulas, convert.cpp:12-87 with C++ overload resolution */
            float r_pqf = (float)pow(fmax(pow((double)x, 1.0 / 78.84375) - 0.8359375, 0.0) / (18.8515625 - 18.6875 * pow((double)x, 1.0 / 78.84375)), 1.0 / 0.1593017578);
            float r_rgf = (float)pow(((double)powf(25.0f, x) - 1.0) / (25.0 - 1.0), (double)2.4f);
            float r_rgr = (float)(log(1.0 + (25.0 - 1.0) * pow((double)x, 1.0 / (double)2.4f)) / (double)logf(25.0f));
            float r_btf = (float)(1.0 * pow(fmax((double)(x + 0.0f), 0.), (double)2.4f));
            float r_btr = (float)(pow(fmax((double)(x / 1.0f), 0.), 1. / (double)2.4f) - 0.0);
            float g[5] = {tf_to_linear(H2Y_TF_PQ, x), tf_to_linear(H2Y_TF_RHO_GAMMA, x), tf_from_linear(H2Y_TF_RHO_GAMMA, x),
                          tf_to_linear(H2Y_TF_BT1886, x), tf_from_linear(H2Y_TF_BT1886, x)};
            float w[5] = {r_pqf, r_rgf, r_rgr, r_btf, r_btr};
            for (int k = 0; k < 5; k++)
                if (f2bits(g[k]) != f2bits(w[k]) && !(g[k] != g[k] && w[k] != w[k])) {
                    if (bad[k]++ < 2) fprintf(stderr, "tf mismatch fn %d x=%a got %a want %a\n", k, x, g[k], w[k]);
                }
        }
        printf("transfer functions vs libm over %ld inputs: PQ_f %ld, RHO_f %ld, RHO_r %ld, BT1886_f %ld, BT1886_r %ld mismatches\n", n, bad[0], bad[1],
               bad[2], bad[3], bad[4]);
        return (bad[0] | bad[1] | bad[2] | bad[3] | bad[4]) ? 1 : 0; /* (RHO_f's inner powf is glibc's algorithm restated: powf25()) */
    }
    if (!strcmp(argv[1], "tfx")) {
        /* The table tier of the other transfer functions (tfn_build_table / tfn_fast, h2y_math.h) against this machine's
         * libm, function by function, over every float x in [LO, HI) (bit patterns; default: the whole table domain
         * [2^-24, 2) plus what lies around it).  A sample the tier flags "slow" is not compared (it goes to the careful
         * tier); one it answers must be the reference's float.  pq_check tfx FN LO HI THREADS [STRIDE] */
        const int fn = atoi(argv[2]);
        uint32_t lo = argc > 3 ? (uint32_t)strtoul(argv[3], 0, 0) : 0x33000000u, hi = argc > 4 ? (uint32_t)strtoul(argv[4], 0, 0) : 0x40800000u;
        int T = argc > 5 ? atoi(argv[5]) : 8;
        uint64_t stride = argc > 6 ? strtoull(argv[6], 0, 0) : 1;
        std::vector<pq_recA> A(2 * H2Y_PQ_NREC);
        const int nbad = tfn_build_table(fn, A.data(), reinterpret_cast<pq_recB *>(A.data() + H2Y_PQ_NREC));
        std::atomic<uint64_t> mism{0}, nslow{0}, total{0};
        std::vector<std::thread> th;
        uint64_t span = (uint64_t)hi - lo;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                uint64_t a = lo + span * t / T, b = lo + span * (t + 1) / T, mm = 0, ns = 0, n = 0;
                for (uint64_t u = a; u < b; u += stride) {
                    float x = bits2f((uint32_t)u);
                    float want;
                    if (fn == H2Y_TFN_RHO_H) { /* the stage's input is (P - 1) / 16 for a float P = powf(25, V) >= 1: walk P, not x */
                        const float P = x;
                        if (!(P >= 1.0f) || !(P < 16777216.0f)) continue;
                        want = (float)pow(((double)P - 1.0) / 24.0, (double)2.4f);
                        x = (P - 1.0f) * 0.0625f;
                    } else
                    switch (fn) {
                    case H2Y_TFN_PQ_R: want = (float)pow((0.8359375 + 18.8515625 * pow((double)x, 0.1593017578)) / (1 + 18.6875 * pow((double)x, 0.1593017578)), 78.84375); break;
                    case H2Y_TFN_PQ_F: { double p = pow((double)x, 1.0 / 78.84375); want = (float)pow(fmax(p - 0.8359375, 0.0) / (18.8515625 - 18.6875 * p), 1.0 / 0.1593017578); break; }
                    case H2Y_TFN_G24: want = (float)(1.0 * pow(fmax((double)(x + 0.0f), 0.), (double)2.4f)); break;
                    case H2Y_TFN_G24INV: want = (float)(pow(fmax((double)(x / 1.0f), 0.), 1. / (double)2.4f) - 0.0); break;
                    case H2Y_TFN_RHO_R: want = (float)(log(1.0 + 24.0 * pow((double)x, 1.0 / (double)2.4f)) / (double)logf(25.0f)); break;
                    default: want = 0.0f; break;
                    }
                    bool slow;
                    const float got = tfn_fast(x, A.data(), tfn_cut_of(fn), tfn_zero_bits(fn), tfn_one_bits(fn), &slow);
                    n++;
                    if (slow) { ns++; continue; }
                    if (f2bits(got) != f2bits(want) && !(got != got && want != want)) {
                        if (mm++ < 3) fprintf(stderr, "tfx fn %d MISMATCH x=%a (0x%08x) got %a want %a\n", fn, x, (uint32_t)u, got, want);
                    }
                }
                mism += mm; nslow += ns; total += n;
            });
        for (auto &x : th) x.join();
        printf("tfx fn %d over [0x%08x,0x%08x) step %llu: %llu floats, mismatches %llu, slow tier %llu (%.4f%%), sentinel segments %d of %d\n", fn, lo, hi,
               (unsigned long long)stride, (unsigned long long)total.load(), (unsigned long long)mism.load(), (unsigned long long)nslow.load(),
               100.0 * nslow.load() / (double)total.load(), nbad, tfn_nseg(tfn_cut_of(fn)));
        return mism ? 1 : 0;
    }
    if (!strcmp(argv[1], "powf")) {
        /* powf25() (glibc's powf algorithm restated, h2y_math.h) against this machine's powf(25.0f, y) for every float
         * y in [LO, HI) (bit patterns); NaN results compare equal.  pq_check powf 0 0x3f800001 8 = all of [0, 1]. */
        uint32_t lo = (uint32_t)strtoul(argv[2], 0, 0), hi = (uint32_t)strtoul(argv[3], 0, 0);
        int T = argc > 4 ? atoi(argv[4]) : 8;
        uint64_t stride = argc > 5 ? strtoull(argv[5], 0, 0) : 1;
        std::atomic<uint64_t> mism{0}, total{0};
        std::vector<std::thread> th;
        uint64_t span = (uint64_t)hi - lo;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                uint64_t a = lo + span * t / T, b = lo + span * (t + 1) / T, mm = 0, n = 0;
                for (uint64_t u = a; u < b; u += stride) {
                    const float y = bits2f((uint32_t)u);
                    volatile float base = 25.0f;
                    const float want = powf(base, y), got = powf25(y);
                    n++;
                    if (f2bits(got) != f2bits(want) && !(got != got && want != want)) {
                        if (mm++ < 3) fprintf(stderr, "powf MISMATCH y=%a (0x%08x) got %a want %a\n", y, (uint32_t)u, got, want);
                    }
                }
                mism += mm;
                total += n;
            });
        for (auto &x : th) x.join();
        printf("powf(25, y) over [0x%08x,0x%08x) step %llu: %llu floats, mismatches %llu\n", lo, hi, (unsigned long long)stride,
               (unsigned long long)total.load(), (unsigned long long)mism.load());
        return mism ? 1 : 0;
    }
    if (!strcmp(argv[1], "ext")) {
        /* pq_check ext <lo> <hi> [threads] [stride]: the table below 2^-24 (pq_build_table_ext, read by pq_slow()): every
         * float of [lo, hi) at the stride -- normal floats below 2^-24 take the table unless ambiguous, everything else
         * the double-double tier -- against the reference chain with this libm */
        uint32_t lo = (uint32_t)strtoul(argv[2], 0, 0), hi = (uint32_t)strtoul(argv[3], 0, 0);
        int T = argc > 4 ? atoi(argv[4]) : 8;
        uint64_t stride = argc > 5 ? strtoull(argv[5], 0, 0) : 1;
        std::vector<pq_ext_rec> X(H2Y_PQX_NSEG);
        pq_build_table_ext(X.data());
        std::atomic<uint64_t> mism{0}, ntab{0}, maxerr{0}, total{0};
        std::vector<std::thread> th;
        uint64_t span = (uint64_t)hi - lo;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                uint64_t a = lo + span * t / T, b = lo + span * (t + 1) / T;
                a += (stride - (a - lo) % stride) % stride;
                uint64_t mm = 0, nt = 0, me = 0, n = 0;
                for (uint64_t u = a; u < b; u += stride, n++) {
                    const float x = bits2f((uint32_t)u);
                    float want, tv;
                    const double vref = ref_chain(x, &want);
                    const float got = pq_slow(x, X.data());
                    if (pq_ext_try(x, X.data(), &tv)) {
                        nt++;
                        const uint32_t idx = ((uint32_t)u >> H2Y_PQ_LOW_BITS) - H2Y_PQX_SEG_BASE;
                        const double v = pq_poly((uint32_t)u, X[idx].a, X[idx].b);
                        const int64_t d = (int64_t)(d2bits(v) - d2bits(vref));
                        const uint64_t ad = d < 0 ? -d : d;
                        if (ad > me) me = ad;
                    }
                    if (f2bits(got) != f2bits(want) && !(got != got && want != want))
                        if (mm++ < 5) fprintf(stderr, "MISMATCH x=%a (0x%08x) got %a want %a\n", x, (uint32_t)u, got, want);
                }
                mism += mm; ntab += nt; total += n;
                uint64_t cur = maxerr.load();
                while (me > cur && !maxerr.compare_exchange_weak(cur, me)) {}
            });
        for (auto &x : th) x.join();
        printf("ext range [0x%08x,0x%08x) step %llu: %llu floats, mismatches %llu, from the table %llu, max table err %llu ulp(double)\n", lo, hi,
               (unsigned long long)stride, (unsigned long long)total.load(), (unsigned long long)mism.load(), (unsigned long long)ntab.load(),
               (unsigned long long)maxerr.load());
        return mism ? 1 : 0;
    }
    bool slow_only = !strcmp(argv[1], "slow");
    uint32_t lo = (uint32_t)strtoul(argv[2], 0, 0), hi = (uint32_t)strtoul(argv[3], 0, 0);
    int T = argc > 4 ? atoi(argv[4]) : 8;
    std::vector<pq_recA> A(H2Y_PQ_NREC);
    std::vector<pq_recB> B(H2Y_PQ_NREC);
    pq_build_table(A.data(), B.data());
    std::atomic<uint64_t> mism{0}, nslow{0}, maxerr{0}, total{0};
    std::vector<std::thread> th;
    uint64_t span = (uint64_t)hi - lo;
    for (int t = 0; t < T; t++) {
        th.emplace_back([&, t]() {
            uint64_t a = lo + span * t / T, b = lo + span * (t + 1) / T;
            uint64_t mm = 0, ns = 0, me = 0;
            for (uint64_t u = a; u < b; u++) {
                float x = bits2f((uint32_t)u);
                float want;
                double vref = ref_chain(x, &want);
                float got;
                if (slow_only) {
                    got = pq_slow(x);
                    ns++;
                } else {
                    bool slow;
                    got = pq_fast(x, A.data(), B.data(), &slow);
                    if (slow) {
                        got = pq_slow(x);
                        ns++;
                    } else {
                        /* fast-tier error in double ulps: recompute v */
                        uint32_t off = pq_rec_offset((uint32_t)u) / 16;
                        double v = pq_poly((uint32_t)u, A[off], B[off]);
                        int64_t d = (int64_t)(d2bits(v) - d2bits(vref));
                        uint64_t ad = d < 0 ? -d : d;
                        if (ad > me) me = ad;
                    }
                }
                if (f2bits(got) != f2bits(want) && !(got != got && want != want)) {
                    mm++;
                    if (mm < 5) fprintf(stderr, "MISMATCH x=%a (0x%08x) got %a want %a\n", x, (uint32_t)u, got, want);
                }
            }
            mism += mm;
            nslow += ns;
            total += b - a;
            uint64_t cur = maxerr.load();
            while (me > cur && !maxerr.compare_exchange_weak(cur, me)) {}
        });
    }
    for (auto &x : th) x.join();
    printf("range [0x%08x,0x%08x): %llu floats, mismatches %llu, slow tier %llu (1 in %.0f), max fast err %llu ulp(double)\n",
           lo, hi, (unsigned long long)total.load(), (unsigned long long)mism.load(),
           (unsigned long long)nslow.load(), nslow ? (double)total / nslow : 0.0, (unsigned long long)maxerr.load());
    return mism ? 1 : 0;
}
