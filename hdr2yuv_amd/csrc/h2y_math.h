/*
 * h2y_math.h -- per-sample arithmetic of the convert path, written once and
 * compiled into the gfx950 kernels (kernels.hip).  Everything here is plain
 * IEEE-754 binary32/binary64 arithmetic with explicit fma() where a fused
 * operation is wanted and -ffp-contract=off everywhere else, so the same
 * source gives the same bits from hipcc's device pass and from a host
 * compiler; tools/pq_exhaustive.cpp and tests/ use that to check every float
 * against the oracle on the CPU (test-only builds; the product library has no
 * host execution path for pixels).
 *
 * What has to be reproduced (reference file:line):
 *   PQ10000_r          convert.cpp:56-63    double pow twice, rounded to float
 *   scale + matrix     convert.cpp:1123-1221
 *   write_yuv clamp    tiff.cpp:457-550
 *
 * PQ strategy.  The reference value is V = (float)pow(g(pow(x,m1)), m2) with
 * both pow()s in double.  Two tiers:
 *   fast  degree-4 polynomial per (exponent, top-6-mantissa-bits) segment of
 *         the float input, coefficients from a table staged in LDS, evaluated
 *         in binary64.  |relative error| < 2^-43 against the exact function.
 *         The result is used only if rounding it to float cannot be affected
 *         by an error of 2^-41 (H2Y_PQ_AMBIG_ULPS); about 1 sample in 2^16
 *         fails that test.
 *   slow  the reference's own sequence of double operations with pow()
 *         replaced by a double-double log/exp pair accurate to ~2^-68 (so its
 *         rounding to double is the correctly rounded result except in ~2^-15
 *         of cases -- the same class of difference as between two libm
 *         versions; SURVEY 8c measured that class to change 0 bytes on 4K
 *         frames).
 */
#ifndef H2Y_MATH_H
#define H2Y_MATH_H

/* Timing-experiment variants of the kernels (some write WRONG BYTES, some dump block clocks and read an environment variable)
 * compile only in a build that declares itself one: `make EXTRA="-DH2Y_EXPERIMENT -DH2Y_EXP_NOCOMPUTE=1"`.  Such a library reports
 * H2Y_ABI_VERSION | H2Y_ABI_EXPERIMENT from h2y_abi_version(): hdr2yuv_amd/api.py, tests/test_abi.py and bench.py refuse it unless told
 * `--allow-experiment` (and then say so in their output). */
#if defined(H2Y_EXP_NOCOMPUTE) || defined(H2Y_HALF_COMPUTE) || defined(H2Y_SKIP_REDO) || defined(H2Y_EXP_NOSTATS) || defined(H2Y_EXP_NOCONFLICT) || \
    defined(H2Y_BLOCK_TIMES)
#ifndef H2Y_EXPERIMENT
#error "timing-experiment variant requested without -DH2Y_EXPERIMENT: this would build a product library with wrong bytes or debug output"
#endif
#endif

#include <stdint.h>
#include <vector>

#if defined(__HIPCC__)
#define H2Y_FN __host__ __device__ __forceinline__
#define H2Y_FN_NOINLINE __host__ __device__ inline __attribute__((noinline))
#else
#define H2Y_FN inline
#define H2Y_FN_NOINLINE inline __attribute__((noinline))
#endif

namespace h2y {

H2Y_FN uint32_t f2bits(float f) { return __builtin_bit_cast(uint32_t, f); }
H2Y_FN float bits2f(uint32_t u) { return __builtin_bit_cast(float, u); }
H2Y_FN uint64_t d2bits(double d) { return __builtin_bit_cast(uint64_t, d); }
H2Y_FN double bits2d(uint64_t u) { return __builtin_bit_cast(double, u); }

/* ------------------------------------------------------------------------
 * double-double arithmetic (error-free transforms; Dekker / Knuth / QD)
 * ---------------------------------------------------------------------- */
struct dd {
    double hi, lo;
};

H2Y_FN dd two_sum(double a, double b)
{
    double s = a + b;
    double bb = s - a;
    double e = (a - (s - bb)) + (b - bb);
    return {s, e};
}
H2Y_FN dd quick_two_sum(double a, double b) /* |a| >= |b| */
{
    double s = a + b;
    double e = b - (s - a);
    return {s, e};
}
H2Y_FN dd two_prod(double a, double b)
{
    double p = a * b;
    double e = __builtin_fma(a, b, -p);
    return {p, e};
}
H2Y_FN dd dd_add(dd a, dd b)
{
    dd s = two_sum(a.hi, b.hi);
    dd t = two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return quick_two_sum(s.hi, s.lo);
}
H2Y_FN dd dd_add_d(dd a, double b)
{
    dd s = two_sum(a.hi, b);
    s.lo += a.lo;
    return quick_two_sum(s.hi, s.lo);
}
H2Y_FN dd dd_mul(dd a, dd b)
{
    dd p = two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return quick_two_sum(p.hi, p.lo);
}
H2Y_FN dd dd_mul_d(dd a, double b)
{
    dd p = two_prod(a.hi, b);
    p.lo = __builtin_fma(a.lo, b, p.lo);
    return quick_two_sum(p.hi, p.lo);
}
/* a / b, both double-double (two Newton-style correction steps) */
H2Y_FN dd dd_div(dd a, dd b)
{
    double q1 = a.hi / b.hi;
    dd r = dd_add(a, dd_mul_d(b, -q1));
    double q2 = r.hi / b.hi;
    r = dd_add(r, dd_mul_d(b, -q2));
    double q3 = r.hi / b.hi;
    dd q = quick_two_sum(q1, q2);
    return dd_add_d(q, q3);
}

/* natural log of a positive, normal double, as double-double (~2^-70 rel.) */
H2Y_FN dd dd_log(double x)
{
    uint64_t b = d2bits(x);
    int k = (int)((b >> 52) & 0x7FF) - 1023;
    double m = bits2d((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull); /* [1,2) */
    if (m >= 1.5) {
        m *= 0.5;
        k += 1;
    } /* m in [0.75,1.5) */
    /* s = (m-1)/(m+1): log m = 2 atanh(s) = 2 s (1 + w/3 + w^2/5 + ...), w = s^2 <= 0.04 */
    double f = m - 1.0; /* exact */
    dd den = two_sum(m, 1.0);
    dd s = dd_div(dd{f, 0.0}, den);
    dd w = dd_mul(s, s);
    double wh = w.hi;
    /* terms w^4/9 ... w^16/33 in double (largest 2.9e-7: 2^-52 of that is < 2^-73) */
    double t = 1.0 / 33.0;
    t = __builtin_fma(t, wh, 1.0 / 31.0);
    t = __builtin_fma(t, wh, 1.0 / 29.0);
    t = __builtin_fma(t, wh, 1.0 / 27.0);
    t = __builtin_fma(t, wh, 1.0 / 25.0);
    t = __builtin_fma(t, wh, 1.0 / 23.0);
    t = __builtin_fma(t, wh, 1.0 / 21.0);
    t = __builtin_fma(t, wh, 1.0 / 19.0);
    t = __builtin_fma(t, wh, 1.0 / 17.0);
    t = __builtin_fma(t, wh, 1.0 / 15.0);
    t = __builtin_fma(t, wh, 1.0 / 13.0);
    t = __builtin_fma(t, wh, 1.0 / 11.0);
    t = __builtin_fma(t, wh, 1.0 / 9.0);
    t = t * ((wh * wh) * (wh * wh));
    /* w/3 + w^2/5 + w^3/7 in double-double */
    const dd c3 = {0x1.5555555555555p-2, 0x1.5555555555555p-56};
    const dd c5 = {0x1.999999999999ap-3, -0x1.999999999999ap-57};
    const dd c7 = {0x1.2492492492492p-3, 0x1.2492492492492p-57};
    dd q = dd_add(dd_mul(w, c7), c5);
    q = dd_add(dd_mul(q, w), c3);
    q = dd_mul(q, w);
    q = dd_add_d(q, t);
    dd lm = dd_add(s, dd_mul(s, q));
    lm.hi *= 2.0;
    lm.lo *= 2.0;
    /* + k ln2 */
    const dd LN2 = {0x1.62e42fefa39efp-1, 0x1.abc9e3b39803fp-56};
    dd kl = dd_mul_d(LN2, (double)k);
    return dd_add(kl, lm);
}

/* exp of a double-double, |z| < 700, as double-double (~2^-70 relative) */
H2Y_FN dd dd_exp_dd(dd z)
{
    const dd LN2 = {0x1.62e42fefa39efp-1, 0x1.abc9e3b39803fp-56};
    double nf = __builtin_rint(z.hi * 0x1.71547652b82fep+0);
    dd r = dd_add(z, dd_mul_d(LN2, -nf)); /* |r| <= 0.35 */
    r.hi *= 0.125;
    r.lo *= 0.125; /* |r| <= 0.044; undone by three squarings */
    double rh = r.hi;
    /* r^4/24 ... r^11/11! in double */
    double t = 1.0 / 39916800.0;
    t = __builtin_fma(t, rh, 1.0 / 3628800.0);
    t = __builtin_fma(t, rh, 1.0 / 362880.0);
    t = __builtin_fma(t, rh, 1.0 / 40320.0);
    t = __builtin_fma(t, rh, 1.0 / 5040.0);
    t = __builtin_fma(t, rh, 1.0 / 720.0);
    t = __builtin_fma(t, rh, 1.0 / 120.0);
    t = __builtin_fma(t, rh, 1.0 / 24.0);
    t = t * ((rh * rh) * (rh * rh));
    /* p = expm1(r) = r (1 + r (1/2 + r/6)) + t, in double-double */
    const dd SIXTH = {0x1.5555555555555p-3, 0x1.5555555555555p-57};
    dd p = dd_add_d(dd_mul(r, SIXTH), 0.5);
    p = dd_add_d(dd_mul(p, r), 1.0);
    p = dd_mul(p, r);
    p = dd_add_d(p, t);
    /* (1+p)^2 - 1 = 2p + p^2, three times */
    for (int i = 0; i < 3; i++) {
        dd p2 = dd_mul(p, p);
        p.hi *= 2.0;
        p.lo *= 2.0;
        p = dd_add(p, p2);
    }
    dd e = dd_add_d(p, 1.0);
    /* scale by 2^n (exact; results on this path are always normal) */
    double sc = bits2d((uint64_t)(1023 + (int)nf) << 52);
    return {e.hi * sc, e.lo * sc};
}
/* ... rounded to double: the high word of a normalised pair is the pair
 * rounded to nearest */
H2Y_FN double dd_exp(dd z) { return dd_exp_dd(z).hi; }

/* x^y for x > 0 finite normal double, y a constant: nearly correctly rounded */
H2Y_FN double pow_dd(double x, double y) { return dd_exp(dd_mul_d(dd_log(x), y)); }

/* PQ constants exactly as the literals of convert.cpp:61 read in binary64 */
#define H2Y_PQ_M1 0.1593017578
#define H2Y_PQ_M2 78.84375
#define H2Y_PQ_C1 0.8359375
#define H2Y_PQ_C2 18.8515625
#define H2Y_PQ_C3 18.6875

/* (float)pow(0.8359375, 78.84375): PQ10000_r(0.0f), since pow(0, m1) == 0 */
#define H2Y_PQ_AT_ZERO_BITS 0x354436e8u

/*
 * Slow tier: PQ10000_r() operation by operation (convert.cpp:61).
 * Out-of-domain inputs follow C's pow(): negative or NaN -> NaN, +inf -> NaN
 * (inf/inf); those are outside the pinned domain anyway.
 */
H2Y_FN bool pq_ext_try(float x, const void *ext, float *v); /* below, with the tables */
H2Y_FN_NOINLINE float pq_slow(float x, const void *ext = nullptr)
{
    if (!(x >= 0.0f) || x > 3.4028234e38f) return bits2f(0x7FC00000u);
    if (x == 0.0f) return bits2f(H2Y_PQ_AT_ZERO_BITS); /* black is common: pow(0, m1) = 0, so this is a constant */
    /* normal floats below 2: the polynomial tier again, from the full-range table in global memory (pq_build_table_ext),
     * before the double-double arithmetic below -- two 16-byte loads instead of ~30 us */
    float ve;
    if (ext && pq_ext_try(x, ext, &ve)) return ve;
    double Ln = pow_dd((double)x, H2Y_PQ_M1);
    double num = H2Y_PQ_C1 + H2Y_PQ_C2 * Ln;
    double den = 1.0 + H2Y_PQ_C3 * Ln;
    double B = num / den;
    double Vd = pow_dd(B, H2Y_PQ_M2);
    return (float)Vd;
}

/* ------------------------------------------------------------------------
 * The other transfer functions of the same dispatch point (convert.cpp:12-87,
 * :1024-1109), careful tier only: the reference's operations one by one with
 * pow()/log() from the double-double routines above.  The reference is C++, so
 * pow(float,float) there is powf and log(float) is logf (see oracle).
 * ---------------------------------------------------------------------- */
#define H2Y_TF_LINEAR 0
#define H2Y_TF_PQ 1
#define H2Y_TF_RHO_GAMMA 2
#define H2Y_TF_BT1886 3 /* BT.709 / BT.601 / BT.2020 10- and 12-bit: gamma 2.4f, Lw 1, Lb 0 */

/* pow(x, y) for any finite x >= 0 and finite y > 0 the way libm defines the edges */
H2Y_FN double pow_gen(double x, double y)
{
    if (!(x >= 0.0)) return bits2d(0x7FF8000000000000ull); /* negative base, non-integer exponent: NaN */
    if (x == 0.0) return 0.0;
    if (x > 1.7976931348623157e308) return x; /* +inf */
    return pow_dd(x, y);
}
/* ------------------------------------------------------------------------
 * powf(25.0f, y) as glibc computes it.
 *
 * RHO_GAMMA_f (convert.cpp:23) calls pow(float rho, float V): in C++ that is powf, and glibc's powf is not a
 * correctly rounded function -- its float depends on its algorithm (0.06 % of results differ from the correctly
 * rounded value).  So that algorithm is restated here: glibc 2.28+ sysdeps/ieee754/flt-32/e_powf.c (Szabolcs
 * Nagy's ARM optimized routines), in the form the x86-64 FMA build runs (libm's ifunc picks __powf_fma wherever the
 * CPU has FMA; every multiply-add of the source is one fused operation there, the product y * log2(x) is not --
 * read off the machine code of glibc 2.35).  x is always 25.0f here, so log2(x) is a constant of the algorithm: its
 * table entry and the degree-4 polynomial, evaluated by powf_log2_of_25() the way the function does.  tools/pq_check powf compares with this machine's powf over every float of [0, 1] and
 * sampled wider ranges (tests/test_pq_math.py).  A host without FMA runs glibc's unfused build and may round a few
 * results the other way; the reference's bytes then differ too (parity follows the FMA build, which is what both
 * this container and the GPU boxes' hosts run).
 * ---------------------------------------------------------------------- */
H2Y_FN double powf_log2_of_25(void)
{
    /* log2_inline(asuint(25.0f)): OFF = 0x3f330000, 16 subintervals; tmp = ix - OFF, i = (tmp >> 19) % 16,
     * top = tmp & 0xff800000, z = asfloat(ix - top), k = top >> 23 */
    const uint32_t ix = 0x41C80000u, tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const uint32_t top = tmp & 0xff800000u;
    const double z = (double)bits2f(ix - top);
    const double k = (double)((int32_t)top >> 23);
    /* __powf_log2_data.tab[i] = {invc, logc} */
    const double invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0, 0x1.3c995b0b80385p+0, 0x1.30d190c8864a5p+0,
                             0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0, 0x1.0953f419900a7p+0, 0x1p+0,
                             0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,  0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1,
                             0x1.767dcf5534862p-1};
    const double logc[16] = {-0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2, -0x1.01d9bf3f2b631p-2,
                             -0x1.97c1d1b3b7afp-3,  -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4, -0x1.a6f9db6475fcep-5, 0x0p+0,
                             0x1.338ca9f24f53dp-4,  0x1.476a9543891bap-3,  0x1.e840b4ac4e4d2p-3,  0x1.40645f0c6651cp-2,  0x1.88e9c2c1b9ff8p-2,
                             0x1.ce0a44eb17bccp-2};
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1,
                 A4 = 0x1.71547652ab82bp+0;
    const double r = __builtin_fma(z, invc[i], -1.0);
    const double y0 = logc[i] + k;
    const double r2 = r * r;
    double y = __builtin_fma(A0, r, A1);
    const double p = __builtin_fma(A2, r, A3);
    const double r4 = r2 * r2;
    double q = __builtin_fma(A4, r, y0);
    q = __builtin_fma(p, r2, q);
    y = __builtin_fma(y, r4, q);
    return y;
}
/* 2^(i/32) as asuint64(2^(i/32)) - (i << 47): __exp2f_data.tab */
H2Y_FN uint64_t exp2f_tab(uint32_t i)
{
    const uint64_t T[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
        0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
        0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
        0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    return T[i & 31u];
}
H2Y_FN float powf25(float yf)
{
    const uint32_t iy = f2bits(yf);
    if (2u * iy - 1u >= 2u * 0x7f800000u - 1u) { /* zeroinfnan(iy): y is 0, inf or NaN */
        if (2u * iy == 0u) return 1.0f;
        if (2u * iy > 2u * 0x7f800000u) return yf + yf; /* NaN */
        return (iy & 0x80000000u) ? 0.0f : yf * yf;     /* |x| > 1: y = -inf gives 0, +inf gives inf */
    }
    const double logx = powf_log2_of_25();
    const double ylogx = (double)yf * logx; /* its own rounding: not fused with what follows */
    if (((d2bits(ylogx) >> 47) & 0xffffu) >= (d2bits(126.0) >> 47)) { /* |y log2 x| >= 126 */
        if (ylogx > 0x1.fffffffd1d571p+6) return bits2f(0x7F800000u); /* overflow */
        if (ylogx <= -150.0) return 0.0f;                              /* underflow */
        if (ylogx < -149.0) return bits2f(1u);                         /* __math_may_uflowf: 0x1.4p-75f squared rounds to the smallest subnormal */
    }
    /* exp2_inline: x = k/32 + r, |r| <= 1/64 */
    const double SHIFT = 0x1.8p+47; /* __exp2f_data.shift_scaled */
    double kd = ylogx + SHIFT;
    const uint64_t ki = d2bits(kd);
    kd -= SHIFT;
    const double r = ylogx - kd;
    uint64_t t = exp2f_tab((uint32_t)ki);
    t += ki << 47; /* 52 - EXP2F_TABLE_BITS */
    const double s = bits2d(t);
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    const double z = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    return (float)y;
}

H2Y_FN float tf_to_linear(int cls, float V)
{
    if (cls == H2Y_TF_PQ) { /* PQ10000_f, convert.cpp:43-51 */
        double p = pow_gen((double)V, 1.0 / 78.84375);
        double num = p - 0.8359375;
        num = num > 0.0 ? num : 0.0; /* fmax(., 0.0): a NaN also gives 0.0 */
        return (float)pow_gen(num / (18.8515625 - 18.6875 * p), 1.0 / 0.1593017578);
    }
    if (cls == H2Y_TF_RHO_GAMMA) { /* RHO_GAMMA_f, convert.cpp:12-27: the inner pow is powf (glibc's, restated above), the outer double */
        const float P = powf25(V);
        return (float)pow_gen(((double)P - 1.0) / 24.0, (double)2.4f);
    }
    if (cls == H2Y_TF_BT1886) { /* bt1886_f, convert.cpp:67-75, a = 1, b = 0 */
        double v = (double)(V + 0.0f);
        v = v > 0.0 ? v : 0.0;
        return (float)pow_gen(v, (double)2.4f);
    }
    return V;
}
H2Y_FN float tf_from_linear(int cls, float L, const void *pq_ext = nullptr)
{
    if (cls == H2Y_TF_PQ) return pq_slow(L, pq_ext); /* PQ10000_r */
    if (cls == H2Y_TF_RHO_GAMMA) {           /* RHO_GAMMA_r, convert.cpp:30-38: log(rho) is logf */
        double a = 1.0 + 24.0 * pow_gen((double)L, 1.0 / (double)2.4f);
        double la = (a > 0.0 && a < 1.7976931348623157e308) ? dd_log(a).hi : bits2d(0x7FF8000000000000ull);
        return (float)(la / (double)bits2f(0x404E0210u)); /* logf(25.0f) = 3.21887589 */
    }
    if (cls == H2Y_TF_BT1886) { /* bt1886_r, convert.cpp:79-87 */
        double v = (double)(L / 1.0f);
        v = v > 0.0 ? v : 0.0;
        return (float)(pow_gen(v, 1. / (double)2.4f) - 0.0);
    }
    return L;
}

/* same function in double-double throughout: the "exact" PQ the table is
 * fitted to (x > 0 given as a double-double) */
H2Y_FN dd pq_exact_dd(dd x)
{
    dd lx = dd_add_d(dd_log(x.hi), x.lo / x.hi); /* log(hi+lo) = log hi + log1p(lo/hi) */
    dd t = dd_exp_dd(dd_mul_d(lx, H2Y_PQ_M1));
    dd num = dd_add_d(dd_mul_d(t, H2Y_PQ_C2), H2Y_PQ_C1);
    dd den = dd_add_d(dd_mul_d(t, H2Y_PQ_C3), 1.0);
    dd B = dd_div(num, den);
    dd lb = dd_add_d(dd_log(B.hi), B.lo / B.hi);
    return dd_exp_dd(dd_mul_d(lb, H2Y_PQ_M2));
}

/* ------------------------------------------------------------------------
 * Fast tier table.
 * Segment index = (float bits >> 17) - H2Y_PQ_SEG_BASE: exponent and top 6
 * mantissa bits, for x in [2^H2Y_PQ_EMIN, 2).  Two 16-byte records per
 * segment, kept in two arrays so that neighbouring segments fall in
 * neighbouring LDS bank groups (one ds_read_b128 each):
 *     A[i] = { c0, c1 }           (2 x binary64)
 *     B[i] = { c2, c3|c4 }        (binary64, 2 x binary32)
 * value = c0 + u(c1 + u(c2 + u(c3 + u c4))), u = offset of the mantissa from
 * the segment centre, in [-2^-7, 2^-7) (units of the binade's leading bit).
 * Entry H2Y_PQ_NSEG is a sentinel every out-of-table input is steered to: its
 * value sits exactly on a float rounding tie, so the ambiguity test below
 * sends the sample to the slow tier without a separate range check.
 * ---------------------------------------------------------------------- */
#define H2Y_PQ_EMIN (-24)
#define H2Y_PQ_SEG_BITS 6
#define H2Y_PQ_SEG_PER_BINADE (1 << H2Y_PQ_SEG_BITS)
#define H2Y_PQ_NBINADES (1 - H2Y_PQ_EMIN) /* exponents EMIN..0 */
#define H2Y_PQ_NSEG (H2Y_PQ_NBINADES * H2Y_PQ_SEG_PER_BINADE)
#define H2Y_PQ_NREC (H2Y_PQ_NSEG + 1)
#define H2Y_PQ_LOW_BITS (23 - H2Y_PQ_SEG_BITS)
#define H2Y_PQ_SEG_BASE ((uint32_t)(127 + H2Y_PQ_EMIN) << H2Y_PQ_SEG_BITS)
#define H2Y_PQ_TABLE_BYTES (H2Y_PQ_NREC * 32)

/* A double's low 29 mantissa bits decide its rounding to float; the tie is
 * at 2^28.  The fast value is trusted when those bits are at least this far
 * (in units of the double's last place) from the tie: 2^12 ulp = 2^-40
 * relative at worst (mantissa in [1,2)), ~6x the measured fast-tier error
 * (tools/pq_check: max 689 ulp over every float in the table's domain). */
#define H2Y_PQ_AMBIG_ULPS 4096u

struct alignas(16) pq_recA {
    double c0, c1;
};
struct alignas(16) pq_recB {
    double c2;
    float c3, c4;
};

/* the polynomial itself (shared with tools/pq_check.cpp) */
H2Y_FN double pq_poly(uint32_t bits, const pq_recA &a, const pq_recB &b)
{
    /* 1.0 + low mantissa bits, minus the segment centre 1 + 2^-7: exact in binary32 */
    float f = bits2f((bits & ((1u << H2Y_PQ_LOW_BITS) - 1u)) | 0x3F800000u);
    float u = f - (1.0f + 1.0f / (float)(2 << H2Y_PQ_SEG_BITS));
    float p = __builtin_fmaf(b.c4, u, b.c3);
    double ud = (double)u;
    double v = __builtin_fma((double)p, ud, b.c2);
    v = __builtin_fma(v, ud, a.c1);
    return __builtin_fma(v, ud, a.c0);
}
H2Y_FN uint32_t umin3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    const uint32_t m = a < b ? a : b;
    return m < c ? m : c;
#endif
}
H2Y_FN uint32_t umax3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    const uint32_t m = a > b ? a : b;
    return m > c ? m : c;
#endif
}
H2Y_FN uint32_t umin32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r; /* keep it one v_min_u32: the optimiser otherwise rewrites it as compare + select */
    asm("v_min_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a < b ? a : b;
#endif
}
/* byte offset of the segment's records; out-of-table inputs (0, tiny, >= 2,
 * negative, NaN) all wrap above the table and are clamped to the sentinel */
H2Y_FN uint32_t pq_rec_offset(uint32_t bits)
{
    uint32_t t = bits - (H2Y_PQ_SEG_BASE << H2Y_PQ_LOW_BITS);
    uint32_t idx = umin32(t >> H2Y_PQ_LOW_BITS, (uint32_t)H2Y_PQ_NSEG);
    return idx << 4;
}
/* low 29 bits within AMBIG of the tie at 2^28: shifting left by 3 drops the other bits */
H2Y_FN bool pq_ambiguous(double v)
{
    uint32_t lo = (uint32_t)d2bits(v);
    return ((lo << 3) + ((H2Y_PQ_AMBIG_ULPS - 0x10000000u) << 3)) < ((2u * H2Y_PQ_AMBIG_ULPS) << 3);
}

/* Fast tier, split so that a caller can issue the table loads of several
 * samples before it consumes any of them (LDS latency is the bottleneck of
 * this tier): pq_fetch() does the two 16-byte loads, pq_eval() the arithmetic. */
#if defined(__clang__)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#else
struct alignas(16) u32x4 {
    uint32_t x, y, z, w;
};
#endif
struct pq_rec {
    u32x4 ra, rb; /* raw pq_recA / pq_recB */
};
H2Y_FN pq_rec pq_fetch(float x, const pq_recA *__restrict__ A, const pq_recB *__restrict__ B)
{
    const uint32_t off = pq_rec_offset(f2bits(x));
    /* (B is A + H2Y_PQ_NREC records in the kernels' LDS image: one address
     * register, the second load uses the instruction's immediate offset) */
    pq_rec r;
    r.ra = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(A) + off);
    r.rb = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(B) + off);
    return r;
}
H2Y_FN float pq_eval(float x, const pq_rec &r, bool *slow)
{
    pq_recA a;
    pq_recB b;
    a.c0 = bits2d((uint64_t)r.ra.x | ((uint64_t)r.ra.y << 32));
    a.c1 = bits2d((uint64_t)r.ra.z | ((uint64_t)r.ra.w << 32));
    b.c2 = bits2d((uint64_t)r.rb.x | ((uint64_t)r.rb.y << 32));
    b.c3 = bits2f(r.rb.z);
    b.c4 = bits2f(r.rb.w);
    const double v = pq_poly(f2bits(x), a, b);
    *slow = pq_ambiguous(v);
    return (float)v;
}
/* Returns the float value; *slow is set when the caller must use pq_slow(x)
 * instead (x outside the table, or rounding too close to call). */
H2Y_FN float pq_fast(float x, const pq_recA *__restrict__ A, const pq_recB *__restrict__ B, bool *slow)
{
    bool sl;
    const float v = pq_eval(x, pq_fetch(x, A, B), &sl);
    /* +0.0 (black bars, unused channels) is outside the table but common: its value is a constant */
    const bool zero = f2bits(x) == 0u;
    *slow = sl & !zero;
    return zero ? bits2f(H2Y_PQ_AT_ZERO_BITS) : v;
}

/* Host-side table builder (context creation).  Per segment: interpolate the
 * double-double PQ at five near-Chebyshev nodes (Newton divided differences
 * in double-double), expand to monomials, round c0..c2 to binary64 and c3,c4
 * to binary32.  Pure IEEE double arithmetic: the same table on every host, no
 * libm involved.  A and B each hold H2Y_PQ_NREC records. */
inline void pq_build_segments(pq_recA *A, pq_recB *B, int emin, int nseg)
{
    const double un[5] = {-0.9510565162951535, -0.5877852522924731, 0.0, 0.5877852522924731, 0.9510565162951535};
    for (int i = 0; i < nseg; i++) {
        int e = emin + i / H2Y_PQ_SEG_PER_BINADE;
        int s = i % H2Y_PQ_SEG_PER_BINADE;
        double scale = bits2d((uint64_t)(1023 + e) << 52);
        double mid = scale * (1.0 + (s + 0.5) / H2Y_PQ_SEG_PER_BINADE); /* exact */
        double half = scale * (0.5 / H2Y_PQ_SEG_PER_BINADE);            /* exact */
        dd dv[5];
        for (int j = 0; j < 5; j++) dv[j] = pq_exact_dd(dd_add_d(two_prod(un[j], half), mid));
        /* divided differences in the normalised variable w in [-1,1] */
        for (int lvl = 1; lvl < 5; lvl++)
            for (int j = 4; j >= lvl; j--) {
                dd num = dd_add(dv[j], dd{-dv[j - 1].hi, -dv[j - 1].lo});
                dv[j] = dd_div(num, dd{un[j] - un[j - lvl], 0.0});
            }
        /* p(w) = dv0 + (w-w0)(dv1 + (w-w1)(dv2 + (w-w2)(dv3 + (w-w3) dv4))) -> monomials */
        dd c[5] = {dv[4], {0, 0}, {0, 0}, {0, 0}, {0, 0}};
        int deg = 0;
        for (int j = 3; j >= 0; j--) {
            dd nc[5] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
            for (int k = 0; k <= deg; k++) {
                nc[k + 1] = dd_add(nc[k + 1], c[k]);
                nc[k] = dd_add(nc[k], dd_mul_d(c[k], -un[j]));
            }
            nc[0] = dd_add(nc[0], dv[j]);
            deg++;
            for (int k = 0; k <= deg; k++) c[k] = nc[k];
        }
        /* the kernel's u is w / 2^(SEG_BITS+1): rescale by exact powers of two */
        const double r = (double)(2 << H2Y_PQ_SEG_BITS);
        A[i].c0 = c[0].hi;
        A[i].c1 = c[1].hi * r;
        B[i].c2 = c[2].hi * r * r;
        B[i].c3 = (float)(c[3].hi * r * r * r);
        B[i].c4 = (float)(c[4].hi * r * r * r * r);
    }
}
inline void pq_build_table(pq_recA *A, pq_recB *B)
{
    pq_build_segments(A, B, H2Y_PQ_EMIN, H2Y_PQ_NSEG);
    /* sentinel: 1 + 2^-24 is exactly half way between two floats */
    A[H2Y_PQ_NSEG].c0 = 1.0 + 0x1p-24;
    A[H2Y_PQ_NSEG].c1 = 0.0;
    B[H2Y_PQ_NSEG].c2 = 0.0;
    B[H2Y_PQ_NSEG].c3 = 0.0f;
    B[H2Y_PQ_NSEG].c4 = 0.0f;
}

/* ------------------------------------------------------------------------
 * The same table over every NORMAL float below 2 (exponents -126 .. 0; 8128 segments, 254 KB), in global memory, read
 * only from pq_slow(): by the samples the LDS tables do not cover -- pictures that did not pass through half floats
 * can hold positive samples below 2^-24 -- and by the other samples of a pixel that pixel_careful() redoes as a whole.
 * The double-double arithmetic costs ~30 us of a wave's time per sample (a 4K frame with 0.02 % of its samples below
 * 2^-24 ran 24 % longer, the fused FIR path twice as long).  Same polynomial, same ambiguity test; subnormal inputs
 * and ambiguous values still go to the double-double tier.  Checked against libm like the LDS table
 * (tools/pq_check ext: every float of [2^-126, 2)).
 * ---------------------------------------------------------------------- */
#define H2Y_PQX_EMIN (-126)
#define H2Y_PQX_NSEG ((1 - H2Y_PQX_EMIN) * H2Y_PQ_SEG_PER_BINADE)
#define H2Y_PQX_SEG_BASE ((uint32_t)(127 + H2Y_PQX_EMIN) << H2Y_PQ_SEG_BITS)
#define H2Y_PQX_TABLE_BYTES (H2Y_PQX_NSEG * 32) /* one 32-byte record per segment: pq_recA, then pq_recB (pq_ext_rec) */
struct alignas(32) pq_ext_rec { /* interleaved, so that one 32-byte scalar load fetches a segment (pq_ext_inline, h2y_device.h) */
    pq_recA a;
    pq_recB b;
};
inline void pq_build_table_ext(pq_ext_rec *X)
{
    pq_recA *A = new pq_recA[H2Y_PQX_NSEG];
    pq_recB *B = new pq_recB[H2Y_PQX_NSEG];
    pq_build_segments(A, B, H2Y_PQX_EMIN, H2Y_PQX_NSEG);
    for (int i = 0; i < H2Y_PQX_NSEG; i++) {
        X[i].a = A[i];
        X[i].b = B[i];
    }
    delete[] A;
    delete[] B;
}
H2Y_FN bool pq_ext_try(float x, const void *ext, float *v)
{
    const uint32_t bits = f2bits(x), idx = (bits >> H2Y_PQ_LOW_BITS) - H2Y_PQX_SEG_BASE; /* subnormal: wraps high; >= 2, negative, NaN: high */
    if (idx >= (uint32_t)H2Y_PQX_NSEG) return false;
    const pq_ext_rec *X = reinterpret_cast<const pq_ext_rec *>(ext);
    const double d = pq_poly(bits, X[idx].a, X[idx].b);
    *v = (float)d;
    return !pq_ambiguous(d);
}

/* ------------------------------------------------------------------------
 * The same tier for the OTHER transfer functions of the dispatch point (convert.cpp:12-87, :1024-1109):
 * every one of them is (float)F(double(x)) for a smooth F, so the table machinery of PQ10000_r carries over -- a
 * degree-4 polynomial per (exponent, top-6-mantissa-bits) segment of x in [2^-24, 2), binary64 evaluation, result
 * trusted only when its rounding to float is not within H2Y_PQ_AMBIG_ULPS of a tie.  What differs per function:
 *   - the reference function in double-double (tfn_exact_dd) the table is fitted to;
 *   - segments the polynomial cannot be trusted on are found by MEASURING it against that reference at 33 points
 *     per segment at build time; they, and their neighbours, get the sentinel record (value on a rounding tie => the
 *     sample goes to the careful tier): the kink of PQ10000_f where its numerator crosses zero, the foot of
 *     RHO_GAMMA_f at P = 1, anything that is NaN or would round to a subnormal float;
 *   - the value at +0.0 (outside the table): 0 for all of them.
 * A transfer pair is two stages with a float in between, as in the reference: source function -> linear light
 * (float) -> destination function.  RHO_GAMMA_f = H(powf25(V)) with H(P) = ((P - 1) / 24)^2.4: powf25() is exact by
 * construction, H is tabulated over P / 16 (P in [1, 25] -> [1/16, 25/16), inside the table's domain).
 * tools/pq_check tfx compares every function's fast tier with libm over every float of the table's domain.
 * ---------------------------------------------------------------------- */
enum : int { H2Y_TFN_NONE = 0, H2Y_TFN_PQ_R = 1, H2Y_TFN_PQ_F = 2, H2Y_TFN_G24 = 3, H2Y_TFN_G24INV = 4, H2Y_TFN_RHO_R = 5, H2Y_TFN_RHO_H = 6, H2Y_TFN_COUNT = 7 };
#define H2Y_GAMMA24 ((double)2.4f)       /* "const float gamma = 2.4" promoted (convert.cpp:15, :1052) */
#define H2Y_LOGF25 ((double)bits2f(0x404E0210u)) /* logf(25.0f), convert.cpp:35 */

/* How a function's table cuts the floats into segments: 2^seg_bits segments per binade over the binades emin .. 0,
 * and -- for a function that needs it -- 2^hi_seg_bits per binade from binade hi_emin up (hi_emin = 1: no such part).
 * At most H2Y_PQ_NSEG segments in all; the records after the last are sentinels, H2Y_PQ_NSEG the catch-all one.
 * Most functions take PQ10000_r's cut (64 per binade from 2^-24).  PQ10000_f is steep where it matters -- locally V^10
 * near V = 1, with a pole at V = 1.99 -- and needs 256 segments per binade from 2^-3 up for the degree-4 fit to reach
 * 2^-43; below that 64 do, down to 2^-12 (PQ code values below 0.00025: the careful tier; exact zero has its own answer). */
struct tfn_cut {
    int seg_bits, emin, hi_seg_bits, hi_emin;
};
H2Y_FN tfn_cut tfn_cut_of(int fn)
{
    return fn == H2Y_TFN_PQ_F ? tfn_cut{6, -12, 8, -3} : tfn_cut{H2Y_PQ_SEG_BITS, H2Y_PQ_EMIN, H2Y_PQ_SEG_BITS, 1};
}
H2Y_FN int tfn_nseg_lo(tfn_cut c) { return ((c.hi_emin < 1 ? c.hi_emin : 1) - c.emin) << c.seg_bits; }
H2Y_FN int tfn_nseg(tfn_cut c) { return tfn_nseg_lo(c) + (c.hi_emin < 1 ? (1 - c.hi_emin) << c.hi_seg_bits : 0); }

H2Y_FN dd dd_log_dd(dd x) { return dd_add_d(dd_log(x.hi), x.lo / x.hi); } /* log(hi + lo) = log hi + log1p(lo / hi) */
H2Y_FN dd dd_pow_d(dd x, double e) { return dd_exp_dd(dd_mul_d(dd_log_dd(x), e)); }
H2Y_FN dd dd_nan(void) { return {bits2d(0x7FF8000000000000ull), 0.0}; }
/* F(x) for x > 0 (a double-double); NaN where the reference's function is NaN or the table must not be used */
H2Y_FN dd tfn_exact_dd(int fn, dd x)
{
    switch (fn) {
    case H2Y_TFN_PQ_R: return pq_exact_dd(x);
    case H2Y_TFN_PQ_F: { /* convert.cpp:49 */
        const dd p = dd_pow_d(x, 1.0 / 78.84375);
        const dd num = dd_add_d(p, -0.8359375);
        if (!(num.hi > 0.0)) return {0.0, 0.0}; /* fmax(., 0.0), then pow(0, 1/m1) = 0 */
        /* Above V = 1 the segments are twice as wide and the function heads for its pole at V = 1.99, where the
         * reference's own double arithmetic (c2 - c3 p, with p rounded) loses digits too: no table there.  V = 1.0
         * itself -- peak white -- has its own answer (tfn_one_bits). */
        if (x.hi >= 1.0) return dd_nan();
        const dd den = dd_add_d(dd_mul_d(p, -18.6875), 18.8515625);
        return dd_pow_d(dd_div(num, den), 1.0 / 0.1593017578);
    }
    case H2Y_TFN_G24: return dd_pow_d(x, H2Y_GAMMA24);          /* bt1886_f with a = 1, b = 0, convert.cpp:73 */
    case H2Y_TFN_G24INV: return dd_pow_d(x, 1. / H2Y_GAMMA24);  /* bt1886_r, convert.cpp:85 */
    case H2Y_TFN_RHO_R: { /* convert.cpp:35 */
        const dd t = dd_pow_d(x, 1.0 / H2Y_GAMMA24);
        const dd a = dd_add_d(dd_mul_d(t, 24.0), 1.0);
        return dd_div(dd_log_dd(a), dd{H2Y_LOGF25, 0.0});
    }
    case H2Y_TFN_RHO_H: /* convert.cpp:23 after the inner powf: x = (P - 1) / 16 (exact in binary32 for P in [1, 2^24)),
                           H = ((P - 1) / 24)^2.4 = (x / 1.5)^2.4 -- a pure power of x: every binade looks the same */
        return dd_pow_d(dd_div(x, dd{1.5, 0.0}), H2Y_GAMMA24);
    default: return dd_nan();
    }
}
/* float value of the function at +0.0 (bits) */
H2Y_FN uint32_t tfn_zero_bits(int fn) { return fn == H2Y_TFN_PQ_R ? H2Y_PQ_AT_ZERO_BITS : 0u; }
/* float value at 1.0f where that input has an answer of its own (PQ10000_f(1) = 1: p = 1, (1 - c1) / (c2 - c3) = 1), else 0 = none */
H2Y_FN uint32_t tfn_one_bits(int fn) { return fn == H2Y_TFN_PQ_F ? 0x3F800000u : 0u; }

/* the polynomial of one segment: pq_poly() with the segment width as a parameter */
H2Y_FN double tfn_poly(uint32_t bits, int seg_bits, const pq_recA &a, const pq_recB &b)
{
    const int low_bits = 23 - seg_bits;
    const float f = bits2f((bits & ((1u << low_bits) - 1u)) | 0x3F800000u);
    const float u = f - bits2f(0x3F800000u + (1u << (low_bits - 1))); /* minus the segment centre 1 + 2^-(seg_bits + 1): exact */
    const float p = __builtin_fmaf(b.c4, u, b.c3);
    const double ud = (double)u;
    double v = __builtin_fma((double)p, ud, b.c2);
    v = __builtin_fma(v, ud, a.c1);
    return __builtin_fma(v, ud, a.c0);
}
/* record index and the segment width there; everything outside the table (0, tiny, >= 2, negative, NaN) wraps above it
 * and lands on a sentinel */
H2Y_FN uint32_t tfn_index(uint32_t bits, tfn_cut c, int *seg_bits)
{
    const uint32_t t_lo = bits - ((uint32_t)(127 + c.emin) << 23), t_hi = bits - ((uint32_t)(127 + c.hi_emin) << 23);
    const bool hi = c.hi_emin < 1 && (int32_t)t_hi >= 0 && (int32_t)bits >= 0;
    *seg_bits = hi ? c.hi_seg_bits : c.seg_bits;
    const uint32_t i_lo = t_lo >> (23 - c.seg_bits), i_hi = (uint32_t)tfn_nseg_lo(c) + (t_hi >> (23 - c.hi_seg_bits));
    return umin32(hi ? i_hi : i_lo, (uint32_t)H2Y_PQ_NSEG);
}
/* the first float of segment i, and its width */
H2Y_FN uint32_t tfn_segment_bits(int i, tfn_cut c, int *seg_bits)
{
    const int n_lo = tfn_nseg_lo(c);
    if (i < n_lo) {
        *seg_bits = c.seg_bits;
        return ((uint32_t)(127 + c.emin) << 23) + ((uint32_t)i << (23 - c.seg_bits));
    }
    *seg_bits = c.hi_seg_bits;
    return ((uint32_t)(127 + c.hi_emin) << 23) + ((uint32_t)(i - n_lo) << (23 - c.hi_seg_bits));
}

/* Generic form of pq_build_table() with the per-segment check described above.  Returns how many segments of the
 * table's domain ended up on the sentinel.  Records beyond the cut's last segment are sentinels too. */
inline int tfn_build_records(int fn, tfn_cut cut, int last_rec /* records 0 .. last_rec are written; beyond the cut: sentinels */, pq_recA *A, pq_recB *B)
{
    const double un[5] = {-0.9510565162951535, -0.5877852522924731, 0.0, 0.5877852522924731, 0.9510565162951535};
    const int nseg = tfn_nseg(cut);
    std::vector<unsigned char> bad(nseg, 0); /* 1: not accurate enough or not defined here; 2: a kink (takes its neighbours along) */
    for (int i = 0; i < nseg; i++) {
        int sbits;
        const uint32_t seg_bits0 = tfn_segment_bits(i, cut, &sbits);
        const int low_bits = 23 - sbits;
        const double x0 = (double)bits2f(seg_bits0), x1 = (double)bits2f(seg_bits0 + (1u << low_bits)); /* the segment's ends */
        const double mid = 0.5 * (x0 + x1), half = 0.5 * (x1 - x0);
        dd dv[5];
        bool ok = true, kink = false;
        int zeros = 0;
        for (int j = 0; j < 5; j++) {
            dv[j] = tfn_exact_dd(fn, dd_add_d(two_prod(un[j], half), mid));
            ok = ok && dv[j].hi == dv[j].hi;
            zeros += dv[j].hi == 0.0;
        }
        kink = zeros != 0 && zeros != 5; /* the function leaves zero inside this segment */
        if (ok) {
            for (int lvl = 1; lvl < 5; lvl++)
                for (int j = 4; j >= lvl; j--) dv[j] = dd_div(dd_add(dv[j], dd{-dv[j - 1].hi, -dv[j - 1].lo}), dd{un[j] - un[j - lvl], 0.0});
            dd c[5] = {dv[4], {0, 0}, {0, 0}, {0, 0}, {0, 0}};
            int deg = 0;
            for (int j = 3; j >= 0; j--) {
                dd nc[5] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
                for (int k = 0; k <= deg; k++) {
                    nc[k + 1] = dd_add(nc[k + 1], c[k]);
                    nc[k] = dd_add(nc[k], dd_mul_d(c[k], -un[j]));
                }
                nc[0] = dd_add(nc[0], dv[j]);
                deg++;
                for (int k = 0; k <= deg; k++) c[k] = nc[k];
            }
            const double r = (double)(2 << sbits);
            A[i].c0 = c[0].hi;
            A[i].c1 = c[1].hi * r;
            B[i].c2 = c[2].hi * r * r;
            B[i].c3 = (float)(c[3].hi * r * r * r);
            B[i].c4 = (float)(c[4].hi * r * r * r * r);
            /* measure: the polynomial as the kernels evaluate it against the reference, 33 floats across the segment */
            for (int k = 0; k <= 32 && ok; k++) {
                const uint32_t low = k == 32 ? (1u << low_bits) - 1u : (uint32_t)k << (low_bits - 5);
                const uint32_t bits = seg_bits0 | low;
                const double v = tfn_poly(bits, sbits, A[i], B[i]);
                const dd ref = tfn_exact_dd(fn, dd{(double)bits2f(bits), 0.0});
                if (!(ref.hi == ref.hi) || !(v == v)) { ok = false; break; }
                const double err = (v - ref.hi) - ref.lo, mag = ref.hi < 0 ? -ref.hi : ref.hi;
                const double aerr = err < 0 ? -err : err;
                if (mag == 0.0) ok = aerr == 0.0;
                else {
                    if (mag < 0x1p-120) kink = true; /* next to where it leaves zero: relative accuracy means nothing yet */
                    ok = mag >= 0x1p-120 && aerr <= mag * 0x1p-43; /* a quarter of the ambiguity margin (2^-41 worst case) */
                }
            }
        }
        bad[i] = kink ? 2 : (ok ? 0 : 1);
    }
    int nbad = 0;
    for (int i = 0; i <= last_rec; i++) {
        bool b = i >= nseg || bad[i]; /* a kink takes its two neighbours on either side with it */
        for (int k = -2; k <= 2 && !b; k++) b = i + k >= 0 && i + k < nseg && bad[i + k] == 2;
        if (b) {
            A[i].c0 = 1.0 + 0x1p-24; /* exactly half way between two floats: always "ambiguous" */
            A[i].c1 = 0.0;
            B[i].c2 = 0.0;
            B[i].c3 = B[i].c4 = 0.0f;
            if (i < nseg) nbad++;
        }
    }
    return nbad;
}
inline int tfn_build_table(int fn, pq_recA *A, pq_recB *B) { return tfn_build_records(fn, tfn_cut_of(fn), H2Y_PQ_NSEG, A, B); }
/* The same function over EVERY normal float below 2 (64 segments per binade from 2^-126: the layout and size of
 * pq_build_table_ext), for global memory: what a stage's LDS table does not reach -- linear light below 2^-24 coming out of
 * a source function, PQ code values below 2^-12 -- is answered from here (tfn_ext_inline / tfn_ext_gather, h2y_device.h)
 * before the double-double careful tier (30 us of a wave's time per sample).  Segments the polynomial cannot be trusted on
 * carry the sentinel as in the LDS tables.  H2Y_TFN_PQ_R's is pq_build_table_ext() itself. */
inline int tfn_build_ext(int fn, pq_ext_rec *X)
{
    const tfn_cut cut = {H2Y_PQ_SEG_BITS, H2Y_PQX_EMIN, H2Y_PQ_SEG_BITS, 1};
    std::vector<pq_recA> A(H2Y_PQX_NSEG);
    std::vector<pq_recB> B(H2Y_PQX_NSEG);
    const int nbad = tfn_build_records(fn, cut, H2Y_PQX_NSEG - 1, A.data(), B.data());
    for (int i = 0; i < H2Y_PQX_NSEG; i++) {
        X[i].a = A[i];
        X[i].b = B[i];
    }
    return nbad;
}
/* lowest float (bits) of a function's LDS table: samples below it are the ones its full-range table is asked for */
H2Y_FN uint32_t tfn_lo_bits(int fn) { return (uint32_t)(127 + tfn_cut_of(fn).emin) << 23; }
/* one stage through its table (A, then B = A + H2Y_PQ_NREC records): pq_fast() with the function's own cut and value
 * at +0.0 */
H2Y_FN float tfn_fast(float x, const pq_recA *__restrict__ A, tfn_cut cut, uint32_t zero_bits, uint32_t one_bits, bool *slow)
{
    int sbits;
    const uint32_t bits = f2bits(x), idx = tfn_index(bits, cut, &sbits);
    const pq_recA a = A[idx];
    const pq_recB b = reinterpret_cast<const pq_recB *>(A + H2Y_PQ_NREC)[idx];
    const double v = tfn_poly(bits, sbits, a, b);
    const bool zero = bits == 0u, one = one_bits != 0u && bits == 0x3F800000u;
    *slow = pq_ambiguous(v) & !(zero | one);
    return zero ? bits2f(zero_bits) : one ? bits2f(one_bits) : (float)v;
}

/* x - floor(x) in [0,1) (v_fract_f32) */
H2Y_FN float fract_f32(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fractf(v);
#else
    float fr = v - __builtin_floorf(v);
    return fr >= 1.0f ? 0x1.fffffep-1f : fr;
#endif
}
/* ------------------------------------------------------------------------
 * First tier ("T1"): binary32 only, one 16-byte record per sample.
 *
 * 256 segments per binade of the input (exponent + top 8 mantissa bits; 6401 records, 100 KB).  All the
 * reference's floats of one segment lie in one binade of the OUTPUT (ulp U) -- the few segments in which
 * the output passes a power of two, and those above 1 + 2^-8, carry a record whose value is NaN (sentinels)
 * and go to the binary64 tier.  The record holds a float c of that binade (its bit pattern plus one, "cb") and the segment's
 * polynomial in units of U, measured from c and shifted by delta - 1/2:
 *     t = (value - c) / U - 1/2 + delta  =  c0 + u (c1 + u c2)     (binary32; u exact)
 * so that   RN(value) = c + (floor(t) + 1) U,   bit pattern cb + floor(t),   one integer add,
 * whenever value / U is further than delta from a rounding tie, i.e. fract(t) >= 2 delta: a test against
 * ONE constant, because t is in ulps already (v_cvt_flr_i32_f32, v_add_u32, v_fract_f32, v_cmp: four
 * instructions after the polynomial; the earlier form -- c0h + w in floats, the exact rounding error of
 * that sum against a threshold made from the sum's exponent -- took five).
 * delta = H2Y_T1_DELTA bounds |t as computed - t of the reference's double| over EVERY float of the
 * domain (tools/pq_check t1 measures it and checks that every sample passing the test is the
 * reference's float); about 1 % of samples fail the test (*unsure: the result may be one ulp high); their pixel
 * is redone by the binary64 tier only when its integers are sensitive to a one-ulp change of V
 * (pix_matrix_t1).
 * ---------------------------------------------------------------------- */
#define H2Y_T1_SEG_BITS 8
#define H2Y_T1_LOW_BITS (23 - H2Y_T1_SEG_BITS)
#define H2Y_T1_NSEG (H2Y_PQ_NBINADES << H2Y_T1_SEG_BITS)
#define H2Y_T1_NREC (H2Y_T1_NSEG + 2) /* record 0 and record NSEG+1 are the sentinels */
#define H2Y_T1_BASE ((uint32_t)(127 + H2Y_PQ_EMIN) << H2Y_T1_SEG_BITS)
/* smallest input of the table: 2^EMIN */
#define H2Y_T1_DOMAIN_LO 0x1p-24f
/* largest: the segment that starts at 1.0 is the last with a polynomial (values above 10 000 cd/m2 are not pictures; and
 * with V <= PQ(1 + 2^-8) the chroma integers provably need no clamp: t1_chroma_in_range()) */
#define H2Y_T1_DOMAIN_HI (1.0f + 0x1p-8f)
#define H2Y_T1_DELTA 0.0040f /* ulps; measured max 0.00345 over every float of the domain (tools/pq_check t1), +15 % */
struct alignas(16) pq_rec1 {
    float cb; /* bit pattern: the base float's + 1 */
    float c0, c1, c2;
};
/* the polynomial's variable: the sample's low mantissa bits, centred */
H2Y_FN float pq_t1_u(uint32_t bits)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t fb; /* one v_and_or_b32 (the compiler splits it into and + or) */
    asm("v_and_or_b32 %0, %1, %2, 1.0" : "=v"(fb) : "v"(bits), "s"((1u << H2Y_T1_LOW_BITS) - 1u));
    float f = bits2f(fb);
#else
    float f = bits2f((bits & ((1u << H2Y_T1_LOW_BITS) - 1u)) | 0x3F800000u);
#endif
    return f - (1.0f + 1.0f / (float)(2 << H2Y_T1_SEG_BITS));
}
/* t of the comment above, as computed; shared with tools */
H2Y_FN float pq_t1_t(uint32_t bits, const pq_rec1 &r)
{
    const float u = pq_t1_u(bits);
    return __builtin_fmaf(__builtin_fmaf(r.c2, u, r.c1), u, r.c0);
}
H2Y_FN int32_t floor_i32_f32(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
#else
    return f != f ? 0 : (int32_t)__builtin_floorf(f); /* |t| < 2^16 for every record with a polynomial; NaN -> 0 as the instruction does */
#endif
}
/* Byte offset of the sample's record.  The sample is first clamped, as a float, between the two
 * sentinel inputs -- the last float segment below the table and 2.0, the first above it -- so
 * zero, tiny, negative (all below), >= 2, +inf (above) and NaN (v_med3_f32 returns the minimum of
 * the other two) all land on a sentinel record (NaN value: never "sure").  Three instructions:
 * v_med3_f32, v_lshrrev_b32, v_lshl_add_u32. */
#define H2Y_T1_LO_SENTINEL_BITS ((H2Y_T1_BASE - 1u) << H2Y_T1_LOW_BITS)
H2Y_FN uint32_t pq_t1_offset(float x)
{
    const float lo = bits2f(H2Y_T1_LO_SENTINEL_BITS);
#if defined(__HIP_DEVICE_COMPILE__)
    const float c = __builtin_amdgcn_fmed3f(x, lo, 2.0f);
#else
    const float c = !(x > lo) ? lo : (x < 2.0f ? x : 2.0f); /* NaN -> lo */
#endif
    return ((f2bits(c) >> H2Y_T1_LOW_BITS) << 4) - ((H2Y_T1_BASE - 1u) << 4);
}
/* the two halves of pq_t1(): the record's fetch, and the evaluation -- so that a kernel can have
 * several records on their way from LDS before the first is used */
H2Y_FN pq_rec1 pq_t1_fetch(float x, const pq_rec1 *__restrict__ T)
{
#if defined(__HIP_DEVICE_COMPILE__)
    /* T is in LDS.  Address = (index << 4) + (table - first index * 16) as ONE v_lshl_add_u32; written
     * out because the compiler prefers shift-right 11, and-not 15, add. */
    typedef const __attribute__((address_space(3))) pq_rec1 *lds_rec;
    const uint32_t tbase = (uint32_t)(uintptr_t)(lds_rec)T - ((H2Y_T1_BASE - 1u) << 4);
    uint32_t seg = f2bits(__builtin_amdgcn_fmed3f(x, bits2f(H2Y_T1_LO_SENTINEL_BITS), 2.0f)) >> H2Y_T1_LOW_BITS;
#ifdef H2Y_EXP_NOCONFLICT /* timing experiment only (wrong records): what would conflict-free table reads buy? */
    seg = (seg & ~15u) | (__builtin_amdgcn_mbcnt_lo(~0u, 0u) & 15u);
#endif
    uint32_t addr;
    asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(addr) : "v"(seg), "s"(tbase));
    return *(lds_rec)(uintptr_t)addr;
#else
    return *reinterpret_cast<const pq_rec1 *>(reinterpret_cast<const char *>(T) + pq_t1_offset(x));
#endif
}
/* *margin = fract(t): the sample is unsure iff it is below H2Y_T1_SURE (a kernel takes the minimum of a pixel's three
 * margins first: one v_min3_f32 and one compare per pixel instead of three compares).  A sentinel record gives t = 0. */
#define H2Y_T1_SURE (2.0f * H2Y_T1_DELTA)
H2Y_FN float pq_t1_eval_m(float x, const pq_rec1 &r, float *margin)
{
    const float t = pq_t1_t(f2bits(x), r);
    *margin = fract_f32(t);
    return bits2f(f2bits(r.cb) + (uint32_t)floor_i32_f32(t));
}
H2Y_FN bool pq_t1_unsure(float margin) { return !(margin >= H2Y_T1_SURE); }
H2Y_FN bool pq_t1_unsure3(float mg, float mb, float mr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float m; /* (margins are never NaN: t of a record is finite for every u) */
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(mg), "v"(mb), "v"(mr));
    return !(m >= H2Y_T1_SURE);
#else
    const float m = mg < mb ? (mg < mr ? mg : mr) : (mb < mr ? mb : mr);
    return !(m >= H2Y_T1_SURE);
#endif
}
H2Y_FN float pq_t1_eval(float x, const pq_rec1 &r, bool *unsure)
{
    float m;
    const float v = pq_t1_eval_m(x, r, &m);
    *unsure = pq_t1_unsure(m);
    return v;
}
H2Y_FN float pq_t1(float x, const pq_rec1 *__restrict__ T, bool *unsure) { return pq_t1_eval(x, pq_t1_fetch(x, T), unsure); }
inline void pq_build_table1(pq_rec1 *T)
{
    const double wn[3] = {-0.8660254037844386, 0.0, 0.8660254037844386};
    const pq_rec1 nan_rec = {bits2f(0x7FC00000u), 0.0f, 0.0f, 0.0f}; /* t = 0 whatever u is: value NaN, margin 0 = never sure */
    for (int i = 0; i < H2Y_T1_NSEG; i++) {
        pq_rec1 &o = T[i + 1]; /* record 0 is the low sentinel */
        int e = H2Y_PQ_EMIN + (i >> H2Y_T1_SEG_BITS);
        int sg = i & ((1 << H2Y_T1_SEG_BITS) - 1);
        const uint32_t xlo = ((uint32_t)(127 + e) << 23) | ((uint32_t)sg << H2Y_T1_LOW_BITS), xhi = xlo + (1u << H2Y_T1_LOW_BITS) - 1u;
        if (bits2f(xlo) >= H2Y_T1_DOMAIN_HI) { o = nan_rec; continue; }
        /* the output's binade: that of the segment's first value; its last may be the next power of two itself (the bit
         * patterns run on), not more */
        const uint32_t vlo = f2bits((float)pq_exact_dd(dd{(double)bits2f(xlo), 0.0}).hi), vhi = f2bits((float)pq_exact_dd(dd{(double)bits2f(xhi), 0.0}).hi);
        if ((vhi >> 23) != (vlo >> 23) && vhi != ((vlo >> 23) + 1u) << 23) { o = nan_rec; continue; }
        const double inv_u = bits2d((uint64_t)(1023 + 23 + 127 - (int)(vlo >> 23)) << 52); /* 1 / U */
        double scale = bits2d((uint64_t)(1023 + e) << 52);
        double mid = scale * (1.0 + (sg + 0.5) / (1 << H2Y_T1_SEG_BITS));
        double half = scale * (0.5 / (1 << H2Y_T1_SEG_BITS));
        dd ev[3];
        for (int j = 0; j < 3; j++) ev[j] = pq_exact_dd(dd_add_d(two_prod(wn[j], half), mid));
        for (int lvl = 1; lvl < 3; lvl++)
            for (int j = 2; j >= lvl; j--) {
                dd num = dd_add(ev[j], dd{-ev[j - 1].hi, -ev[j - 1].lo});
                ev[j] = dd_div(num, dd{wn[j] - wn[j - lvl], 0.0});
            }
        /* p(w) = ev0 + (w-w0) ev1 + (w-w0)(w-w1) ev2, with (w-w0)(w-w1) = w^2 - (w0+w1) w + w0 w1 */
        const dd q2 = ev[2];
        const dd q1 = dd_add(ev[1], dd_mul_d(q2, -(wn[0] + wn[1])));
        const dd q0 = dd_add(dd_add(ev[0], dd_mul_d(ev[1], -wn[0])), dd_mul_d(q2, wn[0] * wn[1]));
        const double r = (double)(2 << H2Y_T1_SEG_BITS);
        const float c = (float)q0.hi; /* in the segment's range of values: of the output's binade (or its upper end) */
        o.cb = bits2f(f2bits(c) + 1u);
        o.c0 = (float)((((q0.hi - (double)c) + q0.lo) * inv_u - 0.5) + (double)H2Y_T1_DELTA);
        o.c1 = (float)(q1.hi * r * inv_u);
        o.c2 = (float)(q2.hi * r * r * inv_u);
    }
    T[0] = T[H2Y_T1_NSEG + 1] = nan_rec;
}
/* ------------------------------------------------------------------------
 * Per-frame constants handed to the kernels.
 * ---------------------------------------------------------------------- */
enum : int { H2Y_MODE_IDENTITY = 0, H2Y_MODE_YDZDX = 1, H2Y_MODE_YCBCR = 2, H2Y_MODE_YPQRS = 3 };

struct pix_params {
    /* matrix_convert */
    int convert_transfer;      /* convert.cpp:930.  0: transfers equal; 1: LINEAR -> PQ (fast tier available);
                                  2: any other pair the reference has code for (careful tier only) */
    int src_tf, dst_tf;        /* H2Y_TF_* classes of the two transfers */
    int src_fn, dst_fn;        /* convert_transfer == 2: H2Y_TFN_* of the two stages' tables in LDS (source table first), 0 = no such stage,
                                  -1 = no fast tier for this pair (careful tier for every sample) */
    int norm_identity;         /* offset 0 and range 1 for all three planes */
    float offset[3], range[3]; /* convert.cpp:939-940 */
    float mulY, addY, mulC, addC; /* scale step convert.cpp:1123-1145; G: mulY/addY, B and R: mulC/addC */
    int mode;                  /* H2Y_MODE_*, convert.cpp:1159-1198 */
    double kr, kg, kb;         /* luma weights as written in the reference */
    double dcb, dcr;           /* chroma divisors */
    double inv_dcb, inv_dcr;
    float P, Q, RR, S;         /* convert.cpp:913-925 */
    uint32_t half_m1;          /* clip->Half - 1, convert.cpp:1200 */
    uint32_t maxCV;            /* tmp picture's, convert.cpp:1207 */
    float fir_max;             /* convert(): FIR clamp, (float)maxCV of the tmp picture */
    /* write_yuv, tiff.cpp:394,469-478 with the OUTPUT picture's limits
     * (0..maxCV when the output is full range) */
    int down_shift;
    uint32_t ylo, yhi, clo, chi;
    /* the same limits on the UNSHIFTED value (pix_limits_finish): lo << s and (hi << s) | (2^s - 1),
     * and for the 2x2 box sum of four chroma values (s + 2) */
    uint32_t ylo_s, yhi_s, clo_s, chi_s, clo_b, chi_b;
    const void *pq_ext;        /* pq_build_table_ext() in device memory (NULL: none): samples below the LDS tables */
    const void *tf_ext[2];     /* convert_transfer == 2: tfn_build_ext() of the source and destination stage (NULL: none) */
};
inline void pix_limits_finish(pix_params *pp)
{
    const int s = pp->down_shift;
    pp->ylo_s = pp->ylo << s; pp->yhi_s = (pp->yhi << s) | ((1u << s) - 1u);
    pp->clo_s = pp->clo << s; pp->chi_s = (pp->chi << s) | ((1u << s) - 1u);
    pp->clo_b = pp->clo << (s + 2); pp->chi_b = (pp->chi << (s + 2)) | ((1u << (s + 2)) - 1u);
}

/* ---- conversions with the hardware's (defined) saturating behaviour --------
 * C leaves float->int casts undefined outside the target range; the kernels
 * use the gfx950 instructions directly (NaN -> 0, saturate at the ends) and
 * the host mirror spells the same function out. */
H2Y_FN int32_t sat_i32_f32(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
#else
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)f;
#endif
}
H2Y_FN int32_t sat_i32_f64(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t r;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(v));
    return r;
#else
    if (v != v) return 0;
    if (v >= 2147483648.0) return 2147483647;
    if (v <= -2147483648.0) return (int32_t)0x80000000;
    return (int32_t)v;
#endif
}
/* (int)v as the reference's x86-64 build computes it (cvttsd2si): NaN gives INT_MIN, the
 * "integer indefinite".  The saturating convert gives 0 for NaN; out-of-range values differ too
 * (INT_MAX instead of INT_MIN) but both clamp to maxCV in chroma_clamped().  Careful tier only:
 * every NaN reaches it (a NaN PQ value is never "sure", a NaN quotient never passes the guard).
 * NaNs are ordinary here: pic_stats' (int)max is 0 for a picture whose maximum is below 1.0,
 * the range becomes 0 and matrix_convert() divides every sample by it (convert.cpp:939-1019). */
H2Y_FN int32_t x86_i32_f64(double v)
{
    const int32_t r = sat_i32_f64(v);
    return v != v ? (int32_t)0x80000000 : r;
}
/* x - floor(x), in [0,1) (v_fract_f64) */
H2Y_FN double fract_f64(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fract(v);
#else
    double fr = v - __builtin_floor(v);
    return fr >= 1.0 ? 0x1.fffffffffffffp-1 : fr;
#endif
}

/* (unsigned int)f followed by the "> maxCV" clamp of convert.cpp:1207.
 * Pinned domain: finite 0 <= f < 2^31, where this is C truncation.
 * Outside it the C cast is undefined; the reference's x86-64 build takes the
 * low word of a 64-bit cvttss2si: -1 < f < 0 gives 0, f <= -1 wraps to a huge
 * unsigned value that the clamp turns into maxCV, NaN gives 0.  The signed
 * saturating convert reproduces all of that: a negative integer read as
 * unsigned is huge.  (f >= 2^31 saturates to maxCV here; unpinned.) */
H2Y_FN uint32_t f2u_clamped(float f, uint32_t maxCV)
{
    uint32_t u = (uint32_t)sat_i32_f32(f);
    return u < maxCV ? u : maxCV;
}
/* (long)Cb + Half - 1, then the unsigned compare against maxCV (convert.cpp:
 * 1200-1212): a negative sum wraps high and clamps to maxCV.
 * 32-bit wrap-around gives the same result as the reference's 64-bit one for
 * every int32 n (Half - 1 < 2^15). */
H2Y_FN uint32_t chroma_clamped(int32_t n, uint32_t half_m1, uint32_t maxCV)
{
    uint32_t c = (uint32_t)n + half_m1;
    return c < maxCV ? c : maxCV;
}

/* q = d * (1/c) + 0.5 (one fma) stands in for the reference's
 * RN(RN(d/c) + 0.5); the two differ by at most a couple of ulps (< 2^-34 for
 * |q| < 2^18), which can only change the truncated integer when q is that
 * close to an integer.  *uncertain is set when the fractional part is below
 * 2^-30 or above 1 - 2^-21 (test on the high word of fract(q)). */
H2Y_FN int32_t div_round_trunc_fast(float df, double inv_c, bool *uncertain)
{
    double q = __builtin_fma((double)df, inv_c, 0.5);
    uint32_t fh = (uint32_t)(d2bits(fract_f64(q)) >> 32);
    *uncertain = (fh - 0x3E100000u) >= (0x3FEFFFFFu - 0x3E100000u);
    return sat_i32_f64(q);
}

/* matrix_convert() from scaled code values to clamped 4:4:4 integers,
 * convert.cpp:1150-1221.  G,B,R are the float code values after the scale
 * step (or the raw samples when convert_transfer == 0).  MODE is one of
 * H2Y_MODE_*; H2Y_MODE_RUNTIME reads pp.mode.  EXACT_DIV = true performs the
 * reference's IEEE divisions (careful tier); false uses the reciprocal form
 * and reports through *uncertain when that cannot be trusted. */
#define H2Y_MODE_RUNTIME (-1)
template <int MODE, bool EXACT_DIV>
H2Y_FN void pix_matrix(const pix_params &pp, float G, float B, float R, uint32_t &Yo, uint32_t &Cbo, uint32_t &Cro,
                       bool *uncertain)
{
    const int mode = MODE == H2Y_MODE_RUNTIME ? pp.mode : MODE;
    *uncertain = false;
    if (mode == H2Y_MODE_IDENTITY) {
        Yo = f2u_clamped(G, pp.maxCV);
        Cbo = f2u_clamped(B, pp.maxCV);
        Cro = f2u_clamped(R, pp.maxCV);
        return;
    }
    int32_t cb, cr;
    if (mode == H2Y_MODE_YCBCR) {
        double yd = ((pp.kr * (double)R + pp.kg * (double)G) + pp.kb * (double)B) + 0.5;
        float tmpF = (float)yd;
        Yo = f2u_clamped(tmpF, pp.maxCV);
        if (EXACT_DIV) {
            cb = x86_i32_f64((double)(B - tmpF) / pp.dcb + 0.5);
            cr = x86_i32_f64((double)(R - tmpF) / pp.dcr + 0.5);
        } else {
            bool u1, u2;
            cb = div_round_trunc_fast(B - tmpF, pp.inv_dcb, &u1);
            cr = div_round_trunc_fast(R - tmpF, pp.inv_dcr, &u2);
            *uncertain = u1 | u2;
        }
    } else if (mode == H2Y_MODE_YDZDX) {
        Yo = f2u_clamped(G, pp.maxCV);
        double hg = (double)(-G) * 0.5; /* -G/2.0: exact either way */
        cb = EXACT_DIV ? x86_i32_f64((hg + (double)B * 0.5) + 0.5) : sat_i32_f64((hg + (double)B * 0.5) + 0.5);
        cr = EXACT_DIV ? x86_i32_f64((hg + (double)R * 0.5) + 0.5) : sat_i32_f64((hg + (double)R * 0.5) + 0.5);
    } else {
        Yo = f2u_clamped(G, pp.maxCV);
        cb = EXACT_DIV ? x86_i32_f64((double)(pp.P * G + pp.Q * B) + 0.5) : sat_i32_f64((double)(pp.P * G + pp.Q * B) + 0.5);
        cr = EXACT_DIV ? x86_i32_f64((double)(pp.RR * R + pp.S * G) + 0.5) : sat_i32_f64((double)(pp.RR * R + pp.S * G) + 0.5);
    }
    Cbo = chroma_clamped(cb, pp.half_m1, pp.maxCV);
    Cro = chroma_clamped(cr, pp.half_m1, pp.maxCV);
}

/* ---- first-tier (T1) form of the matrix step ------------------------------
 * Same arithmetic as pix_matrix<MODE,false>.  When one of the pixel's three PQ
 * values came out of pq_t1() "unsure" (it may be one float ulp off), the
 * integers are accepted only if none of the three pre-truncation values is
 * close enough to an integer for such a change to matter (bounds: t1_bounds()).
 * Returns true when the pixel must be redone by the binary64 tier.
 * The chroma comes out RAW: the truncated integers themselves, signed, in two's complement -- without the
 * reference's "+ Half - 1" and its unsigned clamp to maxCV (convert.cpp:1200-1212, chroma_clamped()).  For a
 * pixel this tier settles the clamp never acts (its three PQ values are table values: t1_bounds() proves the
 * range per launch, or the launch does not use this tier), and the offset is linear: the kernels add it once
 * per 2x2 box or fold it into the FIR's rounding constant instead of once per sample -- four instructions
 * per pixel less. */
struct t1_sens {
    float ty;                 /* luma safe iff |fract(y) - 0.5| < ty                     */
    uint32_t cb_lo, cb_span;  /* chroma unsafe iff (hiword(fract(q)) - lo) >= span (wide) */
    uint32_t cr_lo, cr_span;
    uint32_t c_lo, c_span;    /* one window covering both (the wider of the two) */
    uint32_t a_lo, a_hi;      /* YCbCr: one window for luma and both chromas, on hiword(fract()) of the three binary64
                                 pre-truncation values: unsafe iff < a_lo or >= a_hi */
};
#define H2Y_GUARD_LO 0x3E100000u                  /* hiword(2^-30): the always-on guard of the reciprocal division */
#define H2Y_GUARD_SPAN (0x3FEFFFFFu - 0x3E100000u) /* up to hiword(1 - 2^-21) */
#define H2Y_GUARD_HI 0x3FEFFFFFu

/* two-condition form: the pixel must be redone iff *ra | *rb (kept apart for the kernel: each is the
 * lane mask of one compare, and ORing lane masks is scalar work) */
template <int MODE>
H2Y_FN void pix_matrix_t1(const pix_params &pp, const t1_sens &sn, float G, float B, float R, bool vunc, uint32_t &Yo,
                          uint32_t &Cbo, uint32_t &Cro, bool *ra, bool *rb)
{
    float ylike;
    double qb, qr;
    if (MODE == H2Y_MODE_YCBCR) {
        const double yd = ((pp.kr * (double)R + pp.kg * (double)G) + pp.kb * (double)B) + 0.5;
        const float tmpF = (float)yd;
        qb = __builtin_fma((double)(B - tmpF), pp.inv_dcb, 0.5);
        qr = __builtin_fma((double)(R - tmpF), pp.inv_dcr, 0.5);
        Yo = (uint32_t)sat_i32_f32(tmpF); /* unclamped: the caller's pix_yuv_clamp() bounds it below maxCV anyway (yhi_s <= maxCV) */
        Cbo = (uint32_t)sat_i32_f64(qb);  /* raw: see above */
        Cro = (uint32_t)sat_i32_f64(qr);
        /* One test for the three values: the high words of their fractional parts against one window --
         * with an unsure sample the wide one (one-ulp sensitivity of any of the three), without one the
         * narrow one inside it (the guard of the reciprocal division; harmless for the luma, whose float
         * is exact then).  The luma is judged by its binary64 sum: Y = trunc(RN_float(yd)), and the
         * window's margin includes that rounding (t1_bounds).  NaN => high word above every window. */
        const uint32_t fy = (uint32_t)(d2bits(fract_f64(yd)) >> 32);
        const uint32_t fb = (uint32_t)(d2bits(fract_f64(qb)) >> 32), fr = (uint32_t)(d2bits(fract_f64(qr)) >> 32);
        const uint32_t lo = vunc ? sn.a_lo : H2Y_GUARD_LO, hi = vunc ? sn.a_hi : H2Y_GUARD_HI;
        *ra = umin3(fy, fb, fr) < lo;
        *rb = umax3(fy, fb, fr) >= hi;
        return;
    } else { /* H2Y_MODE_YDZDX */
        ylike = G;
        double hg = (double)(-G) * 0.5;
        qb = (hg + (double)B * 0.5) + 0.5;
        qr = (hg + (double)R * 0.5) + 0.5;
    }
    /* Y is left unclamped: the caller's pix_yuv_clamp() bounds it below maxCV anyway (yhi_s <= maxCV) */
    Yo = (uint32_t)sat_i32_f32(ylike);
    Cbo = (uint32_t)sat_i32_f64(qb);
    Cro = (uint32_t)sat_i32_f64(qr);
    const uint32_t fb = (uint32_t)(d2bits(fract_f64(qb)) >> 32), fr = (uint32_t)(d2bits(fract_f64(qr)) >> 32);
    const bool y_unsafe = !(__builtin_fabsf(fract_f32(ylike) - 0.5f) < sn.ty); /* NaN => unsafe */
    /* YDzDx has no division, so without an unsure sample its chroma needs no guard at all */
    const bool guard = vunc & (((fb - sn.cb_lo) >= sn.cb_span) | ((fr - sn.cr_lo) >= sn.cr_span));
    *ra = guard | (vunc & y_unsafe);
    *rb = false;
}
template <int MODE>
H2Y_FN bool pix_matrix_t1(const pix_params &pp, const t1_sens &sn, float G, float B, float R, bool vunc, uint32_t &Yo,
                          uint32_t &Cbo, uint32_t &Cro)
{
    bool ra, rb;
    pix_matrix_t1<MODE>(pp, sn, G, B, R, vunc, Yo, Cbo, Cro, &ra, &rb);
    return ra | rb;
}

/* Host: how far a one-ulp change of each PQ value can move the pre-truncation
 * values (u23 = 2^-23 bounds one ulp relative to the magnitude):
 *   s = RN(RN(V mul) + add): the product moves by mul ulp(V) and may round one
 *       ulp further, the sum may round one ulp further: <= 3 u23 smax
 *   y (YCbCr): sum of the three weighted by k (sum 1), + one ulp of tmpF
 *   d = RN(s - tmpF): both moves + one ulp; q = d / c
 * 10 % slack on top.  Returns false when the windows get so wide that most
 * pixels with an unsure sample would be redone anyway (high bit depths). */
H2Y_FN uint32_t hiword_of(double v) { return (uint32_t)(d2bits(v) >> 32); }
/* Host: do the chroma integers of every pixel the first tier settles lie in [-(Half - 1), maxCV - (Half - 1)], where
 * chroma_clamped() is the identity?  Such a pixel's PQ values are table values, V in [PQ(2^-24), PQ(1 + 2^-8)]; interval
 * arithmetic through the scale step and the matrix, with 0.02 + 1e-5 |q| for every rounding on the way (a code value's
 * float ulp is 2^-12 at 12 bits, 2^-8 at 16). */
inline bool t1_chroma_in_range(const pix_params &pp)
{
    const double vhi = pq_exact_dd(dd{(double)H2Y_T1_DOMAIN_HI, 0.0}).hi * (1.0 + 1e-6), vlo = pq_exact_dd(dd{(double)H2Y_T1_DOMAIN_LO, 0.0}).hi * (1.0 - 1e-6);
    auto span = [&](double mul, double add, double *lo, double *hi) {
        const double a = add + mul * vlo, b = add + mul * vhi;
        *lo = (a < b ? a : b) - 1e-3 - 1e-6 * __builtin_fabs(a < b ? a : b);
        *hi = (a < b ? b : a) + 1e-3 + 1e-6 * __builtin_fabs(a < b ? b : a);
    };
    double glo, ghi, clo, chi;
    span(pp.mulY, pp.addY, &glo, &ghi);
    span(pp.mulC, pp.addC, &clo, &chi);
    double qlo[2], qhi[2];
    if (pp.mode == H2Y_MODE_YCBCR) {
        if (!(pp.kr >= 0 && pp.kg >= 0 && pp.kb >= 0 && pp.kb <= 1 && pp.kr <= 1 && pp.inv_dcb > 0 && pp.inv_dcr > 0)) return false;
        /* d = B - tmpF = (1 - kb) B - kr R - kg G - 0.5 (and the same with R) */
        const double db_hi = (1 - pp.kb) * chi - pp.kr * clo - pp.kg * glo - 0.5, db_lo = (1 - pp.kb) * clo - pp.kr * chi - pp.kg * ghi - 0.5;
        const double dr_hi = (1 - pp.kr) * chi - pp.kb * clo - pp.kg * glo - 0.5, dr_lo = (1 - pp.kr) * clo - pp.kb * chi - pp.kg * ghi - 0.5;
        qhi[0] = db_hi * pp.inv_dcb + 0.5; qlo[0] = db_lo * pp.inv_dcb + 0.5;
        qhi[1] = dr_hi * pp.inv_dcr + 0.5; qlo[1] = dr_lo * pp.inv_dcr + 0.5;
    } else if (pp.mode == H2Y_MODE_YDZDX) {
        qhi[0] = qhi[1] = (chi - glo) * 0.5 + 0.5;
        qlo[0] = qlo[1] = (clo - ghi) * 0.5 + 0.5;
    } else return false;
    const double top = (double)pp.maxCV - (double)pp.half_m1, bot = -(double)pp.half_m1;
    for (int k = 0; k < 2; k++) {
        const double hi = qhi[k] + 0.02 + 1e-5 * __builtin_fabs(qhi[k]), lo = qlo[k] - 0.02 - 1e-5 * __builtin_fabs(qlo[k]);
        if (!(hi < top + 1.0 && lo > bot - 1.0)) return false; /* truncation towards zero */
    }
    return true;
}
inline bool t1_bounds(const pix_params &pp, t1_sens *sn)
{
    const double u23 = 0x1p-23;
    if (!t1_chroma_in_range(pp)) return false;
    const double sY = (double)pp.mulY * 1.0005 + pp.addY, sC = (double)pp.mulC * 1.0005 + pp.addC; /* PQ <= PQ(1 + 2^-8) = 1.000408 for every sample the first tier answers (H2Y_T1_DOMAIN_HI) */
    const double smax = sY > sC ? sY : sC;
    double Ey, Ecb, Ecr;
    if (pp.mode == H2Y_MODE_YCBCR) {
        const double ymax = smax + 0.5;
        Ey = (3 * u23 * smax + u23 * ymax) * 1.1;
        const double Ed = (3 * u23 * smax + 4 * u23 * ymax + u23 * ymax) * 1.1;
        Ecb = Ed / pp.dcb;
        Ecr = Ed / pp.dcr;
    } else if (pp.mode == H2Y_MODE_YDZDX) {
        Ey = 3 * u23 * sY * 1.1;
        Ecb = Ecr = (3 * u23 * sY + 3 * u23 * sC) * 0.5 * 1.1;
    } else return false;
    sn->ty = (float)(0.5 - Ey);
    sn->cb_lo = hiword_of(Ecb) + 1;
    sn->cb_span = hiword_of(1.0 - Ecb) - sn->cb_lo;
    sn->cr_lo = hiword_of(Ecr) + 1;
    sn->cr_span = hiword_of(1.0 - Ecr) - sn->cr_lo;
    {
        const uint32_t hb = sn->cb_lo + sn->cb_span, hr = sn->cr_lo + sn->cr_span;
        sn->c_lo = sn->cb_lo > sn->cr_lo ? sn->cb_lo : sn->cr_lo; /* safe region = intersection of the two */
        sn->c_span = (hb < hr ? hb : hr) - sn->c_lo;
        /* and with the luma's: fract(yd) in [Ey, 1 - Ey) */
        const uint32_t y_lo = hiword_of(Ey) + 1, y_hi = hiword_of(1.0 - Ey);
        sn->a_lo = sn->c_lo > y_lo ? sn->c_lo : y_lo;
        sn->a_hi = (sn->c_lo + sn->c_span) < y_hi ? (sn->c_lo + sn->c_span) : y_hi;
    }
    /* ~3 % of pixels have an unsure sample; the share of pixels redone is ~0.03 * 2 (Ey + Ecb + Ecr):
     * 0.05 % at 12 bits (0.4 % of the eight-pixel tiles), 0.2 % at 14 and 0.8 % at 16 bits (tools/t1_check).
     * Redoing 64 collected tiles costs a wave about as much time as twenty ordinary tiles (scattered
     * accesses, one wave alone with the exact tiers): measured, 16 bits runs faster on k_fused (33 vs 45 us
     * per 4K frame), so the first tier is used up to 12 bits. */
    return (Ey + Ecb + Ecr) < 0.02;
}

/* scale step, convert.cpp:1123-1145: separate multiply and add in binary32.
 * Full range has no add in the reference; add is then 0.0f, which leaves every
 * product unchanged (-0.0 + 0.0 = +0.0 converts to the same integer). */
H2Y_FN float pix_scale(float v, float mul, float add)
{
    float r = v * mul;
    return r + add;
}

/* write_yuv per-sample step, tiff.cpp:469-478 (luma) / 502-511,533-543 (chroma).
 * lo/hi are minVR/maxVR (minVRC/maxVRC for chroma), or 0/maxCV when the output
 * is full range -- set up once in pix_params. */
H2Y_FN uint32_t umed3(uint32_t v, uint32_t lo, uint32_t hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "s"(lo), "v"(hi));
    return r;
#else
    return v < lo ? lo : (v > hi ? hi : v);
#endif
}
/* Clamping the unshifted value to [lo << s, (hi << s) | (2^s - 1)] and shifting afterwards gives the
 * same number as shifting first and clamping to [lo, hi]: one v_med3_u32 and one shift. */
template <bool NOSHIFT = false> /* NOSHIFT: the caller knows down_shift == 0 (float input) */
H2Y_FN uint32_t pix_yuv_clamp(const pix_params &pp, uint32_t v, bool chroma)
{
    const uint32_t c = umed3(v, chroma ? pp.clo_s : pp.ylo_s, chroma ? pp.chi_s : pp.yhi_s);
    return NOSHIFT ? c : c >> pp.down_shift;
}
/* the 2x2 box: (a+b+c+d)/4 (convert.cpp:157-160, unsigned truncation), write_yuv's shift and clamp */
template <bool NOSHIFT = false>
H2Y_FN uint32_t pix_box_clamp(const pix_params &pp, uint32_t sum4)
{
    const uint32_t c = umed3(sum4, pp.clo_b, pp.chi_b);
    return NOSHIFT ? c >> 2 : c >> (pp.down_shift + 2);
}

/* clamp to [0, maxCV] and truncate (convert.cpp:314-317, 372-374); t is never NaN (sums of finite samples) */
H2Y_FN uint32_t fir_clamp_trunc(float t, float fmaxcv)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_fmed3f(t, 0.0f, fmaxcv);
#else
    if (t > fmaxcv) t = fmaxcv;
    if (t < 0.0f) t = 0.0f;
    return (uint32_t)t;
#endif
}
/* Subsample444to420_FIR stage 1, convert.cpp:305-317.  s[-5..5] around an
 * even column, already edge-replicated.  float sums left to right, +0.5
 * (the reference's double add of 0.5 rounded back to float is the same
 * number as a float add: both addends are exact multiples of 2^-9 < 2^18). */
H2Y_FN uint32_t fir_h(float m5, float m3, float m1, float c, float p1, float p3, float p5, float fmaxcv)
{
    const float c21 = 21.0f / 512.0f, c52 = 52.0f / 512.0f, c159 = 159.0f / 512.0f, c256 = 256.0f / 512.0f;
    float acc = c21 * (m5 + p5) - c52 * (m3 + p3);
    acc = acc + c159 * (m1 + p1);
    acc = acc + c256 * c;
    return fir_clamp_trunc(acc + 0.5f, fmaxcv);
}

/* stage 2, convert.cpp:365-374: rows j-5..j+6 of the 4:2:2 intermediate */
H2Y_FN uint32_t fir_v(float m5, float m4, float m3, float m2, float m1, float m0, float p1, float p2, float p3,
                      float p4, float p5, float p6, float fmaxcv)
{
    const float c228 = 228.0f / 512.0f, c70 = 70.0f / 512.0f, c37 = 37.0f / 512.0f;
    const float c21 = 21.0f / 512.0f, c11 = 11.0f / 512.0f, c5 = 5.0f / 512.0f;
    float acc = c228 * (m0 + p1) + c70 * (m1 + p2);
    acc = acc - c37 * (m2 + p3);
    acc = acc - c21 * (m3 + p4);
    acc = acc + c11 * (m4 + p5);
    acc = acc + c5 * (m5 + p6);
    return fir_clamp_trunc(acc + 0.5f, fmaxcv);
}

/* ---- the same two stages in integers, for code values up to 14 bits ------------------------------
 * Every coefficient is k/512 and every sample an integer below 2^14, so each product, each partial sum and the
 * final "+0.5" of fir_h() / fir_v() is a multiple of 2^-9 below 2^15 in magnitude (the positive taps add up to
 * 616/512 and 628/512 of maxCV <= 16383: 19 711 and 20 095): all of them are exact in binary32, the float
 * expression IS the integer S/512 + 1/2, and clamp-then-truncate is med3(floor((S + 256) / 512), 0, maxCV).
 * (At 15 and 16 bits the sums pass 2^15 and round; the two-pass float form stays in charge there.)
 * tools/fir_int_check.cpp compares the two forms (tests/test_pq_math.py). */
H2Y_FN int32_t imed3(int32_t v, int32_t lo, int32_t hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t r; /* one scalar operand per instruction on this family: lo from a scalar register, hi from a vector one */
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "s"(lo), "v"(hi));
    return r;
#else
    return v < lo ? lo : (v > hi ? hi : v);
#endif
}
/* clamp to [0, hi] (hi uniform) */
H2Y_FN int32_t imed3_0(int32_t v, int32_t hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(hi));
    return r;
#else
    return v < 0 ? 0 : (v > hi ? hi : v);
#endif
}
#define H2Y_FIR_INT_MAX_DEPTH 14
H2Y_FN uint32_t fir_h_int(int32_t m5, int32_t m3, int32_t m1, int32_t c, int32_t p1, int32_t p3, int32_t p5, int32_t maxcv)
{
    const int32_t S = 21 * (m5 + p5) - 52 * (m3 + p3) + 159 * (m1 + p1) + 256 * c + 256;
    return (uint32_t)imed3(S >> 9, 0, maxcv);
}
/* lo/hi: [0, maxCV] for the bare stage, or write_yuv's chroma range (inside [0, maxCV]) to fold that clamp in */
H2Y_FN uint32_t fir_v_int(int32_t m5, int32_t m4, int32_t m3, int32_t m2, int32_t m1, int32_t m0, int32_t p1, int32_t p2, int32_t p3,
                          int32_t p4, int32_t p5, int32_t p6, int32_t lo, int32_t hi)
{
    const int32_t S = 228 * (m0 + p1) + 70 * (m1 + p2) - 37 * (m2 + p3) - 21 * (m3 + p4) + 11 * (m4 + p5) + 5 * (m5 + p6) + 256;
    return (uint32_t)imed3(S >> 9, lo, hi);
}

/* ---- Subsample420to444, convert.cpp:1869-1986 -----------------------------------------------
 * clamp to [lo, hi] and truncate (:1932-1934 and alike); t is never NaN (sums of finite samples) */
H2Y_FN uint32_t up_clamp_trunc(float t, float lo, float hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_fmed3f(t, lo, hi);
#else
    if (t > hi) t = hi;
    if (t < lo) t = lo;
    return (uint32_t)t;
#endif
}
/* vertical stage, :1925-1931 (even output rows: a..f = rows j-3..j+2) and :1936-1942 (odd rows: the same taps
 * mirrored, a..f = rows j+3..j-2).  Products and sums left to right in float; the reference's trailing double
 * "+0.5" rounded back to float is the float addition (one rounding of the exact sum either way). */
H2Y_FN uint32_t up_fir6(float a, float b, float c, float d, float e, float f, float lo, float hi)
{
    const float c3 = 3.0f / 256.0f, c16 = 16.0f / 256.0f, c67 = 67.0f / 256.0f, c227 = 227.0f / 256.0f, c32 = 32.0f / 256.0f,
                c7 = 7.0f / 256.0f;
    float acc = c3 * a - c16 * b;
    acc = acc + c67 * c;
    acc = acc + c227 * d;
    acc = acc - c32 * e;
    acc = acc + c7 * f;
    return up_clamp_trunc(acc + 0.5f, lo, hi);
}
/* horizontal stage, odd samples, :1972-1977: m0..m5 = intermediate columns i-2 .. i+3 */
H2Y_FN uint32_t up_fir_odd(float m0, float m1, float m2, float m3, float m4, float m5, float lo, float hi)
{
    const float c21 = 21.0f / 256.0f, c52 = 52.0f / 256.0f, c159 = 159.0f / 256.0f;
    float acc = c21 * (m0 + m5) - c52 * (m1 + m4);
    acc = acc + c159 * (m2 + m3);
    return up_clamp_trunc(acc + 0.5f, lo, hi);
}

} // namespace h2y
#endif /* H2Y_MATH_H */
