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

#include <stdint.h>

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
H2Y_FN_NOINLINE float pq_slow(float x)
{
    if (!(x >= 0.0f) || x > 3.4028234e38f) return bits2f(0x7FC00000u);
    double Ln = (x == 0.0f) ? 0.0 : pow_dd((double)x, H2Y_PQ_M1);
    double num = H2Y_PQ_C1 + H2Y_PQ_C2 * Ln;
    double den = 1.0 + H2Y_PQ_C3 * Ln;
    double B = num / den;
    double Vd = pow_dd(B, H2Y_PQ_M2);
    return (float)Vd;
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
 * neighbouring LDS bank groups:
 *     A[i] = { c0, c1 }           (2 x binary64)
 *     B[i] = { c2, c3|c4 }        (binary64, 2 x binary32)
 * value = c0 + u(c1 + u(c2 + u(c3 + u c4))),  u in [-1,1) across the segment.
 * ---------------------------------------------------------------------- */
#define H2Y_PQ_EMIN (-24)
#define H2Y_PQ_SEG_BITS 6
#define H2Y_PQ_SEG_PER_BINADE (1 << H2Y_PQ_SEG_BITS)
#define H2Y_PQ_NBINADES (1 - H2Y_PQ_EMIN) /* exponents EMIN..0 */
#define H2Y_PQ_NSEG (H2Y_PQ_NBINADES * H2Y_PQ_SEG_PER_BINADE)
#define H2Y_PQ_LOW_BITS (23 - H2Y_PQ_SEG_BITS)
#define H2Y_PQ_SEG_BASE ((uint32_t)(127 + H2Y_PQ_EMIN) << H2Y_PQ_SEG_BITS)
#define H2Y_PQ_TABLE_BYTES (H2Y_PQ_NSEG * 32)

/* A double's low 29 mantissa bits decide its rounding to float; the tie is
 * at 2^28.  The fast value is trusted when those bits are at least this far
 * (in units of the double's last place) from the tie: 2^12 ulp = 2^-40
 * relative at worst (mantissa in [1,2)), ~8x the measured fast-tier error. */
#define H2Y_PQ_AMBIG_ULPS 4096u

struct pq_recA {
    double c0, c1;
};
struct pq_recB {
    double c2;
    float c3, c4;
};

/* Fast tier.  Returns the float value; *slow is set when the caller must use
 * pq_slow(x) instead (x outside the table, or rounding too close to call). */
H2Y_FN float pq_fast(float x, const pq_recA *__restrict__ A, const pq_recB *__restrict__ B, bool *slow)
{
    uint32_t bits = f2bits(x);
    uint32_t t = bits - (H2Y_PQ_SEG_BASE << H2Y_PQ_LOW_BITS);
    bool inrange = t < ((uint32_t)H2Y_PQ_NSEG << H2Y_PQ_LOW_BITS); /* also rejects negatives, 0, NaN */
    uint32_t idx = inrange ? (t >> H2Y_PQ_LOW_BITS) : 0u;
    /* u = 2*frac - 1 over the segment, exact in binary32 */
    float f = bits2f((bits & ((1u << H2Y_PQ_LOW_BITS) - 1u)) | 0x3F800000u);
    float u = __builtin_fmaf(f, (float)(1 << (H2Y_PQ_SEG_BITS + 1)), -(float)((1 << (H2Y_PQ_SEG_BITS + 1)) + 1));
    pq_recA a = A[idx];
    pq_recB b = B[idx];
    float p = __builtin_fmaf(b.c4, u, b.c3);
    double ud = (double)u;
    double v = __builtin_fma((double)p, ud, b.c2);
    v = __builtin_fma(v, ud, a.c1);
    v = __builtin_fma(v, ud, a.c0);
    uint32_t lo = (uint32_t)d2bits(v);
    uint32_t dist = (lo + (H2Y_PQ_AMBIG_ULPS - 0x10000000u)) & 0x1FFFFFFFu;
    *slow = !inrange || dist < 2u * H2Y_PQ_AMBIG_ULPS;
    return (float)v;
}

/* Host-side table builder (context creation).  Per segment: interpolate the
 * double-double PQ at five near-Chebyshev nodes (Newton divided differences
 * in double-double), expand to monomials in u, round c0..c2 to binary64 and
 * c3,c4 to binary32.  Pure IEEE double arithmetic: the same table on every
 * host, no libm involved. */
inline void pq_build_table(pq_recA *A, pq_recB *B)
{
    const double un[5] = {-0.9510565162951535, -0.5877852522924731, 0.0, 0.5877852522924731, 0.9510565162951535};
    for (int i = 0; i < H2Y_PQ_NSEG; i++) {
        int e = H2Y_PQ_EMIN + i / H2Y_PQ_SEG_PER_BINADE;
        int s = i % H2Y_PQ_SEG_PER_BINADE;
        double scale = bits2d((uint64_t)(1023 + e) << 52);
        double mid = scale * (1.0 + (s + 0.5) / H2Y_PQ_SEG_PER_BINADE); /* exact */
        double half = scale * (0.5 / H2Y_PQ_SEG_PER_BINADE);            /* exact */
        dd dv[5];
        for (int j = 0; j < 5; j++) dv[j] = pq_exact_dd(dd_add_d(two_prod(un[j], half), mid));
        /* divided differences in u */
        for (int lvl = 1; lvl < 5; lvl++)
            for (int j = 4; j >= lvl; j--) {
                dd num = dd_add(dv[j], dd{-dv[j - 1].hi, -dv[j - 1].lo});
                dv[j] = dd_div(num, dd{un[j] - un[j - lvl], 0.0});
            }
        /* p(u) = dv0 + (u-u0)(dv1 + (u-u1)(dv2 + (u-u2)(dv3 + (u-u3) dv4))) -> monomials */
        dd c[5] = {dv[4], {0, 0}, {0, 0}, {0, 0}, {0, 0}};
        int deg = 0;
        for (int j = 3; j >= 0; j--) {
            /* c(u) <- c(u) * (u - u_j) + dv[j] */
            dd nc[5] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
            for (int k = 0; k <= deg; k++) {
                nc[k + 1] = dd_add(nc[k + 1], c[k]);
                nc[k] = dd_add(nc[k], dd_mul_d(c[k], -un[j]));
            }
            nc[0] = dd_add(nc[0], dv[j]);
            deg++;
            for (int k = 0; k <= deg; k++) c[k] = nc[k];
        }
        A[i].c0 = c[0].hi;
        A[i].c1 = c[1].hi;
        B[i].c2 = c[2].hi;
        B[i].c3 = (float)c[3].hi;
        B[i].c4 = (float)c[4].hi;
    }
}

/* ------------------------------------------------------------------------
 * Per-frame constants handed to the kernels.
 * ---------------------------------------------------------------------- */
enum : int { H2Y_MODE_IDENTITY = 0, H2Y_MODE_YDZDX = 1, H2Y_MODE_YCBCR = 2, H2Y_MODE_YPQRS = 3 };
enum : int { H2Y_SCALE_NONE = 0, H2Y_SCALE_FULL = 1, H2Y_SCALE_GBR = 2, H2Y_SCALE_YCC = 3 };

struct pix_params {
    /* matrix_convert */
    int convert_transfer;  /* convert.cpp:930 */
    int norm_identity;     /* offset 0 and range 1 for all three planes */
    float offset[3], range[3]; /* convert.cpp:939-940 */
    int scale_mode;        /* H2Y_SCALE_*, convert.cpp:1123-1145 */
    float mulY, addY, mulC, addC; /* G: mulY/addY, B and R: mulC/addC */
    int mode;              /* H2Y_MODE_*, convert.cpp:1159-1198 */
    double kr, kg, kb;     /* luma weights as written in the reference */
    double dcb, dcr;       /* chroma divisors */
    double inv_dcb, inv_dcr;
    float P, Q, RR, S;     /* convert.cpp:913-925 */
    uint32_t half_m1;      /* clip->Half - 1, convert.cpp:1200 */
    uint32_t maxCV;        /* tmp picture's, convert.cpp:1207 */
    /* convert(): FIR clamp uses the tmp picture's maxCV as float */
    float fir_max;
    /* write_yuv, tiff.cpp:394,469-478 with the OUTPUT picture's limits */
    int down_shift;
    int full_range;
    uint32_t ylo, yhi, clo, chi, out_maxCV;
};

/* float -> unsigned int.  Pinned domain: finite 0 <= f < 2^32,
 * where this is C truncation.  Outside it the C cast is undefined; the
 * reference's x86-64 build takes the low word of a 64-bit cvttss2si, which
 * this follows for -2^31 < f < 0 (wraps high, later clamps to maxCV) and for
 * NaN (0); f >= 2^32 saturates here (documented, unpinned). */
H2Y_FN uint32_t f2u_ref(float f)
{
    if (f >= 0.0f) return f < 4294967296.0f ? (uint32_t)f : 0xFFFFFFFFu;
    if (f > -2147483648.0f) return (uint32_t)(int32_t)f;
    return f != f ? 0u : 0x80000000u;
}
/* double -> int32 truncation (cvttsd2si r32) */
H2Y_FN int32_t d2i_ref(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return (int32_t)0x80000000;
    return (int32_t)v;
}

/* (int)(d / c + 0.5) exactly as IEEE division would give it, without the
 * division: q = d * (1/c) differs from d / c by at most 1 ulp, which can only
 * change the truncated integer if q + 0.5 sits within a few ulps of an
 * integer; in that (never yet observed) case do the real division. */
H2Y_FN int32_t div_round_trunc(float df, double c, double inv_c)
{
    double d = (double)df;
    double q = d * inv_c + 0.5;
    double n = __builtin_rint(q);
    if (__builtin_fabs(q - n) < 0x1p-46 * (1.0 + __builtin_fabs(q))) q = d / c + 0.5;
    return d2i_ref(q);
}

/* matrix_convert() from scaled code values to clamped 4:4:4 integers,
 * convert.cpp:1150-1221.  G,B,R are the float code values after the scale
 * step (or the raw samples when convert_transfer == 0). */
H2Y_FN void pix_matrix(const pix_params &pp, float G, float B, float R, uint32_t &Yo, uint32_t &Cbo, uint32_t &Cro)
{
    uint32_t Y;
    int64_t Cb, Cr;
    if (pp.mode == H2Y_MODE_IDENTITY) {
        Y = f2u_ref(G);
        Cb = (int64_t)f2u_ref(B);
        Cr = (int64_t)f2u_ref(R);
    } else {
        if (pp.mode == H2Y_MODE_YCBCR) {
            double yd = ((pp.kr * (double)R + pp.kg * (double)G) + pp.kb * (double)B) + 0.5;
            float tmpF = (float)yd;
            Y = f2u_ref(tmpF);
            Cb = div_round_trunc(B - tmpF, pp.dcb, pp.inv_dcb);
            Cr = div_round_trunc(R - tmpF, pp.dcr, pp.inv_dcr);
        } else if (pp.mode == H2Y_MODE_YDZDX) {
            Y = f2u_ref(G);
            double hg = (double)(-G) / 2.0;
            Cb = d2i_ref((hg + (double)B / 2.0) + 0.5);
            Cr = d2i_ref((hg + (double)R / 2.0) + 0.5);
        } else {
            Y = f2u_ref(G);
            Cb = d2i_ref((double)(pp.P * G + pp.Q * B) + 0.5);
            Cr = d2i_ref((double)(pp.RR * R + pp.S * G) + 0.5);
        }
        Cb += (int64_t)pp.half_m1;
        Cr += (int64_t)pp.half_m1;
    }
    /* unsigned 64-bit compares: negatives wrap and clamp high */
    uint64_t y64 = Y, cb64 = (uint64_t)Cb, cr64 = (uint64_t)Cr;
    Yo = y64 > pp.maxCV ? pp.maxCV : (uint32_t)y64;
    Cbo = cb64 > pp.maxCV ? pp.maxCV : (uint32_t)cb64;
    Cro = cr64 > pp.maxCV ? pp.maxCV : (uint32_t)cr64;
}

/* scale step, convert.cpp:1123-1145: separate multiply and add in binary32 */
H2Y_FN float pix_scale(float v, float mul, float add, int scale_mode)
{
    float r = v * mul;
    if (scale_mode != H2Y_SCALE_FULL) r = r + add;
    return r;
}

/* write_yuv per-sample step, tiff.cpp:469-478 (luma) / 502-511,533-543 (chroma) */
H2Y_FN uint32_t pix_yuv_clamp(const pix_params &pp, uint32_t v, bool chroma)
{
    v = (v >> pp.down_shift) & 0xFFFFu;
    if (pp.full_range == 0) {
        uint32_t lo = chroma ? pp.clo : pp.ylo, hi = chroma ? pp.chi : pp.yhi;
        v = v < lo ? lo : v;
        v = v > hi ? hi : v;
    } else {
        v = v > pp.out_maxCV ? pp.out_maxCV : v;
    }
    return v;
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
    float t = acc + 0.5f;
    if (t > fmaxcv) t = fmaxcv;
    if (t < 0.0f) t = 0.0f;
    return (uint32_t)t;
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
    float t = acc + 0.5f;
    if (t > fmaxcv) t = fmaxcv;
    if (t < 0.0f) t = 0.0f;
    return (uint32_t)t;
}

} // namespace h2y
#endif /* H2Y_MATH_H */
