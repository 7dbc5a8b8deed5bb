/*
 * h2y_device.h -- device-side helpers shared by the kernel translation units (h2y_kernels.hip,
 * h2y_fir_fused.hip): wave reductions, pic_stats' min/max accumulators, global-address-space loads and
 * stores, the sample loaders of the three input types, the exact tiers of one pixel, table staging.
 * Device code only; included after <hip/hip_runtime.h>, h2y_math.h and h2y_kernels.h.
 */
#ifndef H2Y_DEVICE_H
#define H2Y_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "h2y_kernels.h"
#include "h2y_math.h"

using namespace h2y;

#define WAVE 64

/* ---- wave reductions (DPP via __shfl_xor on 64 lanes) ------------------- */
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}

/* pic_stats update, common.cpp:126-127: "s < min ? s : min" -- a NaN never
 * wins, which v_min/v_max reproduce (they return the non-NaN operand).
 * v_min3/v_max3 fold two new samples per instruction. */
__device__ __forceinline__ float min3f(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max3f(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
struct mm6 {
    float lo[3], hi[3];
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            lo[c] = 3.402823466e+38f; /* numeric_limits<float>::max(), common.cpp:118 */
            hi[c] = 1.175494351e-38f; /* numeric_limits<float>::min(), common.cpp:119 */
        }
    }
    __device__ __forceinline__ void add(int c, float s)
    {
        lo[c] = min3f(lo[c], s, s);
        hi[c] = max3f(hi[c], s, s);
    }
    __device__ __forceinline__ void add2(int c, float s, float t)
    {
        lo[c] = min3f(lo[c], s, t);
        hi[c] = max3f(hi[c], s, t);
    }
    __device__ __forceinline__ void add2_max(int c, float s, float t) { hi[c] = max3f(hi[c], s, t); }
    /* the eight samples of one tile and channel: their own min/max come back too */
    __device__ __forceinline__ void add8(int c, const float (&p)[4], const float (&q)[4], float &tlo, float &thi)
    {
        tlo = min3f(min3f(p[0], p[1], p[2]), min3f(p[3], q[0], q[1]), min3f(q[2], q[3], q[3]));
        thi = max3f(max3f(p[0], p[1], p[2]), max3f(p[3], q[0], q[1]), max3f(q[2], q[3], q[3]));
        lo[c] = min3f(lo[c], tlo, tlo);
        hi[c] = max3f(hi[c], thi, thi);
    }
};

/* Block-wide reduce of a thread's mm6 and one plain store of the six floats
 * by thread 0.  All threads of the block must call it. */
template <int NWAVES>
__device__ __forceinline__ void block_store_mm(mm6 &m, float *smem /* NWAVES*6 */, float *dst)
{
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float a = wave_min(m.lo[c]), b = wave_max(m.hi[c]);
        if (lane == 0) {
            smem[wave * 6 + 2 * c] = a;
            smem[wave * 6 + 2 * c + 1] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = smem[threadIdx.x];
        const bool is_max = threadIdx.x & 1;
        for (int w = 1; w < NWAVES; w++) {
            float o = smem[w * 6 + threadIdx.x];
            v = is_max ? fmaxf(v, o) : fminf(v, o);
        }
        dst[threadIdx.x] = v;
        smem[threadIdx.x] = v; /* (its own column's first entry, read above: after the barrier smem[0..5] is the block's result --
                                  dst may be host memory, not to be read back) */
    }
    __syncthreads();
}

/* Same, but per wave and with no barrier: the fused kernels' waves drift apart across
 * frames instead of draining the CU's memory pipeline at every frame boundary.
 * dst points at this wave's six floats; k_stats_final folds grid * waves entries. */
__device__ __forceinline__ void wave_store_mm(mm6 &m, float *dst)
{
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float a = wave_min(m.lo[c]), b = wave_max(m.hi[c]);
        if (lane == 0) {
            dst[2 * c] = a;
            dst[2 * c + 1] = b;
        }
    }
}

/* ---- sample loads ------------------------------------------------------- */
/* Picture planes are addressed as GLOBAL memory, not through flat pointers: a flat load counts on
 * the LDS counter as well (every table read would wait for the prefetched tile), and a global
 * load takes a uniform base in scalar registers plus a 32-bit lane offset.  The plane pointers of
 * a frame come out of memory, so the compiler cannot know either fact by itself. */
#define H2Y_GLOBAL __attribute__((address_space(1)))
/* index counts T's; with a 32-bit index the BYTE offset is formed in 32 bits too (pictures are
 * limited to 2^28 samples), which is what lets the lane offset stay one register */
template <typename T> __device__ __forceinline__ T gload(const void *base, uint32_t index)
{
    return *(const H2Y_GLOBAL T *)((const H2Y_GLOBAL char *)base + index * (uint32_t)sizeof(T));
}
template <typename T> __device__ __forceinline__ T gload(const void *base, size_t index)
{
    return *((const H2Y_GLOBAL T *)base + index);
}
template <typename T> __device__ __forceinline__ void gstore(void *base, uint32_t index, T v)
{
    *(H2Y_GLOBAL T *)((H2Y_GLOBAL char *)base + index * (uint32_t)sizeof(T)) = v;
}
/* Picture samples are read once and output bytes written once per launch: non-temporal ("nt") accesses
 * keep them from displacing each other in the caches (-1.5 % on C2, both together; either alone: nothing).
 * Not for the 4:4:4 chroma scratch of the FIR path, which k_fir420 reads back. */
#ifndef H2Y_NT
#define H2Y_NT 3 /* 1: picture loads, 2: output stores, 3: both, 0: neither (A/B timing) */
#endif
template <typename T> __device__ __forceinline__ T gload_nt(const void *base, uint32_t index)
{
#if H2Y_NT & 1
    return __builtin_nontemporal_load((const H2Y_GLOBAL T *)((const H2Y_GLOBAL char *)base + index * (uint32_t)sizeof(T)));
#else
    return gload<T>(base, index);
#endif
}
template <typename T> __device__ __forceinline__ void gstore_nt(void *base, uint32_t index, T v)
{
#if H2Y_NT & 2
    __builtin_nontemporal_store(v, (H2Y_GLOBAL T *)((H2Y_GLOBAL char *)base + index * (uint32_t)sizeof(T)));
#else
    gstore<T>(base, index, v);
#endif
}
/* a pointer every lane holds the same value of, moved to scalar registers */
template <typename P> __device__ __forceinline__ P *uniform_ptr(P *p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<P *>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ frame_io uniform_io(const frame_io *p)
{
    frame_io io = *p;
#pragma unroll
    for (int c = 0; c < 3; c++) io.in[c] = uniform_ptr(io.in[c]);
    io.out = uniform_ptr(io.out);
    io.tmp_cb = uniform_ptr(io.tmp_cb);
    io.tmp_cr = uniform_ptr(io.tmp_cr);
    return io;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int KIND> struct in_traits;
template <> struct in_traits<H2Y_IN_F32> {
    typedef float T;
    /* i: sample index, a multiple of 4 (16-byte aligned planes) */
    template <typename I> static __device__ __forceinline__ void load4(const void *p, I i, float v[4]) { load4q(p, i >> 2, v); }
    /* q: quad index (sample index / 4) */
    template <typename I> static __device__ __forceinline__ void load4q(const void *p, I q4, float v[4])
    {
        const f32x4 q = gload_nt<f32x4>(p, (uint32_t)q4);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
    template <typename I> static __device__ __forceinline__ float load1(const void *p, I i) { return gload<float>(p, i); }
};
template <> struct in_traits<H2Y_IN_F16> {
    typedef _Float16 T;
    template <typename I> static __device__ __forceinline__ void load4(const void *p, I i, float v[4]) { load4q(p, i >> 2, v); }
    template <typename I> static __device__ __forceinline__ void load4q(const void *p, I q4, float v[4])
    {
        /* exr.cpp:233-235: half widened to float, exact */
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 q = gload_nt<h4>(p, (uint32_t)q4);
        v[0] = (float)q.x; v[1] = (float)q.y; v[2] = (float)q.z; v[3] = (float)q.w;
    }
    template <typename I> static __device__ __forceinline__ float load1(const void *p, I i) { return (float)gload<_Float16>(p, i); }
};
template <> struct in_traits<H2Y_IN_U16> {
    typedef uint16_t T;
    template <typename I> static __device__ __forceinline__ void load4(const void *p, I i, float v[4]) { load4q(p, i >> 2, v); }
    template <typename I> static __device__ __forceinline__ void load4q(const void *p, I q4, float v[4])
    {
        /* convert.cpp:989-994: (float) of the unsigned short */
        const u32x2 q = gload_nt<u32x2>(p, (uint32_t)q4);
        v[0] = (float)(q.x & 0xFFFFu); v[1] = (float)(q.x >> 16);
        v[2] = (float)(q.y & 0xFFFFu); v[3] = (float)(q.y >> 16);
    }
    template <typename I> static __device__ __forceinline__ float load1(const void *p, I i) { return (float)gload<uint16_t>(p, i); }
};

/* ---- one pixel: normalise -> PQ -> scale -> matrix ---------------------- */

/* Careful tier for one pixel: the reference's operations one by one (double-
 * double pow in place of libm's, IEEE divisions).  Out of line: about one
 * pixel in 10^4 comes here. */
struct ycc {
    uint32_t y, cb, cr;
};
template <int MODE>
__device__ __attribute__((noinline)) ycc pixel_careful(const pix_params *spp /* block's copy in LDS */, float G, float B, float R)
{
    const pix_params &pp = *spp;
    ycc o;
    uint32_t &Y = o.y, &Cb = o.cb, &Cr = o.cr;
    if (pp.convert_transfer) {
        /* convert.cpp:1024-1109: source transfer -> linear -> destination transfer, then the scale step */
        G = pix_scale(tf_from_linear(pp.dst_tf, tf_to_linear(pp.src_tf, G), pp.pq_ext), pp.mulY, pp.addY);
        B = pix_scale(tf_from_linear(pp.dst_tf, tf_to_linear(pp.src_tf, B), pp.pq_ext), pp.mulC, pp.addC);
        R = pix_scale(tf_from_linear(pp.dst_tf, tf_to_linear(pp.src_tf, R), pp.pq_ext), pp.mulC, pp.addC);
    }
    bool dummy;
    pix_matrix<MODE, true>(pp, G, B, R, Y, Cb, Cr, &dummy);
    return o;
}

/* PQ10000_r of the samples the LDS tables do not reach -- normal floats below 2^-24: dark pictures that never went through
 * half floats hold them (a frame of squared uniform noise: 0.024 % of its samples) --, answered IN LINE from the table over
 * every normal float below 2 in global memory (pq_build_table_ext, 254 KB, L2-resident), one flagged lane at a time: the
 * lane's sample is read into a scalar register, its two 16-byte records fetched by SCALAR loads and the polynomial run on
 * them.  Scalar loads count on lgkmcnt: the wave does not wait for its outstanding picture loads (vmcnt), which is what made
 * the out-of-line careful tier cost ~5 us per such pixel inside the loops with a rolling prefetch (k_fir_fused on that
 * frame: 2.39 ms against 1.76).  `slow` in: the LDS tier's verdict; out: still unsettled (ambiguous rounding, subnormal,
 * negative, NaN, >= 2: the careful tier as before). */
typedef const __attribute__((address_space(4))) uint32_t *h2y_const_u32;
__device__ __forceinline__ float pq_ext_inline(float x, float v, bool &slow, const void *ext, uint32_t lo_bits = (uint32_t)(127 + H2Y_PQ_EMIN) << 23)
{
    const uint32_t xb = f2bits(x);
    /* candidates: flagged, and 2^-126 <= x < the LDS table's lowest float, 2^-24 for PQ10000_r (in the LDS table's own
     * domain a flag means an ambiguous rounding, which the same polynomial would only repeat) */
    const bool cand = slow && (xb - 0x00800000u) < (lo_bits - 0x00800000u);
    uint64_t m = __builtin_amdgcn_ballot_w64(cand);
    if (__builtin_expect(m != 0 && ext != nullptr, 0)) {
        const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const uint64_t base = reinterpret_cast<uint64_t>(ext);
        do {
            const uint32_t l = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            const uint32_t bits = __builtin_amdgcn_readlane(xb, l);
            const uint32_t idx = (bits >> H2Y_PQ_LOW_BITS) - H2Y_PQX_SEG_BASE;
            const h2y_const_u32 pr = reinterpret_cast<h2y_const_u32>(base + (uint64_t)idx * 32u); /* pq_ext_rec: one s_load_dwordx8 */
            const uint32_t w0 = pr[0], w1 = pr[1], w2 = pr[2], w3 = pr[3], w4 = pr[4], w5 = pr[5], w6 = pr[6], w7 = pr[7];
            pq_recA a;
            pq_recB b;
            a.c0 = bits2d((uint64_t)w0 | ((uint64_t)w1 << 32));
            a.c1 = bits2d((uint64_t)w2 | ((uint64_t)w3 << 32));
            b.c2 = bits2d((uint64_t)w4 | ((uint64_t)w5 << 32));
            b.c3 = bits2f(w6);
            b.c4 = bits2f(w7);
            const double d = pq_poly(bits, a, b);
            const bool mine = lane == l;
            v = mine ? (float)d : v;
            slow = mine ? pq_ambiguous(d) : slow;
        } while (m);
    }
    return v;
}

/* The same for code that has no picture loads in flight (k_fused_t1's redo passes, k_fused): every candidate lane reads ITS record
 * with two 16-byte vector loads -- one latency for the wave however many lanes hold such a sample (a redo pass over a dark
 * picture has several per pixel position, which the scalar form would take one after the other). */
__device__ __forceinline__ float pq_ext_gather(float x, float v, bool &slow, const void *ext, uint32_t lo_bits = (uint32_t)(127 + H2Y_PQ_EMIN) << 23)
{
    const uint32_t xb = f2bits(x);
    const bool cand = slow && (xb - 0x00800000u) < (lo_bits - 0x00800000u);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(cand) != 0 && ext != nullptr, 0)) {
        const uint32_t idx = cand ? (xb >> H2Y_PQ_LOW_BITS) - H2Y_PQX_SEG_BASE : 0u;
        const pq_ext_rec *X = reinterpret_cast<const pq_ext_rec *>(ext);
        const pq_recA a = X[idx].a;
        const pq_recB b = X[idx].b;
        const double d = pq_poly(xb, a, b);
        v = cand ? (float)d : v;
        slow = cand ? pq_ambiguous(d) : slow;
    }
    return v;
}

/* one stage of a transfer pair by the careful tier (h2y_math.h), out of line: what the table tier of the generic pairs
 * falls back to for a single sample */
static __device__ __attribute__((noinline)) float tf_to_linear_careful(int cls, float v) { return tf_to_linear(cls, v); }
static __device__ __attribute__((noinline)) float tf_from_linear_careful(int cls, float v) { return tf_from_linear(cls, v); }

/* PIPE: what is known at compile time about the front of the pixel pipeline */
#define H2Y_PIPE_RUNTIME 0  /* read pp.convert_transfer / pp.norm_identity */
#define H2Y_PIPE_PQ_IDENT 1 /* LINEAR -> PQ, floor 0 / ceiling 1: no normalisation arithmetic */
#define H2Y_PIPE_PQ_NORM 2  /* LINEAR -> PQ with (x - offset) / range */
#define H2Y_PIPE_NONE 6     /* equal transfers: the samples go to the matrix as they are (k_fused2 only) */
#define H2Y_PIPE_TFN 7      /* any other transfer pair through its two stages' tables (pp.src_fn / pp.dst_fn), loop form (k_fused2 only) */

/* normalisation of one sample, convert.cpp:1017-1019: binary32 subtract, IEEE divide */
template <int PIPE> __device__ __forceinline__ float norm1(const pix_params &pp, int c, float v)
{
    const bool conv = PIPE == H2Y_PIPE_RUNTIME ? pp.convert_transfer != 0 : true;
    const bool ident = PIPE == H2Y_PIPE_RUNTIME ? pp.norm_identity : PIPE == H2Y_PIPE_PQ_IDENT;
    return (conv && !ident) ? (v - pp.offset[c]) / pp.range[c] : v;
}

/* fast tier of one pixel (inputs normalised).  Returns true when the result
 * cannot be trusted and pixel_careful() must be used instead. */
template <int MODE, int PIPE>
__device__ __forceinline__ bool pixel_fast(const pix_params &pp, const pq_recA *sA, const pq_recB *sB, float G, float B, float R,
                                           uint32_t &Y, uint32_t &Cb, uint32_t &Cr)
{
    const bool conv = PIPE == H2Y_PIPE_RUNTIME ? pp.convert_transfer != 0 : true;
    float g = G, b = B, r = R;
    bool unsure = false;
    if (PIPE == H2Y_PIPE_RUNTIME && pp.convert_transfer == 2) {
        /* any other transfer pair (convert.cpp:1024-1109): source function -> linear light (a float, as in the reference)
         * -> destination function, each through its own table (tfn_fast, h2y_math.h): the source's in the first table
         * slot of the block's LDS image, the destination's in the second */
        if (pp.src_fn < 0) {
            Y = Cb = Cr = 0;
            return true; /* no tables for this launch: careful tier */
        }
        const pq_recA *t_src = sA, *t_dst = sA + 2 * H2Y_PQ_NREC;
        float v[3] = {G, B, R};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float x = v[c];
            if (pp.src_fn) {
                bool s1;
                const float x0 = x;
                if (pp.src_fn == H2Y_TFN_RHO_H) x = (powf25(x) - 1.0f) * 0.0625f; /* RHO_GAMMA_f's inner powf, then (P - 1) / 16: both exact */
                const float xin = x;
                x = tfn_fast(x, t_src, tfn_cut_of(pp.src_fn), tfn_zero_bits(pp.src_fn), tfn_one_bits(pp.src_fn), &s1);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(s1) != 0, 0)) {
                    x = pq_ext_gather(xin, x, s1, pp.tf_ext[0], tfn_lo_bits(pp.src_fn)); /* below the LDS table: the function's full-range table */
                    if (s1) x = tf_to_linear_careful(pp.src_tf, x0); /* this sample alone, not its pixel */
                }
            }
            if (pp.dst_fn) {
                bool s2;
                const float x1 = x;
                x = tfn_fast(x, t_dst, tfn_cut_of(pp.dst_fn), tfn_zero_bits(pp.dst_fn), tfn_one_bits(pp.dst_fn), &s2);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(s2) != 0, 0)) {
                    x = pq_ext_gather(x1, x, s2, pp.tf_ext[1], tfn_lo_bits(pp.dst_fn));
                    if (s2) x = tf_from_linear_careful(pp.dst_tf, x1);
                }
            }
            unsure |= !(x == x); /* a NaN (negative sample through a power, ...) takes its pixel to the careful matrix: the reference's x86 conversions */
            v[c] = x;
        }
        g = pix_scale(v[0], pp.mulY, pp.addY);
        b = pix_scale(v[1], pp.mulC, pp.addC);
        r = pix_scale(v[2], pp.mulC, pp.addC);
    } else if (conv) {
        bool sg, sb, sr;
        float vg = pq_fast(G, sA, sB, &sg), vb = pq_fast(B, sA, sB, &sb), vr = pq_fast(R, sA, sB, &sr);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(sg | sb | sr) != 0, 0)) { /* samples below the LDS table: in line, from the full-range table */
            vg = pq_ext_gather(G, vg, sg, pp.pq_ext);
            vb = pq_ext_gather(B, vb, sb, pp.pq_ext);
            vr = pq_ext_gather(R, vr, sr, pp.pq_ext);
        }
        g = pix_scale(vg, pp.mulY, pp.addY);
        b = pix_scale(vb, pp.mulC, pp.addC);
        r = pix_scale(vr, pp.mulC, pp.addC);
        unsure = sg | sb | sr;
    }
    bool um;
    pix_matrix<MODE, false>(pp, g, b, r, Y, Cb, Cr, &um);
    return unsure | um;
}

/* one pixel, both tiers (used by the narrow-width kernel) */
template <int MODE>
__device__ __forceinline__ void pixel(const pix_params &pp, const pix_params *spp, const pq_recA *sA, const pq_recB *sB,
                                      float G, float B, float R, uint32_t &Y, uint32_t &Cb, uint32_t &Cr)
{
    G = norm1<H2Y_PIPE_RUNTIME>(pp, 0, G);
    B = norm1<H2Y_PIPE_RUNTIME>(pp, 1, B);
    R = norm1<H2Y_PIPE_RUNTIME>(pp, 2, R);
    if (__builtin_expect(pixel_fast<MODE, H2Y_PIPE_RUNTIME>(pp, sA, sB, G, B, R, Y, Cb, Cr), 0)) {
        const ycc o = pixel_careful<MODE>(spp, G, B, R);
        Y = o.y;
        Cb = o.cb;
        Cr = o.cr;
    }
}

/* convert.cpp:939-940: range = ceiling - floor (int), offset = floor, as float */
__device__ __forceinline__ pix_params with_assumed(const pix_params &in, const assumed_stats *as)
{
    pix_params pp = in;
    bool ident = true;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int fl = as->floor_[c], ce = as->ceil_[c];
        pp.offset[c] = (float)fl;
        pp.range[c] = (float)(ce - fl);
        ident = ident && fl == 0 && (ce - fl) == 1;
    }
    pp.norm_identity = ident ? 1 : 0;
    return pp;
}

/* unsigned divide by a launch constant: q = n / d via a 32-bit reciprocal */
__device__ __forceinline__ uint32_t udiv_magic(uint32_t n, uint32_t d, uint32_t magic, uint32_t &rem)
{
    uint32_t q = __umulhi(n, magic);
    uint32_t r = n - q * d;
    if (r >= d) {
        q++;
        r -= d;
    }
    rem = r;
    return q;
}

/* N 16-byte records from global memory to LDS: every load of a thread is issued before the first is
 * stored (as a plain loop the copy is one memory round trip per iteration -- ten of them, 10-15 us, at the
 * head of every launch of the first-tier kernel) */
template <int THREADS, int N> __device__ __forceinline__ void stage16(const void *src, void *dst)
{
    constexpr int IT = (N + THREADS - 1) / THREADS;
    const uint4 *g = reinterpret_cast<const uint4 *>(src);
    uint4 *l = reinterpret_cast<uint4 *>(dst);
    uint4 r[IT];
#pragma unroll
    for (int j = 0; j < IT; j++) {
        const int i = threadIdx.x + j * THREADS;
        if (j < IT - 1 || i < N) r[j] = g[i];
    }
#pragma unroll
    for (int j = 0; j < IT; j++) {
        const int i = threadIdx.x + j * THREADS;
        if (j < IT - 1 || i < N) l[i] = r[j];
    }
}
template <int THREADS> __device__ __forceinline__ void stage_table(const void *table, pq_recA *s_tab)
{
    stage16<THREADS, 2 * H2Y_PQ_NREC>(table, s_tab);
}

/* The half-input kernels' table in LDS: PQ10000_r of every half in [0, 2) (a.lut16) taken through the scale step as well --
 * convert.cpp:1123-1145, one binary32 multiply and one add, the same for every pixel of a launch -- once with the luma
 * constants (the G / Y plane), once with the chroma ones (B and R): 2 x 64 KB instead of 64 KB and six instructions a pixel. */
template <int THREADS> __device__ __forceinline__ void stage_lut16_scaled(const float *lut, const pix_params &pp, float *s_y, float *s_c)
{
    const f32x4 *g = reinterpret_cast<const f32x4 *>(lut);
    f32x4 *ly = reinterpret_cast<f32x4 *>(s_y), *lc = reinterpret_cast<f32x4 *>(s_c);
    for (int i = threadIdx.x; i < H2Y_LUT16_N / 4; i += THREADS) {
        const f32x4 q = g[i];
        ly[i] = f32x4{pix_scale(q.x, pp.mulY, pp.addY), pix_scale(q.y, pp.mulY, pp.addY), pix_scale(q.z, pp.mulY, pp.addY), pix_scale(q.w, pp.mulY, pp.addY)};
        lc[i] = f32x4{pix_scale(q.x, pp.mulC, pp.addC), pix_scale(q.y, pp.mulC, pp.addC), pix_scale(q.z, pp.mulC, pp.addC), pix_scale(q.w, pp.mulC, pp.addC)};
    }
}

/* ---- matrix_inverse (convert.cpp:1320-1867), one pixel: see k_inverse (h2y_kernels.hip) for the behaviour kept ----
 * BT.709 green: the reference divides by 0.7152 in binary64.  Here: one fused multiply-add with the reciprocal, and the IEEE
 * division only when the result's rounding to float could differ (its low 29 bits within 4096 ulp(double) of the tie:
 * the reciprocal form is off by an ulp or two; 1.5e-5 of the pixels). */
__device__ __forceinline__ void inverse_pixel(const inverse_args &a, uint32_t y, uint32_t cb, uint32_t cr, uint32_t &G, uint32_t &B, uint32_t &R)
{
    float Yav = (float)y;
    const float Cb = (float)cb, Cr = (float)cr;
    float Rp, Bp;
    if (!a.d709) {
        Rp = (float)((2.0 * (double)Cr - 4095.0) + (double)Yav);
        Bp = (float)((2.0 * (double)Cb - 4095.0) + (double)Yav);
    } else {
        float t = (float)(((double)Cb - 2047.5) * 1.8556 + (double)Yav);
        if (t > 4095.0f) t = 4095.0f;
        Bp = t;
        t = (float)(((double)Cr - 2047.5) * 1.5748 + (double)Yav);
        if (t > 4095.0f) t = 4095.0f;
        Rp = t;
        const double num = ((double)Yav - 0.07222 * (double)Bp) - 0.2126 * (double)Rp;
        double q = __builtin_fma(num, 1.0 / 0.7152, 0.5);
        if (__builtin_expect(pq_ambiguous(q), 0)) q = num / 0.7152 + 0.5;
        t = (float)q;
        if (t > 4095.0f) t = 4095.0f;
        Yav = t;
    }
    int g = sat_i32_f32(Yav), b = sat_i32_f32(Bp), r = sat_i32_f32(Rp);
    /* negative -> 0, then the clamp to [minVR, maxVR] (minVR >= 0): one clamp does both */
    g = min(max(g, (int)a.minVR), (int)a.maxVR);
    b = min(max(b, (int)a.minVR), (int)a.maxVR);
    r = min(max(r, (int)a.minVR), (int)a.maxVR);
    if (a.shift_right) { g >>= a.shift; b >>= a.shift; r >>= a.shift; }
    else { g <<= a.shift; b <<= a.shift; r <<= a.shift; }
    G = (uint32_t)g & 0xFFFFu; /* stored into an unsigned short */
    B = (uint32_t)b & 0xFFFFu;
    R = (uint32_t)r & 0xFFFFu;
}

#endif /* H2Y_DEVICE_H */
