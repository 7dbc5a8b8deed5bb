// tools/membench.hip -- what does the MEMORY side of the box path allow?  Streaming kernels with k_fused_t1's access shape
// (three fp32 planes read 16 B per lane and row, 2-byte 4:2:0 output: luma 8 B per lane and row, chroma 4 B per lane and
// plane) and no arithmetic, in several forms: occupancy, loads in flight per wave, tile width, walk order.
// Timing experiment only; nothing in the library depends on it.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o build/membench && build/membench [frames]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

struct args {
    const float *in[3];
    uint16_t *out; /* Y plane, then Cb, Cr quarter planes, of ONE tall picture (frames stacked) */
    uint32_t W, H;  /* H = frames * 2160 */
    uint32_t lds_bytes_used; /* (only to keep the dynamic LDS alive) */
};

__device__ __forceinline__ uint32_t pack(f4 a) { return (__float_as_uint(a.x) >> 20) | ((__float_as_uint(a.y) >> 20) << 16); }
__device__ __forceinline__ uint32_t pack2(f4 a) { return (__float_as_uint(a.z) >> 20) | ((__float_as_uint(a.w) >> 20) << 16); }

/* one tile = COLS columns x 2 rows; tile tt: row pair tt / (W / COLS), column group tt % (W / COLS) */
template <int COLS> struct tile {
    f4 v[3][2][COLS / 4];
};
template <int COLS, bool NT>
__device__ __forceinline__ void load_tile(const args &a, uint32_t tt, tile<COLS> &t)
{
    const uint32_t wq = a.W / COLS, rp = tt / wq, cg = tt - rp * wq;
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int k = 0; k < COLS / 4; k++) {
                const f4 *src = reinterpret_cast<const f4 *>(a.in[p] + (size_t)(2 * rp + r) * a.W + cg * COLS) + k;
                t.v[p][r][k] = NT ? __builtin_nontemporal_load(src) : *src;
            }
}
template <int COLS, bool NT>
__device__ __forceinline__ void store_tile(const args &a, uint32_t tt, const tile<COLS> &t)
{
    const uint32_t wq = a.W / COLS, rp = tt / wq, cg = tt - rp * wq;
    const size_t npix = (size_t)a.W * a.H;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        uint16_t *y = a.out + (size_t)(2 * rp + r) * a.W + cg * COLS;
        if (COLS == 4) {
            u2 o = {pack(t.v[0][r][0]), pack2(t.v[0][r][0])};
            if (NT) __builtin_nontemporal_store(o, reinterpret_cast<u2 *>(y)); else *reinterpret_cast<u2 *>(y) = o;
        } else {
            u4 o = {pack(t.v[0][r][0]), pack2(t.v[0][r][0]), pack(t.v[0][r][1]), pack2(t.v[0][r][1])};
            if (NT) __builtin_nontemporal_store(o, reinterpret_cast<u4 *>(y)); else *reinterpret_cast<u4 *>(y) = o;
        }
    }
#pragma unroll
    for (int p = 1; p < 3; p++) {
        uint16_t *c = a.out + npix + (p - 1) * (npix / 4) + (size_t)rp * (a.W / 2) + cg * (COLS / 2);
        if (COLS == 4) {
            uint32_t o = pack(t.v[p][0][0]) + pack2(t.v[p][1][0]);
            if (NT) __builtin_nontemporal_store(o, reinterpret_cast<uint32_t *>(c)); else *reinterpret_cast<uint32_t *>(c) = o;
        } else {
            u2 o = {pack(t.v[p][0][0]) + pack2(t.v[p][1][0]), pack(t.v[p][0][1]) + pack2(t.v[p][1][1])};
            if (NT) __builtin_nontemporal_store(o, reinterpret_cast<u2 *>(c)); else *reinterpret_cast<u2 *>(c) = o;
        }
    }
}

/* form A: one tile per thread, as many blocks as tiles / 256 (the plain streaming kernel) */
template <int COLS, bool NT>
__global__ __launch_bounds__(256) void k_plain(args a)
{
    const uint32_t tt = blockIdx.x * 256u + threadIdx.x;
    if (tt >= (a.W / COLS) * (a.H / 2)) return;
    tile<COLS> t;
    load_tile<COLS, NT>(a, tt, t);
    store_tile<COLS, NT>(a, tt, t);
}

/* form B: persistent blocks of THREADS, DEPTH tiles in flight per lane; WALK 0: the card sweeps the picture together (wave
 * slices round-robin over all waves), WALK 1: every block owns a contiguous range */
template <int COLS, int THREADS, int DEPTH, int WALK, bool NT>
__global__ __launch_bounds__(THREADS) void k_persist(args a)
{
    extern __shared__ uint32_t lds[];
    if (a.lds_bytes_used == 1u) lds[threadIdx.x] = 1u; /* never */
    const uint32_t n_tiles = (a.W / COLS) * (a.H / 2), n_slices = (n_tiles + 63u) / 64u;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, wpb = THREADS / 64;
    uint32_t s, s_end, s_step;
    if (WALK == 0) {
        s = blockIdx.x * wpb + wave; s_end = n_slices; s_step = gridDim.x * wpb;
    } else {
        const uint32_t per = (n_slices + gridDim.x - 1) / gridDim.x;
        s = blockIdx.x * per + wave; s_end = min(n_slices, (blockIdx.x + 1) * per); s_step = wpb;
    }
    tile<COLS> t[DEPTH];
    uint32_t id[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        const uint32_t sd = s + d * s_step;
        id[d] = min(sd * 64u + lane, n_tiles - 1u);
        if (sd < s_end) load_tile<COLS, NT>(a, id[d], t[d]);
    }
    for (; s < s_end; s += DEPTH * s_step) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const uint32_t sd = s + d * s_step;
            if (sd >= s_end) break;
            store_tile<COLS, NT>(a, id[d], t[d]);
            const uint32_t sn = sd + DEPTH * s_step;
            id[d] = min(sn * 64u + lane, n_tiles - 1u);
            if (sn < s_end) load_tile<COLS, NT>(a, id[d], t[d]);
        }
    }
}

static args g_a;
static float time_it(const char *name, void (*launch)(), int reps, double bytes)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-46s %8.3f ms  %6.3f TB/s  %.3f of 8\n", name, ms, bytes / ms * 1e-9, bytes / ms * 1e-9 / 8.0);
    fflush(stdout);
    return ms;
}
template <int COLS, bool NT> static void l_plain()
{
    const uint32_t n = (g_a.W / COLS) * (g_a.H / 2);
    hipLaunchKernelGGL((k_plain<COLS, NT>), dim3((n + 255) / 256), dim3(256), 0, 0, g_a);
}
template <int COLS, int THREADS, int DEPTH, int WALK, bool NT, int LDS, int BPC> static void l_persist()
{
    static bool set = false;
    if (!set) { CHECK(hipFuncSetAttribute((const void *)k_persist<COLS, THREADS, DEPTH, WALK, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; }
    hipLaunchKernelGGL((k_persist<COLS, THREADS, DEPTH, WALK, NT>), dim3(256 * BPC), dim3(THREADS), LDS, 0, g_a);
}

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 64;
    g_a.W = 3840; g_a.H = 2160u * frames;
    const size_t npix = (size_t)g_a.W * g_a.H;
    for (int p = 0; p < 3; p++) {
        float *d;
        CHECK(hipMalloc(&d, npix * 4));
        CHECK(hipMemset(d, 0x3c + p, npix * 4));
        g_a.in[p] = d;
    }
    CHECK(hipMalloc(&g_a.out, npix * 3));
    CHECK(hipMemset(g_a.out, 0, npix * 3));
    const double bytes = (double)npix * 15.0;
    const int R = 10;
    printf("%d frames of 3840x2160: %.3f GB per launch\n", frames, bytes * 1e-9);
    time_it("plain 4 cols, 256 thr", l_plain<4, false>, R, bytes);
    time_it("plain 4 cols, 256 thr, nt", l_plain<4, true>, R, bytes);
    time_it("plain 8 cols, 256 thr", l_plain<8, false>, R, bytes);
    time_it("plain 8 cols, 256 thr, nt", l_plain<8, true>, R, bytes);
    /* the box kernel's shape: 1024 threads, one block per CU (150 KB of LDS), one tile in flight */
    time_it("persist 4c 1024thr depth1 sweep 150K", l_persist<4, 1024, 1, 0, true, 150 * 1024, 1>, R, bytes);
    time_it("persist 4c 1024thr depth1 ranges 150K", l_persist<4, 1024, 1, 1, true, 150 * 1024, 1>, R, bytes);
    time_it("persist 4c 1024thr depth2 sweep 150K", l_persist<4, 1024, 2, 0, true, 150 * 1024, 1>, R, bytes);
    time_it("persist 4c 1024thr depth2 ranges 150K", l_persist<4, 1024, 2, 1, true, 150 * 1024, 1>, R, bytes);
    time_it("persist 4c 1024thr depth3 sweep 150K", l_persist<4, 1024, 3, 0, true, 150 * 1024, 1>, R, bytes);
    time_it("persist 8c 1024thr depth1 sweep 150K", l_persist<8, 1024, 1, 0, true, 150 * 1024, 1>, R, bytes);
    time_it("persist 8c 1024thr depth1 ranges 150K", l_persist<8, 1024, 1, 1, true, 150 * 1024, 1>, R, bytes);
    time_it("persist 8c 1024thr depth2 sweep 150K", l_persist<8, 1024, 2, 0, true, 150 * 1024, 1>, R, bytes);
    time_it("persist 8c 512thr depth2 sweep 150K", l_persist<8, 512, 2, 0, true, 150 * 1024, 1>, R, bytes);
    time_it("persist 8c 512thr depth4 sweep 150K", l_persist<8, 512, 4, 0, true, 150 * 1024, 1>, R, bytes);
    time_it("persist 4c 512thr depth4 sweep 150K", l_persist<4, 512, 4, 0, true, 150 * 1024, 1>, R, bytes);
    time_it("persist 4c 1024thr depth1 sweep, 2 blocks/CU", l_persist<4, 1024, 1, 0, true, 1024, 2>, R, bytes);
    time_it("persist 4c 1024thr depth2 sweep, 2 blocks/CU", l_persist<4, 1024, 2, 0, true, 1024, 2>, R, bytes);
    time_it("persist 4c 1024thr depth1 sweep 150K plain ld/st", l_persist<4, 1024, 1, 0, false, 150 * 1024, 1>, R, bytes);
    return 0;
}
