// tools/ubench.hip -- instruction-throughput microbenchmarks on gfx950 (GPU box).
// hipcc --offload-arch=gfx950 -O2 tools/ubench.hip -o build/ubench && build/ubench
// For each op: one workgroup per CU with WAVES_PER_SIMD*4 waves, 8 independent
// chains per lane, N iterations; reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define ITER 512
#define CHAINS 8

template <int OP> __device__ __forceinline__ void body(double (&d)[CHAINS], float (&f)[CHAINS], int (&i)[CHAINS], const char* lds)
{
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %1" : "+v"(f[c]) : "v"(f[(c + 1) % CHAINS]));
        if (OP == 1) asm volatile("v_fma_f64 %0, %0, %0, %1" : "+v"(d[c]) : "v"(d[(c + 1) % CHAINS]));
        if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[(c + 1) % CHAINS]));
        if (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[(c + 1) % CHAINS]));
        if (OP == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[c]) : "v"(f[c]));
        if (OP == 5) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[c]) : "v"(d[c]));
        if (OP == 6) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i[c]) : "v"(d[c]));
        if (OP == 7) asm volatile("v_rndne_f64 %0, %1" : "=v"(d[c]) : "v"(d[(c + 1) % CHAINS]));
        if (OP == 8) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[c]) : "v"(i[c]));
        if (OP == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i[c]) : "v"(i[(c + 1) % CHAINS]));
        if (OP == 10) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[c]) : "v"(i[c]), "v"(i[(c + 1) % CHAINS]) : "vcc");
        if (OP == 11) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(i[c]) : "v"(f[c]));
        if (OP == 12) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[c]), "v"(d[(c + 1) % CHAINS]) : "vcc");
        if (OP == 13) asm volatile("v_fract_f64 %0, %1" : "=v"(d[c]) : "v"(d[(c + 1) % CHAINS]));
        if (OP == 14) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[c]) : "v"(f[(c + 1) % CHAINS]));
        if (OP == 15) asm volatile("v_pk_fma_f32 %0, %0, %0, %1" : "+v"(d[c]) : "v"(d[(c + 1) % CHAINS]));
        if (OP == 16) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[c]) : "v"(i[c]));
        if (OP == 17) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(f[(c + 1) % CHAINS]), "v"(f[(c + 2) % CHAINS]));
        if (OP == 18) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i[c]) : "v"(i[(c + 1) % CHAINS]) : "vcc");
        if (OP == 19) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[c]) : "v"(i[c]));
        if (OP == 20) asm volatile("v_fma_f64 %0, %0, %1, 0.5" : "+v"(d[c]) : "v"(d[(c + 1) % CHAINS]));
        if (OP == 21) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[c]) : "v"(f[(c + 1) % CHAINS]), "v"(f[(c + 2) % CHAINS]), "v"(f[(c + 3) % CHAINS]));
        if (OP == 22) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[c]) : "v"(f[(c + 1) % CHAINS]), "v"(f[(c + 2) % CHAINS]));
        if (OP == 23) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(i[c]) : "v"(i[(c + 1) % CHAINS]), "v"(i[(c + 2) % CHAINS]), "v"(i[(c + 3) % CHAINS]));
        if (OP == 24) asm volatile("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(i[c]) : "v"(i[(c + 1) % CHAINS]), "v"(i[(c + 2) % CHAINS]));
        if (OP == 25) asm volatile("v_med3_f32 %0, %1, %2, 2.0" : "=v"(f[c]) : "v"(f[(c + 1) % CHAINS]), "v"(f[(c + 2) % CHAINS]));
        if (OP == 26) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(f[(c + 1) % CHAINS]));
        if (OP == 27) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(f[c]), "v"(f[(c + 1) % CHAINS]) : "s20", "s21");
        if (OP == 28) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(f[c]), "v"(f[(c + 1) % CHAINS]) : "vcc");
        if (OP == 29) asm volatile("v_lshrrev_b32 %0, 15, %1" : "=v"(i[c]) : "v"(i[(c + 1) % CHAINS]));
        if (OP == 30) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(d[c]) : "v"(d[(c + 1) % CHAINS]), "v"(d[(c + 2) % CHAINS]));
        if (OP == 31) asm volatile("v_and_or_b32 %0, %1, s20, 1.0" : "=v"(i[c]) : "v"(i[(c + 1) % CHAINS]) : "s20");
        if (OP == 32) asm volatile("v_fmamk_f32 %0, %1, 0x33000000, %2" : "=v"(f[c]) : "v"(f[(c + 1) % CHAINS]), "v"(f[(c + 2) % CHAINS]));
        if (OP == 33) asm volatile("v_med3_u32 %0, %1, s20, %2" : "=v"(i[c]) : "v"(i[(c + 1) % CHAINS]), "v"(i[(c + 2) % CHAINS]) : "s20");
        if (OP == 34) asm volatile("v_min_u32 %0, %0, %1" : "+v"(i[c]) : "v"(i[(c + 1) % CHAINS]));
        if (OP == 35) asm volatile("v_fma_f32 %0, %1, s20, %2" : "=v"(f[c]) : "v"(f[(c + 1) % CHAINS]), "v"(f[(c + 2) % CHAINS]) : "s20");
        if (OP == 36) asm volatile("v_mul_f64 %0, s[20:21], %1" : "=v"(d[c]) : "v"(d[(c + 1) % CHAINS]) : "s20", "s21");
        if (OP == 37) asm volatile("v_add_f32 %0, s20, %1" : "=v"(f[c]) : "v"(f[(c + 1) % CHAINS]) : "s20");
        if (OP == 38) asm volatile("v_mul_f32 %0, s20, %1" : "=v"(f[c]) : "v"(f[(c + 1) % CHAINS]) : "s20");
        if (OP == 39) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[c]) : "v"(f[(c + 1) % CHAINS]));
        if (OP == 40) asm volatile("v_and_b32 %0, %0, %1" : "+v"(i[c]) : "v"(i[(c + 1) % CHAINS]));
        if (OP == 41) asm volatile("v_and_b32 %0, 0x7fff, %1" : "=v"(i[c]) : "v"(i[(c + 1) % CHAINS]));
        if (OP == 42) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(i[c]) : "v"(i[(c + 1) % CHAINS]));
        if (OP == 43) asm volatile("v_add_u32 %0, s20, %1" : "=v"(i[c]) : "v"(i[(c + 1) % CHAINS]) : "s20");
        if (OP == 44) asm volatile("v_cndmask_b32 %0, %1, %2, s[20:21]" : "=v"(i[c]) : "v"(i[(c + 1) % CHAINS]), "v"(i[(c + 2) % CHAINS]) : "s20", "s21");
        if (OP == 45) asm volatile("v_max_u32 %0, %0, %1" : "+v"(i[c]) : "v"(i[(c + 1) % CHAINS]));
        if (OP == 46) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[c]) : "v"(f[(c + 1) % CHAINS]));
        if (OP == 47) asm volatile("v_fract_f32 %0, %1" : "=v"(f[c]) : "v"(f[(c + 1) % CHAINS]));
        if (OP == 48) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(i[c]) : "v"(f[(c + 1) % CHAINS]));
        if (OP == 49) asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(i[c]) : "v"(i[(c + 1) % CHAINS]), "v"(i[(c + 2) % CHAINS]));
        if (OP == 50) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[c]) : "v"(f[(c + 1) % CHAINS]));
        if (OP == 51) asm volatile("v_or_b32 %0, 1.0, %1" : "=v"(i[c]) : "v"(i[(c + 1) % CHAINS]));
        if (OP == 52) asm volatile("v_add_f64 %0, %1, 0.5" : "=v"(d[c]) : "v"(d[(c + 1) % CHAINS]));
    }
}

template <int OP> __global__ void k(long long* out, float seed)
{
    __shared__ char lds[1024];
    double d[CHAINS]; float f[CHAINS]; int i[CHAINS];
    for (int c = 0; c < CHAINS; c++) { d[c] = 1.0 + seed * (c + threadIdx.x); f[c] = 1.0f + seed * c; i[c] = c + threadIdx.x; }
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) body<OP>(d, f, i, lds);
    __builtin_amdgcn_s_waitcnt(0);
    long long t1 = __builtin_amdgcn_s_memtime();
    double acc = 0; for (int c = 0; c < CHAINS; c++) acc += d[c] + f[c] + i[c];
    if (acc == 123.456) out[1000] = 1;
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 64 + threadIdx.x / 64] = t1 - t0;
}

// LDS read throughput: conflict-free vs random 16B/8B reads
template <int BYTES, int PATTERN> __global__ void klds(long long* out, const int* rnd)
{
    __shared__ uint4 tab[3200];
    for (int j = threadIdx.x; j < 3200; j += blockDim.x) tab[j] = make_uint4(j, j, j, j);
    __syncthreads();
    int idx[8];
    for (int c = 0; c < 8; c++) {
        int r = rnd[(threadIdx.x * 8 + c) & 4095];
        idx[c] = PATTERN == 0 ? (threadIdx.x % 64 + c * 64) : PATTERN == 1 ? (r % 1600) : (r % 64 + 1000);
    }
    unsigned acc = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            if (BYTES == 16) { uint4 v = tab[idx[c]]; acc += v.x ^ v.y ^ v.z ^ v.w; idx[c] = (idx[c] + (v.x & 0)) ; }
            else { uint2 v = *reinterpret_cast<uint2*>(&tab[idx[c]]); acc += v.x ^ v.y; }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 12345) out[1000] = 1;
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 64 + threadIdx.x / 64] = t1 - t0;
}

template <int OP> void run(const char* name, long long* dout, int ncu)
{
    for (int wps : {1, 2, 4}) {
        int threads = 256 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(ncu), dim3(threads), 0, 0, dout, 1e-9f);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k<OP>, dim3(ncu), dim3(threads), 0, 0, dout, 1e-9f);
        hipDeviceSynchronize();
        std::vector<long long> h(ncu * 64);
        hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
        double mx = 0; int nw = threads / 64;
        for (int b = 0; b < ncu; b++) for (int w = 0; w < nw; w++) mx = mx < h[b * 64 + w] ? h[b * 64 + w] : mx;
        // s_memtime ticks at 100 MHz on gfx9? report ticks and per-instr ticks; also wall-derived cycles
        double insts_per_simd = (double)ITER * CHAINS * wps; // wave-instructions issued on one SIMD
        printf("%-16s waves/SIMD %d : %10.0f ticks  -> %.3f ticks per wave-instr per SIMD\n", name, wps, mx, mx / insts_per_simd);
    }
}

template <int BYTES, int PATTERN> void runlds(const char* name, long long* dout, int* drnd, int ncu)
{
    for (int wps : {1, 2, 4}) {
        int threads = 256 * wps;
        hipLaunchKernelGGL((klds<BYTES, PATTERN>), dim3(ncu), dim3(threads), 0, 0, dout, drnd);
        hipDeviceSynchronize();
        hipLaunchKernelGGL((klds<BYTES, PATTERN>), dim3(ncu), dim3(threads), 0, 0, dout, drnd);
        hipDeviceSynchronize();
        std::vector<long long> h(ncu * 64);
        hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
        double mx = 0; int nw = threads / 64;
        for (int b = 0; b < ncu; b++) for (int w = 0; w < nw; w++) mx = mx < h[b * 64 + w] ? h[b * 64 + w] : mx;
        double insts_per_cu = (double)ITER * 8 * wps * 4;
        printf("%-16s waves/SIMD %d : %10.0f ticks  -> %.3f ticks per wave-instr per CU\n", name, wps, mx, mx / insts_per_cu);
    }
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.name, ncu, p.clockRate);
    long long* dout; hipMalloc(&dout, (ncu * 64 + 2048) * 8);
    int* drnd; hipMalloc(&drnd, 4096 * 4);
    std::vector<int> r(4096); srand(1); for (auto& x : r) x = rand();
    hipMemcpy(drnd, r.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<26>("v_add_f32", dout, ncu);
    run<34>("v_min_u32", dout, ncu);
    run<29>("v_lshrrev_b32", dout, ncu);
    run<21>("v_fma_f32 3src", dout, ncu);
    run<35>("v_fma_f32 v,s,v", dout, ncu);
    run<22>("v_fmac_f32", dout, ncu);
    run<32>("v_fmamk_f32", dout, ncu);
    run<23>("v_and_or 3src", dout, ncu);
    run<31>("v_and_or v,s,c", dout, ncu);
    run<24>("v_lshl_add_u32", dout, ncu);
    run<25>("v_med3_f32", dout, ncu);
    run<33>("v_med3_u32", dout, ncu);
    run<27>("v_cmp_f32 e64", dout, ncu);
    run<28>("v_cmp_f32 e32", dout, ncu);
    run<30>("v_pk_mov_b32", dout, ncu);
    run<36>("v_mul_f64 s,v", dout, ncu);
    run<37>("v_add_f32 s,v", dout, ncu);
    run<38>("v_mul_f32 s,v", dout, ncu);
    run<43>("v_add_u32 s,v", dout, ncu);
    run<39>("v_max_f32", dout, ncu);
    run<50>("v_min_f32", dout, ncu);
    run<45>("v_max_u32", dout, ncu);
    run<40>("v_and_b32", dout, ncu);
    run<41>("v_and_b32 lit", dout, ncu);
    run<51>("v_or_b32 1.0", dout, ncu);
    run<42>("v_sub_u32", dout, ncu);
    run<46>("v_sub_f32", dout, ncu);
    run<44>("v_cndmask sgpr", dout, ncu);
    run<47>("v_fract_f32", dout, ncu);
    run<48>("v_cvt_i32_f32", dout, ncu);
    run<49>("v_lshl_or_b32", dout, ncu);
    run<52>("v_add_f64 0.5", dout, ncu);
    run<0>("v_fma_f32", dout, ncu);
    run<14>("v_mul_f32", dout, ncu);
    run<15>("v_pk_fma_f32", dout, ncu);
    run<9>("v_add_u32", dout, ncu);
    run<18>("v_cndmask_b32", dout, ncu);
    run<17>("v_min3_f32", dout, ncu);
    run<11>("v_cvt_u32_f32", dout, ncu);
    run<16>("v_cvt_f32_u32", dout, ncu);
    run<1>("v_fma_f64", dout, ncu);
    run<20>("v_fma_f64 lit", dout, ncu);
    run<2>("v_mul_f64", dout, ncu);
    run<3>("v_add_f64", dout, ncu);
    run<4>("v_cvt_f64_f32", dout, ncu);
    run<5>("v_cvt_f32_f64", dout, ncu);
    run<6>("v_cvt_i32_f64", dout, ncu);
    run<8>("v_cvt_f64_i32", dout, ncu);
    run<7>("v_rndne_f64", dout, ncu);
    run<13>("v_fract_f64", dout, ncu);
    run<12>("v_cmp_lt_f64", dout, ncu);
    run<19>("v_ldexp_f64", dout, ncu);
    run<10>("v_mad_u64_u32", dout, ncu);
    runlds<16, 0>("ds_b128 linear", dout, drnd, ncu);
    runlds<16, 1>("ds_b128 random", dout, drnd, ncu);
    runlds<16, 2>("ds_b128 rnd64", dout, drnd, ncu);
    runlds<8, 0>("ds_b64 linear", dout, drnd, ncu);
    runlds<8, 1>("ds_b64 random", dout, drnd, ncu);
    return 0;
}
