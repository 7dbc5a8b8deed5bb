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
