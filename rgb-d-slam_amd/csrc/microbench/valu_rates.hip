// Microbenchmark: issue rate of the VALU instructions the cell-fit kernel is made of (gfx950).
// Prints cycles per wave-instruction per SIMD, assuming 256 CUs x 4 SIMDs at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;

template <int OP> __global__ void k(double* out, float seedf, double seedd)
{
    double d[UNROLL];
    float f[UNROLL];
    unsigned u[UNROLL];
    unsigned long long m[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { d[i] = seedd + i + threadIdx.x; f[i] = seedf + i + threadIdx.x; u[i] = 7u * i + threadIdx.x; }
    for (int it = 0; it < ITERS; ++it)
    {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i)
        {
            if (OP == 0) d[i] = d[i] + seedd;                       // v_add_f64
            if (OP == 1) d[i] = d[i] * seedd;                       // v_mul_f64
            if (OP == 2) d[i] = __builtin_fma(d[i], seedd, seedd);  // v_fma_f64
            if (OP == 3) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i])); }          // f32 -> f64
            if (OP == 4) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i])); }          // f64 -> f32
            if (OP == 5) f[i] = f[i] * seedf;                       // v_mul_f32
            if (OP == 6) f[i] = f[i] + seedf;                       // v_add_f32
            if (OP == 7) { asm volatile("v_max_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[i]), "v"(seedf)); }
            // round 2: which f32-class / integer instructions run at the doubled (SIMD-32) rate and which do not
            if (OP == 8) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f[i]) : "v"(f[i]), "v"(seedf)); }
            if (OP == 9) { asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(seedf) : "vcc"); }
            if (OP == 10) { asm volatile("v_addc_co_u32 %0, vcc, %1, 0, vcc" : "=v"(u[i]) : "v"(u[i]) : "vcc"); }
            if (OP == 11) { asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(f[i]) : "v"(f[i]), "v"(seedf), "v"(f[(i + 1) % UNROLL])); }
            if (OP == 12) { asm volatile("v_min_u32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) % UNROLL])); }
            if (OP == 13) { asm volatile("v_min3_u32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) % UNROLL]), "v"(u[(i + 2) % UNROLL])); }
            if (OP == 14) { asm volatile("v_add_u32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) % UNROLL])); }
            if (OP == 15) { asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) % UNROLL])); }
            if (OP == 16) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[i]) : "v"(f[i]), "v"(seedf), "v"(seedf)); }
            if (OP == 17) { asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) % UNROLL])); }
            if (OP == 18) { asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(u[i])); }
            if (OP == 19) { asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(u[i]) : "v"(u[i])); }
            if (OP == 20) { asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) % UNROLL]), "v"(u[(i + 2) % UNROLL])); }
            if (OP == 21) { asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[i]), "v"(seedf)); }
            if (OP == 22) { asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(f[i]) : "v"(f[i]), "v"(seedf)); }
            if (OP == 23) { asm volatile("v_max_f32_e64 %0, %1, %1" : "=v"(f[i]) : "v"(f[i])); }
            if (OP == 24) { asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(f[i]) : "v"(f[i]), "v"(seedf), "v"(seedf)); }
            if (OP == 25) { asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m[i % 4]) : "v"(f[i]), "v"(seedf)); }
            if (OP == 26) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[i]) : "v"(seedf), "v"(seedf)); }
            if (OP == 27) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d[i]) : "v"(d[i]), "v"(seedd)); }
            if (OP == 28) { asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f[i]) : "v"(u[i])); }
            if (OP == 29) { asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) % UNROLL])); }
            if (OP == 30) { asm volatile("v_sub_u32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) % UNROLL])); }
            if (OP == 31) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[i]), "v"(seedf)); }
        }
    }
    double acc = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc += d[i] + (double)f[i] + (double)u[i];
    acc += (double)(m[0] ^ m[1] ^ m[2] ^ m[3]);
    if (acc == 12345.678) out[0] = acc;
}

template <int OP> int run(const char* name, double* out)
{
    const int blocks = 256 * 8, threads = 256; // 8 waves/SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 1.0000001);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 1.0000001);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double waveInstr = (double)blocks * (threads / 64) * ITERS * UNROLL;
    const double perSimd = waveInstr / (256.0 * 4.0);
    const double ns = ms * 1e6 / perSimd;
    printf("%-16s %8.3f ms  %6.3f ns/wave-instr/SIMD  (= %.2f cycles @2.4GHz, %.2f @2.0GHz)\n", name, ms, ns, ns * 2.4, ns * 2.0);
    return 0;
}

int main()
{
    double* out; CHECK(hipMalloc((void**)&out, 64));
    run<0>("v_add_f64", out); run<1>("v_mul_f64", out); run<2>("v_fma_f64", out);
    run<3>("v_cvt_f64_f32", out); run<4>("v_cvt_f32_f64", out);
    run<5>("v_mul_f32", out); run<6>("v_add_f32", out); run<7>("v_max_f32", out);
    run<8>("v_cndmask_b32", out); run<9>("v_cmp_gt_f32 vcc", out); run<10>("v_addc_co_u32", out); run<11>("v_min3_f32", out);
    run<12>("v_min_u32", out); run<13>("v_min3_u32", out); run<14>("v_add_u32", out); run<15>("v_and_b32", out);
    run<16>("v_fma_f32", out); run<17>("v_mov_b32", out); run<18>("v_cvt_f32_u32", out); run<19>("v_lshlrev_b32", out);
    run<20>("v_add3_u32", out); run<21>("v_sub_f32", out); run<22>("v_mul_f32 clamp", out); run<23>("v_max_f32 x,x", out);
    run<24>("v_med3_f32", out); run<25>("v_cmp_gt_f32 sgpr", out); run<26>("v_fmac_f32", out); run<27>("v_pk_mul_f32", out);
    run<28>("v_cvt_f32_f16", out); run<29>("v_mul_u32_u24", out); run<30>("v_sub_u32", out); run<31>("v_add_f32 asm", out);
    return 0;
}
