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
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { d[i] = seedd + i + threadIdx.x; f[i] = seedf + i + threadIdx.x; }
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
        }
    }
    double acc = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc += d[i] + (double)f[i];
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
    return 0;
}
