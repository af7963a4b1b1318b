#include <hip/hip_runtime.h>
#include <cstdio>
// Microbenchmark: could the matrix core take some of A1's exact f64 additions off the vector ALU?  (With B = identity a
// v_mfma_f64_4x4x4 is 64 per-lane additions, and A1's sums are exact in any order.)  Question: do v_mfma_f64_4x4x4 and
// v_add_f64 run concurrently on gfx950, or do they share the f64 datapath?  Answer: they add up -- 16 adds take 77 cycles,
// 4 MFMAs 68, both together 165 per wave-slot at four waves per SIMD.  Nothing to gain.
template <int NADD, int NMFMA> __global__ void k(double* out, double seed, int iters)
{
    double a[16], m[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = seed * i;
    const double one = (threadIdx.x & 3) == ((threadIdx.x >> 2) & 3) ? 1.0 : 0.0;
    const double inc = seed;
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
#pragma unroll
            for (int u = 0; u < NADD / 4; ++u)
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[(r * (NADD / 4) + u) & 15]) : "v"(inc));
#pragma unroll
            for (int u = 0; u < NMFMA / 4; ++u)
                m[(r * (NMFMA / 4) + u) & 7] = __builtin_amdgcn_mfma_f64_4x4x4f64(inc, one, m[(r * (NMFMA / 4) + u) & 7], 0, 0, 0);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += m[i];
    if (s == 12345.678) out[0] = s;
}
template <int NADD, int NMFMA> void run(const char* name, double* out, int wavesPerSimd)
{
    const int iters = 20000, blocks = 256, threads = 64 * 4 * wavesPerSimd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NADD, NMFMA>), dim3(blocks), dim3(threads), 0, 0, out, 1.0000001, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NADD, NMFMA>), dim3(blocks), dim3(threads), 0, 0, out, 1.0000001, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wavesPerSimd waves x iters iterations
    const double nsPerIterPerSimd = ms * 1e6 / (iters * (double)wavesPerSimd);
    printf("%d waves/SIMD  %-34s %8.3f ms   %.1f ns per iteration per wave-slot (= %.1f cycles @2.4GHz)\n", wavesPerSimd, name, ms, nsPerIterPerSimd, nsPerIterPerSimd * 2.4);
}
int main()
{
    double* out; (void)hipMalloc(&out, 64);
    for (int w : {1, 2, 4})
    {
        run<16, 0>("16 v_add_f64", out, w);
        run<0, 4>("4 v_mfma_f64_4x4x4", out, w);
        run<0, 8>("8 v_mfma_f64_4x4x4", out, w);
        run<16, 4>("16 v_add_f64 + 4 mfma", out, w);
        run<16, 8>("16 v_add_f64 + 8 mfma", out, w);
        run<8, 4>("8 v_add_f64 + 4 mfma", out, w);
    }
    return 0;
}
