#include <hip/hip_runtime.h>
#include <cstdio>
// Microbenchmark: ONE wave on a SIMD, a dependent v_add_f64 chain (the shape of every ordered sum of the grow kernels) with
// k independent cheap instructions per step.  Question: what fits in the shadow of the dependent add?  Answer (gfx950):
// nothing -- every extra instruction, vector or scalar, adds 4-8 cycles to the step.  A lone wave is bound by its
// instruction COUNT, not by the chain's latency; only other waves on the SIMD fill the gaps.
template <int KIND, int K> __global__ void k(double* out, unsigned long long* ticks, double seed, int iters)
{
    double a = seed + threadIdx.x, b = seed * 3;
    unsigned m0 = threadIdx.x, m1 = threadIdx.x + 1, m2 = threadIdx.x + 2, m3 = threadIdx.x + 3;
    unsigned s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int u = 0; u < 16; ++u)
        {
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));
            if (KIND == 0) { // v_mov_b32
                if (K > 0) asm volatile("v_mov_b32 %0, %1" : "=v"(m0) : "v"(m1));
                if (K > 1) asm volatile("v_mov_b32 %0, %1" : "=v"(m2) : "v"(m3));
                if (K > 2) asm volatile("v_mov_b32 %0, %1" : "=v"(m1) : "v"(m2));
                if (K > 3) asm volatile("v_mov_b32 %0, %1" : "=v"(m3) : "v"(m0));
            }
            if (KIND == 1) { // v_readlane_b32 to SGPRs (unused)
                if (K > 0) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(m1));
                if (K > 1) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s1) : "v"(m3));
                if (K > 2) asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(s2) : "v"(m2));
                if (K > 3) asm volatile("v_readlane_b32 %0, %1, 9" : "=s"(s3) : "v"(m0));
            }
            if (KIND == 2) { // s_mov (SALU)
                if (K > 0) asm volatile("s_mov_b32 %0, %1" : "=s"(s0) : "s"(s1));
                if (K > 1) asm volatile("s_mov_b32 %0, %1" : "=s"(s2) : "s"(s3));
                if (K > 2) asm volatile("s_mov_b32 %0, %1" : "=s"(s1) : "s"(s2));
                if (K > 3) asm volatile("s_mov_b32 %0, %1" : "=s"(s3) : "s"(s0));
            }
            if (KIND == 3) { // v_add_f64 independent (second chain)
                if (K > 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(b) : "v"(seed));
            }
            if (KIND == 4) { // the add takes an SGPR pair written by v_readlane one step earlier
                if (K > 0) { asm volatile("v_readlane_b32 %0, %2, 3\n v_readlane_b32 %1, %3, 3" : "=s"(s0), "=s"(s1) : "v"(m1), "v"(m2));
                             asm volatile("v_add_f64 %0, %0, s[2:3]" : "+v"(a) : : "s2", "s3"); }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    if (a + b + m0 + m1 + m2 + m3 + s0 + s1 + s2 + s3 == 12345.0) out[0] = a;
}
template <int KIND, int K> void run(const char* name, double* out, unsigned long long* t)
{
    const int iters = 5000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<KIND, K>), dim3(1), dim3(64), 0, 0, out, t, 1.0000001, iters); (void)hipDeviceSynchronize(); }
    printf("%-44s %.2f ticks per step\n", name, (double)t[0] / (iters * 16.0));
}
int main()
{
    double* out; unsigned long long* t; (void)hipMalloc(&out, 64); (void)hipMallocManaged(&t, 64);
    run<0, 0>("add only", out, t);
    run<0, 1>("add + 1 v_mov_b32", out, t); run<0, 2>("add + 2 v_mov_b32", out, t); run<0, 3>("add + 3 v_mov_b32", out, t); run<0, 4>("add + 4 v_mov_b32", out, t);
    run<1, 1>("add + 1 v_readlane_b32", out, t); run<1, 2>("add + 2 v_readlane_b32", out, t); run<1, 4>("add + 4 v_readlane_b32", out, t);
    run<2, 2>("add + 2 s_mov_b32", out, t); run<2, 4>("add + 4 s_mov_b32", out, t);
    run<3, 1>("add + 1 independent v_add_f64", out, t);
    return 0;
}
