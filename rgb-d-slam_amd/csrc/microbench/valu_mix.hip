// Microbenchmark: do f64-rate and f32-class / integer VALU instructions of DIFFERENT waves on one SIMD overlap (separate
// pipes) or serialise (one VALU port)?  Runs 8 waves per SIMD where even waves issue op A and odd waves op B, and compares
// with all waves on A, all on B.  If mixed ~ (A + B) / 2 the port is shared; if mixed ~ max(A, B) / 2 they overlap.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096, UNROLL = 16;

template <int OP> __device__ __forceinline__ void loop(double (&d)[UNROLL], float (&f)[UNROLL], unsigned (&u)[UNROLL], float seedf, double seedd)
{
    for (int it = 0; it < ITERS; ++it)
    {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i)
        {
            if (OP == 0) d[i] = d[i] + seedd;                                                           // v_add_f64
            if (OP == 1) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i])); }
            if (OP == 2) f[i] = f[i] * seedf;                                                           // v_mul_f32
            if (OP == 3) { asm volatile("v_max_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[i]), "v"(seedf)); }
            if (OP == 4) { asm volatile("v_min3_u32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) % UNROLL]), "v"(u[(i + 2) % UNROLL])); }
            if (OP == 5) { asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(seedf) : "vcc"); }
        }
    }
}

template <int OPA, int OPB> __global__ void k(double* out, float seedf, double seedd)
{
    double d[UNROLL];
    float f[UNROLL];
    unsigned u[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { d[i] = seedd + i + threadIdx.x; f[i] = seedf + i + threadIdx.x; u[i] = 7u * i + threadIdx.x; }
    if ((threadIdx.x >> 6) & 1) // wave-uniform: odd waves run op B, even waves op A, each in its own loop
        loop<OPB>(d, f, u, seedf, seedd);
    else
        loop<OPA>(d, f, u, seedf, seedd);
    double acc = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc += d[i] + (double)f[i] + (double)u[i];
    if (acc == 12345.678) out[0] = acc;
}

template <int A, int B> float run(double* out)
{
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 1.0000001);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 1.0000001);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double* out; CHECK(hipMalloc((void**)&out, 64));
    const char* names[] = {"v_add_f64", "v_cvt_f64_f32", "v_mul_f32", "v_max_f32", "v_min3_u32", "v_cmp_gt_f32"};
    const float a = run<0, 0>(out), c = run<1, 1>(out), m = run<2, 2>(out), x = run<3, 3>(out), n3 = run<4, 4>(out), cm = run<5, 5>(out);
    printf("alone (ms): %s %.3f  %s %.3f  %s %.3f  %s %.3f  %s %.3f  %s %.3f\n", names[0], a, names[1], c, names[2], m, names[3], x, names[4], n3, names[5], cm);
    printf("mixed v_add_f64 | v_mul_f32     : %.3f ms   (shared port: %.3f, overlapped: %.3f)\n", run<0, 2>(out), (a + m) / 2, (a > m ? a : m) / 2);
    printf("mixed v_add_f64 | v_max_f32     : %.3f ms   (shared port: %.3f, overlapped: %.3f)\n", run<0, 3>(out), (a + x) / 2, (a > x ? a : x) / 2);
    printf("mixed v_add_f64 | v_min3_u32    : %.3f ms   (shared port: %.3f, overlapped: %.3f)\n", run<0, 4>(out), (a + n3) / 2, (a > n3 ? a : n3) / 2);
    printf("mixed v_add_f64 | v_cmp_gt_f32  : %.3f ms   (shared port: %.3f, overlapped: %.3f)\n", run<0, 5>(out), (a + cm) / 2, (a > cm ? a : cm) / 2);
    printf("mixed v_add_f64 | v_cvt_f64_f32 : %.3f ms   (shared port: %.3f, overlapped: %.3f)\n", run<0, 1>(out), (a + c) / 2, (a > c ? a : c) / 2);
    printf("mixed v_mul_f32 | v_max_f32     : %.3f ms   (shared port: %.3f, overlapped: %.3f)\n", run<2, 3>(out), (m + x) / 2, (m > x ? m : x) / 2);
    return 0;
}
