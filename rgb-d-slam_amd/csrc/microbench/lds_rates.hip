#include <hip/hip_runtime.h>
#include <cstdio>
// Microbenchmark: cost of LDS read instructions for the access patterns of the ordered traversals (cape_staged.h): eight
// reads + one s_waitcnt per wave, 4 / 8 / 12 waves on one CU.  ds_read2_b64 costs four times a ds_read_b64; a wave gets a
// read through only every ~16 cycles whatever the CU load; masking lanes off does not make a read cheaper.
// (the two "ds_read_b128 pattern" rows use 8-byte aligned addresses on purpose: the misaligned slow path)
template <int MODE> __global__ void k(double* out, unsigned long long* ticks, int iters)
{
    __shared__ __attribute__((aligned(16))) double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned base = (unsigned)(size_t)(lds);
    unsigned addr;
    if (MODE == 0 || MODE == 3) addr = base + 8u * lane;                      // conflict-free, 64 distinct
    if (MODE == 1 || MODE == 2 || MODE == 4 || MODE == 6 || MODE == 8 || MODE == 9 || MODE == 10) addr = base + 8u * (lane < 10 ? lane : 0); // traversal pattern
    if (MODE == 5) addr = base;                                               // uniform
    if (MODE == 7) addr = base + 8u * (lane & 15);                            // 16 distinct, repeated per row
    double2 r[8];
    for (int u = 0; u < 8; ++u) r[u] = make_double2(0, 0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if ((MODE == 4 || MODE == 9) && lane >= 16) iters = 0;   // ONE exec mask around everything: only the first row reads
    if (MODE == 10 && lane >= 10) iters = 0;
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int u = 0; u < 8; ++u)
        {
            if (MODE == 0 || MODE == 1 || MODE == 7) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[u].x) : "v"(addr), "n"(u * 80) : "memory");
            if (MODE == 2) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(r[u]) : "v"(addr), "n"(u * 20), "n"(u * 20 + 10) : "memory");
            if (MODE == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[u]) : "v"(addr + 8u * lane), "n"(u * 16) : "memory");
            if (MODE == 4 || MODE == 10) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[u].x) : "v"(addr), "n"(u * 80) : "memory");
            if (MODE == 8 || MODE == 9) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[u]) : "v"(addr), "n"(u * 16) : "memory");
            if (MODE == 5) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[u]) : "v"(addr), "n"(u * 16) : "memory");
            if (MODE == 6) { unsigned w; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(w) : "v"(addr), "n"(u * 80) : "memory"); r[u].x = w; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    __syncthreads();
    double s = 0; for (int u = 0; u < 8; ++u) s += r[u].x;
    if (s == 12345.0) out[0] = s;
}
int main()
{
    double* out; unsigned long long* t; (void)hipMalloc(&out, 64); (void)hipMallocManaged(&t, 64);
    const char* names[] = {"ds_read_b64  64 distinct", "ds_read_b64  lanes>=10 -> lane 0", "ds_read2_b64 lanes>=10 -> lane 0", "ds_read_b128 64 distinct", "ds_read_b64  pattern, exec = 16 lanes", "ds_read_b128 uniform", "ds_read_b32  lanes>=10 -> lane 0", "ds_read_b64  lane&15 pattern", "ds_read_b128 pattern, all lanes", "ds_read_b128 pattern, exec = 16 lanes", "ds_read_b64  pattern, exec = 10 lanes"};
    const int iters = 4000;
    for (int waves : {1, 2, 3})
        for (int op = 0; op < 11; ++op)
        {
            for (int rep = 0; rep < 2; ++rep)
            {
                dim3 g(1), b(64 * waves * 4);
                switch (op) {
                case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, t, iters); break; case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, t, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, t, iters); break; case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, t, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, g, b, 0, 0, out, t, iters); break; case 5: hipLaunchKernelGGL(k<5>, g, b, 0, 0, out, t, iters); break;
                case 6: hipLaunchKernelGGL(k<6>, g, b, 0, 0, out, t, iters); break; case 7: hipLaunchKernelGGL(k<7>, g, b, 0, 0, out, t, iters); break;
                case 8: hipLaunchKernelGGL(k<8>, g, b, 0, 0, out, t, iters); break; case 9: hipLaunchKernelGGL(k<9>, g, b, 0, 0, out, t, iters); break;
                case 10: hipLaunchKernelGGL(k<10>, g, b, 0, 0, out, t, iters); break; }
                (void)hipDeviceSynchronize();
            }
            const double perInstrCU = (double)t[0] / (iters * 8.0) / (waves * 4.0);
            printf("%2d waves/CU  %-36s %.1f ticks per 8 reads per wave = %.2f clk per instruction on the CU\n", waves * 4, names[op], (double)t[0] / iters, perInstrCU);
        }
    return 0;
}
