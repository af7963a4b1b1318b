#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// Microbenchmark: what does a dispatch cost as a function of its dynamic LDS request, when the workgroups do (almost) nothing?
// Found while profiling the polygon matcher: a persistent grid of 256 workgroups that found an empty work list
// and left at once was on the timeline for 240 us with 52 KB of LDS per workgroup, 11 us with 1 KB.
// Kernels: exit at once (EXIT), or touch one LDS word first (TOUCH).
// REGS: 0 = a handful of VGPRs, 1 = 200 VGPRs (clobber), 2 = 250 VGPRs + 16 AGPRs
template <int TOUCH, int REGS = 0> __global__ __launch_bounds__(64) void k(const unsigned* count, unsigned* out)
{
    extern __shared__ unsigned lds[];
    if (blockIdx.x >= *count)
        return;
    if (REGS == 1)
        asm volatile("v_mov_b32 v199, 0" ::: "v199");
    if (REGS == 2)
        asm volatile("v_mov_b32 v249, 0\n v_accvgpr_write_b32 a15, v249" ::: "v249", "a15");
    if (TOUCH)
        lds[threadIdx.x] = threadIdx.x;
    out[blockIdx.x] = TOUCH ? lds[threadIdx.x ^ 1] : 1u;
}
int main()
{
    unsigned *count, *out;
    hipMalloc(&count, 4);
    hipMalloc(&out, 1 << 22);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int ldsSizes[] = {0, 1024, 8192, 16384, 32768, 49152, 53248, 65536};
    const int grids[] = {1, 64, 256, 1024, 4096, 65536};
    for (int mode = 0; mode < 5; ++mode) // 0: empty list (all exit), 1: every workgroup stores a word, 2: touches LDS too, 3/4: as 0 with many registers
    {
        const unsigned c = (mode == 0 || mode >= 3) ? 0u : 0xffffffffu;
        hipMemcpy(count, &c, 4, hipMemcpyHostToDevice);
        printf("mode %d (%s)\n  lds\\grid", mode, mode == 0 ? "all workgroups exit at once" : mode == 1 ? "one global store per thread" : mode == 2 ? "LDS word + global store" : mode == 3 ? "exit at once, 200 VGPRs" : "exit at once, 250 VGPRs + 16 AGPRs");
        for (int g : grids)
            printf(" %8d", g);
        printf("   [us per dispatch]\n");
        for (int l : ldsSizes)
        {
            printf("  %7d ", l);
            for (int g : grids)
            {
                auto launch = [&] {
                    if (mode == 2)
                        hipLaunchKernelGGL(k<1>, dim3(g), dim3(64), l, 0, count, out);
                    else if (mode == 3)
                        hipLaunchKernelGGL((k<0, 1>), dim3(g), dim3(64), l, 0, count, out);
                    else if (mode == 4)
                        hipLaunchKernelGGL((k<0, 2>), dim3(g), dim3(64), l, 0, count, out);
                    else
                        hipLaunchKernelGGL(k<0>, dim3(g), dim3(64), l, 0, count, out);
                };
                if (mode == 2 && l < 256)
                {
                    printf(" %8s", "-");
                    continue;
                }
                for (int i = 0; i < 3; ++i)
                    launch();
                hipEventRecord(e0, 0);
                const int reps = 20;
                for (int i = 0; i < reps; ++i)
                    launch();
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                printf(" %8.1f", 1e3 * ms / reps);
            }
            printf("\n");
        }
    }
    return 0;
}
