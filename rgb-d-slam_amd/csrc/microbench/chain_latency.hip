// Latency microbenchmark behind DESIGN.md's stage-B budget: cost of one step of the dependent chains the grow kernel
// cannot avoid (ordered f64 sums), in s_memtime ticks, and what a tick is (against the 100 MHz s_memrealtime counter).
// One wavefront per workgroup; `waves` workgroups per launch to see the effect of SIMD sharing.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void chain_kernel(double* out, unsigned long long* ticks, int n, double x)
{
    __shared__ double s[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x)
        s[i] = x + i;
    __syncthreads();
    double acc = x;
    const unsigned long long r0 = wall_clock64();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // (1) dependent register chain
#pragma unroll 16
    for (int i = 0; i < n; ++i)
        acc += x;
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    // (2) ordered sum from LDS, 8 elements per trip with the next 8 requested ahead (as in ordered_sum_lds)
    {
        const double2* v = reinterpret_cast<const double2*>(s);
        double2 a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
        for (int j = 0; j + 8 <= n; j += 8)
        {
            const int k = ((j + 8) & 1016) / 2;
            const double2 b0 = v[k], b1 = v[k + 1], b2 = v[k + 2], b3 = v[k + 3];
            acc += a0.x; acc += a0.y; acc += a1.x; acc += a1.y; acc += a2.x; acc += a2.y; acc += a3.x; acc += a3.y;
            a0 = b0, a1 = b1, a2 = b2, a3 = b3;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    // (3) naive LDS chain: load, wait, add
    for (int j = 0; j < n; ++j)
    {
        acc += s[j & 1023];
        __builtin_amdgcn_s_waitcnt(0);
    }
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = wall_clock64();
    if (threadIdx.x == 0)
    {
        unsigned long long* o = ticks + blockIdx.x * 5;
        o[0] = t1 - t0, o[1] = t2 - t1, o[2] = t3 - t2, o[3] = t3 - t0, o[4] = r1 - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    const int n = 4096;
    for (int groups : {1, 256 * 4, 256 * 8, 256 * 12})
    {
        double* out;
        unsigned long long* ticks;
        hipMalloc(&out, sizeof(double) * 64 * groups);
        hipMalloc(&ticks, sizeof(unsigned long long) * 5 * groups);
        for (int rep = 0; rep < 2; ++rep)
            hipLaunchKernelGGL(chain_kernel, dim3(groups), dim3(64), 0, 0, out, ticks, n, 1.0000001);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(5 * groups);
        hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
        double m[5] = {0, 0, 0, 0, 0};
        for (int g = 0; g < groups; ++g)
            for (int k = 0; k < 5; ++k)
                m[k] += (double)h[g * 5 + k] / groups;
        std::printf("waves %5d : reg chain %.2f ticks/add | pipelined LDS sum %.2f ticks/elem | naive LDS chain %.2f ticks/elem | "
                    "s_memtime ticks per 100MHz tick %.2f (=> %.0f MHz)\n",
                    groups, m[0] / n, m[1] / n, m[2] / n, m[3] / m[4], m[3] / m[4] * 100.0);
        hipFree(out);
        hipFree(ticks);
    }
    return 0;
}
