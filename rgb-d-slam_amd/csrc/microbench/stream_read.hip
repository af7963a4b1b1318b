// Calibration microbenchmark for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section):
// streams a known number of bytes with the same access width as the cell-fit kernel (16 B per lane, rows of
// 2560 B) and writes a known number of bytes, so the counters can be scaled to true bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void stream_read_f4(const float4* __restrict__ in, float* __restrict__ out, size_t n4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n4; i += stride)
    {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

__global__ void stream_write_f64(double* __restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        out[i] = (double)i;
}

int main(int argc, char** argv)
{
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 4096ull) << 20; // MiB
    float4* in; float* out; double* wbuf;
    CHECK(hipMalloc((void**)&in, bytes));
    CHECK(hipMalloc((void**)&out, 64));
    CHECK(hipMalloc((void**)&wbuf, bytes / 8));
    CHECK(hipMemset(in, 0, bytes));
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep)
    {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream_read_f4, dim3(256 * 8), dim3(256), 0, 0, in, out, bytes / 16);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("stream_read_f4  %zu bytes  %.3f ms  %.1f GB/s\n", bytes, ms, bytes / ms / 1e6);
    }
    for (int rep = 0; rep < 3; ++rep)
    {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream_write_f64, dim3(256 * 8), dim3(256), 0, 0, wbuf, bytes / 64);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("stream_write_f64 %zu bytes  %.3f ms  %.1f GB/s\n", bytes / 8, ms, bytes / 8 / ms / 1e6);
    }
    return 0;
}
