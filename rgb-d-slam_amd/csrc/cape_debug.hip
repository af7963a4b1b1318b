// Debug / parity entry: evaluates device scalar math on caller-supplied operands so tests can compare the
// gfx950 instruction sequences (f64 div / sqrt expansion, ocml acos / atan2, the restated eigen-solver and plane
// fit) against the CPU oracle bit for bit.  Not on the hot path.
#include <hip/hip_runtime.h>

#include <string>

#include "cape_device.h"
#include "cape_internal.h"

namespace cape {

__global__ void debug_eval_kernel(int op, const double* a, const double* b, double* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    switch (op)
    {
    case CAPE_DEBUG_SQRT: out[i] = sqrt(a[i]); break;
    case CAPE_DEBUG_DIV: out[i] = a[i] / b[i]; break;
    case CAPE_DEBUG_ACOS: out[i] = acos(a[i]); break;
    case CAPE_DEBUG_ATAN2: out[i] = atan2(a[i], b[i]); break;
    case CAPE_DEBUG_QUANT: out[i] = depth_quantization(a[i]); break;
    case CAPE_DEBUG_SQRTF: out[i] = (double)sqrtf((float)a[i]); break;
    case CAPE_DEBUG_EIGEN3:
    {
        // a: n x 6 (m00 m10 m11 m20 m21 m22) ; out: n x 12 (3 eigenvalues, 9 eigenvector entries row-major)
        const double* m = a + (size_t)i * 6;
        Eig3 e;
        self_adjoint_eigen3(m[0], m[1], m[2], m[3], m[4], m[5], e);
        double* o = out + (size_t)i * 12;
        o[0] = e.val[0]; o[1] = e.val[1]; o[2] = e.val[2];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                o[3 + 3 * r + c] = e.q[r][c];
        break;
    }
    case CAPE_DEBUG_FIT_PLANE:
    {
        // a: n x 10 (9 sums + count) ; out: n x 10 (nx ny nz d cx cy cz mse score planar)
        const double* s = a + (size_t)i * 10;
        double S[9];
        for (int k = 0; k < 9; ++k)
            S[k] = s[k];
        PlaneFit f;
        fit_plane(S, (uint32_t)s[9], f);
        double* o = out + (size_t)i * 10;
        o[0] = f.nx; o[1] = f.ny; o[2] = f.nz; o[3] = f.d;
        o[4] = f.cx; o[5] = f.cy; o[6] = f.cz;
        o[7] = f.mse; o[8] = f.score; o[9] = f.planar ? 1.0 : 0.0;
        break;
    }
    default: out[i] = 0.0;
    }
}

} // namespace cape

extern "C" int cape_debug_eval(int op, const double* a, const double* b, double* out, int n)
{
    if (!a || !out || n <= 0)
        return CAPE_ERR_INVALID_ARGUMENT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return CAPE_ERR_NO_DEVICE;
    size_t inW = 1, outW = 1;
    if (op == CAPE_DEBUG_EIGEN3) { inW = 6; outW = 12; }
    if (op == CAPE_DEBUG_FIT_PLANE) { inW = 10; outW = 10; }
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    int rc = CAPE_OK;
    if (hipMalloc((void**)&da, n * inW * 8) != hipSuccess || hipMalloc((void**)&dout, n * outW * 8) != hipSuccess)
        rc = CAPE_ERR_HIP;
    if (rc == CAPE_OK && b && hipMalloc((void**)&db, (size_t)n * 8) != hipSuccess)
        rc = CAPE_ERR_HIP;
    if (rc == CAPE_OK)
    {
        (void)hipMemcpy(da, a, n * inW * 8, hipMemcpyHostToDevice);
        if (b)
            (void)hipMemcpy(db, b, (size_t)n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cape::debug_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op, da, db ? db : da, dout, n);
        if (hipMemcpy(out, dout, n * outW * 8, hipMemcpyDeviceToHost) != hipSuccess)
            rc = CAPE_ERR_HIP;
    }
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    return rc;
}
