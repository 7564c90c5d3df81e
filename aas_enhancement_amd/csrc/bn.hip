// Train-mode BatchNorm over rows-by-channels [R, C] (+ fused LeakyReLU), forward and backward.
// HBM-bound: two passes over x (statistics, apply); per-column partial sums are accumulated in
// fp64 device atomics so that E[x^2]-mean^2 cannot cancel catastrophically.
// Reference: nn.BatchNorm1d in train mode at Speech_enhancement_by_AAS/model.py:72,82 (via
// SequenceWise :44-49), :290,:298 (+ LeakyReLU(slope=map) :291,:299), :316.
#include "common.h"

namespace {

constexpr int RPB = 128;  // rows per block

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int64_t R, int C,
                                                       double* __restrict__ wsd) {
    __shared__ float ps[4][64], pq[4][64];
    const int cl = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int64_t r0 = (int64_t)blockIdx.y * RPB;
    const int64_t r1 = r0 + RPB < R ? r0 + RPB : R;
    float s = 0.f, q = 0.f;
    if (c < C)
        for (int64_t r = r0 + w; r < r1; r += 4) {
            float v = x[r * C + c];
            s += v;
            q += v * v;
        }
    ps[w][cl] = s;
    pq[w][cl] = q;
    __syncthreads();
    if (w == 0 && c < C) {
        atomicAdd(wsd + c, (double)ps[0][cl] + (double)ps[1][cl] + (double)ps[2][cl] + (double)ps[3][cl]);
        atomicAdd(wsd + C + c, (double)pq[0][cl] + (double)pq[1][cl] + (double)pq[2][cl] + (double)pq[3][cl]);
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t R,
                                                       int C, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, float slope,
                                                       float* __restrict__ stats, float* __restrict__ rmean,
                                                       float* __restrict__ rvar, float momentum,
                                                       const double* __restrict__ wsd) {
    const int cl = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (c >= C) return;
    const double mean_d = wsd[c] / (double)R;
    double var_d = wsd[C + c] / (double)R - mean_d * mean_d;
    if (var_d < 0.0) var_d = 0.0;
    const float mean = (float)mean_d;
    const float invstd = (float)(1.0 / sqrt(var_d + (double)eps));
    if (blockIdx.y == 0 && w == 0) {
        stats[c] = mean;
        stats[C + c] = invstd;
        if (rmean) {
            const double unb = R > 1 ? var_d * (double)R / (double)(R - 1) : var_d;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
        }
    }
    const float g = gamma[c] * invstd, b = beta[c] - mean * g;
    const int64_t r0 = (int64_t)blockIdx.y * RPB;
    const int64_t r1 = r0 + RPB < R ? r0 + RPB : R;
    for (int64_t r = r0 + w; r < r1; r += 4) {
        float v = x[r * C + c] * g + b;
        y[r * C + c] = v > 0.f ? v : v * slope;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            int64_t R, int C, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float slope,
                                                            const float* __restrict__ stats, double* __restrict__ wsd) {
    __shared__ float ps[4][64], pq[4][64];
    const int cl = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int64_t r0 = (int64_t)blockIdx.y * RPB;
    const int64_t r1 = r0 + RPB < R ? r0 + RPB : R;
    float s = 0.f, q = 0.f;
    if (c < C) {
        const float mean = stats[c], invstd = stats[C + c], g = gamma[c], b = beta[c];
        for (int64_t r = r0 + w; r < r1; r += 4) {
            const float xh = (x[r * C + c] - mean) * invstd;
            float d = dy[r * C + c];
            if (slope != 1.f && !(xh * g + b > 0.f)) d *= slope;
            s += d;
            q += d * xh;
        }
    }
    ps[w][cl] = s;
    pq[w][cl] = q;
    __syncthreads();
    if (w == 0 && c < C) {
        atomicAdd(wsd + c, (double)ps[0][cl] + (double)ps[1][cl] + (double)ps[2][cl] + (double)ps[3][cl]);
        atomicAdd(wsd + C + c, (double)pq[0][cl] + (double)pq[1][cl] + (double)pq[2][cl] + (double)pq[3][cl]);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dx, int64_t R, int C,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float slope,
                                                           float* __restrict__ stats, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int accumulate,
                                                           const double* __restrict__ wsd) {
    const int cl = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (c >= C) return;
    const float sdy = (float)wsd[c], sdyx = (float)wsd[C + c];
    if (blockIdx.y == 0 && w == 0) {
        stats[2 * C + c] = sdy;
        stats[3 * C + c] = sdyx;
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + sdyx : sdyx;
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + sdy : sdy;
    }
    const float mean = stats[c], invstd = stats[C + c], g = gamma[c], b = beta[c];
    const float k = g * invstd, m1 = sdy / (float)R, m2 = sdyx / (float)R;
    const int64_t r0 = (int64_t)blockIdx.y * RPB;
    const int64_t r1 = r0 + RPB < R ? r0 + RPB : R;
    for (int64_t r = r0 + w; r < r1; r += 4) {
        const float xh = (x[r * C + c] - mean) * invstd;
        float d = dy[r * C + c];
        if (slope != 1.f && !(xh * g + b > 0.f)) d *= slope;
        dx[r * C + c] = k * (d - m1 - xh * m2);
    }
}

}  // namespace

extern "C" int aas_bn_fwd(aasStream_t stream, const float* x, float* y, int64_t R, int C, const float* gamma,
                          const float* beta, float eps, float slope, float* stats, float* running_mean,
                          float* running_var, float momentum, double* wsd) {
    AAS_CHECK(x && y && gamma && beta && stats && wsd && R > 0 && C > 0, "aas_bn_fwd: bad args");
    AAS_CHECK((running_mean == nullptr) == (running_var == nullptr), "aas_bn_fwd: running stats must both be set or both NULL");
    hipStream_t s = (hipStream_t)stream;
    AAS_HIP(hipMemsetAsync(wsd, 0, sizeof(double) * 2 * C, s));
    dim3 grid(cdiv(C, 64), cdiv(R, RPB));
    hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, s, x, R, C, wsd);
    hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(256), 0, s, x, y, R, C, gamma, beta, eps, slope, stats, running_mean,
                       running_var, momentum, wsd);
    AAS_LAUNCH_CHECK("aas_bn_fwd");
    return 0;
}

extern "C" int aas_bn_bwd(aasStream_t stream, const float* x, const float* dy, float* dx, int64_t R, int C,
                          const float* gamma, const float* beta, float slope, float* stats, float* dgamma,
                          float* dbeta, int accumulate, double* wsd) {
    AAS_CHECK(x && dy && dx && gamma && beta && stats && wsd && R > 0 && C > 0, "aas_bn_bwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    AAS_HIP(hipMemsetAsync(wsd, 0, sizeof(double) * 2 * C, s));
    dim3 grid(cdiv(C, 64), cdiv(R, RPB));
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(256), 0, s, x, dy, R, C, gamma, beta, slope, stats, wsd);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, s, x, dy, dx, R, C, gamma, beta, slope, stats, dgamma,
                       dbeta, accumulate, wsd);
    AAS_LAUNCH_CHECK("aas_bn_bwd");
    return 0;
}
