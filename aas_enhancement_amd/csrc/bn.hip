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
                                                       const double* __restrict__ wsd, const double* __restrict__ rs_dev,
                                                       int64_t Rs_host) {
    const int cl = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (c >= C) return;
    // rows behind the statistics: this launch's R, or the all-reduced row count under SyncBN (device or host value)
    const double Rs = rs_dev ? rs_dev[0] : (double)Rs_host;
    const double mean_d = wsd[c] / Rs;
    double var_d = wsd[C + c] / Rs - mean_d * mean_d;
    if (var_d < 0.0) var_d = 0.0;
    const float mean = (float)mean_d;
    const float invstd = (float)(1.0 / sqrt(var_d + (double)eps));
    if (blockIdx.y == 0 && w == 0) {
        stats[c] = mean;
        stats[C + c] = invstd;
        if (rmean) {
            const double unb = Rs > 1.0 ? var_d * Rs / (Rs - 1.0) : var_d;
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
                                                           const double* __restrict__ wsd, const double* __restrict__ wsd_local,
                                                           const double* __restrict__ rs_dev, int64_t Rs_host) {
    const int cl = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (c >= C) return;
    const float sdy = (float)wsd[c], sdyx = (float)wsd[C + c];
    if (blockIdx.y == 0 && w == 0) {
        stats[2 * C + c] = sdy;
        stats[3 * C + c] = sdyx;
        // parameter gradients are THIS rank's sums (the gradient all-reduce adds the ranks up); dx uses the global sums
        const float pdy = wsd_local ? (float)wsd_local[c] : sdy, pdyx = wsd_local ? (float)wsd_local[C + c] : sdyx;
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + pdyx : pdyx;
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + pdy : pdy;
    }
    const float Rs = rs_dev ? (float)rs_dev[0] : (float)Rs_host;
    const float mean = stats[c], invstd = stats[C + c], g = gamma[c], b = beta[c];
    const float k = g * invstd, m1 = sdy / Rs, m2 = sdyx / Rs;
    const int64_t r0 = (int64_t)blockIdx.y * RPB;
    const int64_t r1 = r0 + RPB < R ? r0 + RPB : R;
    for (int64_t r = r0 + w; r < r1; r += 4) {
        const float xh = (x[r * C + c] - mean) * invstd;
        float d = dy[r * C + c];
        if (slope != 1.f && !(xh * g + b > 0.f)) d *= slope;
        dx[r * C + c] = k * (d - m1 - xh * m2);
    }
}

// eval mode: y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta (+ LeakyReLU)
__global__ __launch_bounds__(256) void bn_eval_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t R, int C,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                      float eps, float slope) {
    const int cl = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (c >= C) return;
    const float g = gamma[c] / sqrtf(rvar[c] + eps), b = beta[c] - rmean[c] * g;
    const int64_t r0 = (int64_t)blockIdx.y * RPB;
    const int64_t r1 = r0 + RPB < R ? r0 + RPB : R;
    for (int64_t r = r0 + w; r < r1; r += 4) {
        float v = x[r * C + c] * g + b;
        y[r * C + c] = v > 0.f ? v : v * slope;
    }
}

// row softmax over C <= 64 classes: one wavefront per row, max / sum by cross-lane reduction
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t R, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float v = lane < C ? x[r * C + lane] : -INFINITY;
    float m = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const float e = lane < C ? __expf(v - m) : 0.f;
    const float sum = wave_sum(e);
    if (lane < C) y[r * C + lane] = e / sum;
}

}  // namespace

extern "C" int aas_bn_stats(aasStream_t stream, const float* x, int64_t R, int C, double* wsd) {
    AAS_CHECK(x && wsd && R > 0 && C > 0, "aas_bn_stats: bad args");
    hipStream_t s = (hipStream_t)stream;
    AAS_HIP(hipMemsetAsync(wsd, 0, sizeof(double) * 2 * C, s));
    dim3 grid(cdiv(C, 64), cdiv(R, RPB));
    hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, s, x, R, C, wsd);
    AAS_LAUNCH_CHECK("aas_bn_stats");
    return 0;
}

extern "C" int aas_bn_apply(aasStream_t stream, const float* x, float* y, int64_t R, int C, const float* gamma,
                            const float* beta, float eps, float slope, float* stats, float* running_mean,
                            float* running_var, float momentum, const double* wsd, const double* d_rows) {
    AAS_CHECK(x && y && gamma && beta && stats && wsd && R > 0 && C > 0, "aas_bn_apply: bad args");
    AAS_CHECK((running_mean == nullptr) == (running_var == nullptr), "aas_bn_apply: running stats must both be set or both NULL");
    dim3 grid(cdiv(C, 64), cdiv(R, RPB));
    hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, R, C, gamma, beta, eps, slope, stats,
                       running_mean, running_var, momentum, wsd, d_rows, R);
    AAS_LAUNCH_CHECK("aas_bn_apply");
    return 0;
}

extern "C" int aas_bn_bwd_reduce(aasStream_t stream, const float* x, const float* dy, int64_t R, int C, const float* gamma,
                                 const float* beta, float slope, const float* stats, double* wsd) {
    AAS_CHECK(x && dy && gamma && beta && stats && wsd && R > 0 && C > 0, "aas_bn_bwd_reduce: bad args");
    hipStream_t s = (hipStream_t)stream;
    AAS_HIP(hipMemsetAsync(wsd, 0, sizeof(double) * 2 * C, s));
    dim3 grid(cdiv(C, 64), cdiv(R, RPB));
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(256), 0, s, x, dy, R, C, gamma, beta, slope, stats, wsd);
    AAS_LAUNCH_CHECK("aas_bn_bwd_reduce");
    return 0;
}

extern "C" int aas_bn_bwd_apply(aasStream_t stream, const float* x, const float* dy, float* dx, int64_t R, int C,
                                const float* gamma, const float* beta, float slope, float* stats, float* dgamma,
                                float* dbeta, int accumulate, const double* wsd, const double* wsd_local,
                                const double* d_rows) {
    AAS_CHECK(x && dy && dx && gamma && beta && stats && wsd && R > 0 && C > 0, "aas_bn_bwd_apply: bad args");
    dim3 grid(cdiv(C, 64), cdiv(R, RPB));
    hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, dy, dx, R, C, gamma, beta, slope,
                       stats, dgamma, dbeta, accumulate, wsd, wsd_local, d_rows, R);
    AAS_LAUNCH_CHECK("aas_bn_bwd_apply");
    return 0;
}

extern "C" int aas_bn_eval(aasStream_t stream, const float* x, float* y, int64_t R, int C, const float* gamma,
                           const float* beta, const float* running_mean, const float* running_var, float eps,
                           float slope) {
    AAS_CHECK(x && y && gamma && beta && running_mean && running_var && R > 0 && C > 0, "aas_bn_eval: bad args");
    dim3 grid(cdiv(C, 64), cdiv(R, RPB));
    hipLaunchKernelGGL(bn_eval_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, R, C, gamma, beta, running_mean,
                       running_var, eps, slope);
    AAS_LAUNCH_CHECK("aas_bn_eval");
    return 0;
}

extern "C" int aas_softmax_rows(aasStream_t stream, const float* x, float* y, int64_t R, int C) {
    AAS_CHECK(x && y && R > 0 && C > 0 && C <= 64, "aas_softmax_rows: needs 1 <= C <= 64 (got %d)", C);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, x, y, R, C);
    AAS_LAUNCH_CHECK("aas_softmax_rows");
    return 0;
}

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
                       running_var, momentum, wsd, (const double*)nullptr, R);
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
                       dbeta, accumulate, wsd, (const double*)nullptr, (const double*)nullptr, R);
    AAS_LAUNCH_CHECK("aas_bn_bwd");
    return 0;
}
