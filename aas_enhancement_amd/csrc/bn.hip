// Train-mode BatchNorm over rows-by-channels [R, C] (+ fused LeakyReLU), forward and backward.
// HBM-bound: two passes over x (statistics, apply); per-column partial sums are accumulated in
// fp64 device atomics so that E[x^2]-mean^2 cannot cancel catastrophically.
// Reference: nn.BatchNorm1d in train mode at Speech_enhancement_by_AAS/model.py:72,82 (via
// SequenceWise :44-49), :290,:298 (+ LeakyReLU(slope=map) :291,:299), :316.
#include "common.h"

#include <mutex>
#include <unordered_map>

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
                                                       int64_t Rs_host, long long* __restrict__ nbt) {
    const int cl = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (nbt && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) nbt[0] += 1;   // num_batches_tracked
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

// ---- 16-byte path (C % 4 == 0, 16-byte aligned rows) ---------------------------------------------------------------------
// 256 threads = 16 column quads x 16 row lanes: a wave reads 4 rows x 256 contiguous bytes per load, every load of a thread is
// independent of the others.  Per-block column sums go to a workspace [RB][2][C] of doubles - no atomics and no zero-fill launch
// in front, so the statistics are the same bits on every run - and the consumer adds the RB partials up in fp64.
template <bool BWD>
__global__ __launch_bounds__(256) void bn_part4_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t R, int C, int RPB,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                                                       const float* __restrict__ stats, double* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float ps[16][68], pq[16][68];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    const int64_t r0 = (int64_t)blockIdx.y * RPB;
    const int64_t r1 = r0 + RPB < R ? r0 + RPB : R;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        f32x4 mean = s, invstd = s, g = s, b = s;
        if (BWD) {
            mean = *reinterpret_cast<const f32x4*>(stats + c);
            invstd = *reinterpret_cast<const f32x4*>(stats + C + c);
            g = *reinterpret_cast<const f32x4*>(gamma + c);
            b = *reinterpret_cast<const f32x4*>(beta + c);
        }
#pragma unroll 8
        for (int64_t r = r0 + rl; r < r1; r += 16) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + c);
            if (!BWD) {
                s += v;
                q += v * v;
            } else {
                const f32x4 xh = (v - mean) * invstd;
                f32x4 d = *reinterpret_cast<const f32x4*>(dy + r * C + c);
                if (slope != 1.f) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!(xh[e] * g[e] + b[e] > 0.f)) d[e] *= slope;
                }
                s += d;
                q += d * xh;
            }
        }
    }
    *reinterpret_cast<f32x4*>(&ps[rl][cq * 4]) = s;
    *reinterpret_cast<f32x4*>(&pq[rl][cq * 4]) = q;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, col = threadIdx.x & 63;
        const int cc = blockIdx.x * 64 + col;
        float (*pp)[68] = which ? pq : ps;
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += (double)pp[i][col];
        if (cc < C) part[((int64_t)blockIdx.y * 2 + which) * C + cc] = t;
    }
}

// column totals of this block's 64 columns from RB partial rows (RB = 1: `part` already holds the totals); two threads per
// (sum, column), the loads of a thread independent of each other
__device__ __forceinline__ void bn_totals(const double* __restrict__ part, int RB, int C, double (*tot)[64]) {
    __shared__ double half2[2][2][64];
    const int which = (threadIdx.x >> 6) & 1, hf = threadIdx.x >> 7, col = threadIdx.x & 63;
    const int cc = blockIdx.x * 64 + col;
    double t = 0.0;
    if (cc < C) {
        const double* pp = part + (int64_t)which * C + cc;
#pragma unroll 8
        for (int rb = hf; rb < RB; rb += 2) t += pp[(int64_t)rb * 2 * C];
    }
    half2[hf][which][col] = t;
    __syncthreads();
    if (threadIdx.x < 128) tot[which][col] = half2[0][which][col] + half2[1][which][col];
    __syncthreads();
}

__global__ __launch_bounds__(256) void bn_finish4_kernel(const double* __restrict__ part, int RB, int C, double* __restrict__ wsd) {
    const int i = blockIdx.x * 256 + threadIdx.x;   // which * C + c
    if (i >= 2 * C) return;
    double t = 0.0;
#pragma unroll 8
    for (int rb = 0; rb < RB; ++rb) t += part[(int64_t)rb * 2 * C + i];
    wsd[i] = t;
}

__global__ __launch_bounds__(256) void bn_apply4_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t R, int C, int RPB,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        float slope, float* __restrict__ stats, float* __restrict__ rmean,
                                                        float* __restrict__ rvar, float momentum, const double* __restrict__ part, int RB,
                                                        const double* __restrict__ rs_dev, int64_t Rs_host, long long* __restrict__ nbt) {
    __shared__ double tot[2][64];
    bn_totals(part, RB, C, tot);
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    if (nbt && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) nbt[0] += 1;   // num_batches_tracked
    if (c >= C) return;
    const double Rs = rs_dev ? rs_dev[0] : (double)Rs_host;
    f32x4 g, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double mean_d = tot[0][cq * 4 + e] / Rs;
        double var_d = tot[1][cq * 4 + e] / Rs - mean_d * mean_d;
        if (var_d < 0.0) var_d = 0.0;
        const float mean = (float)mean_d;
        const float invstd = (float)(1.0 / sqrt(var_d + (double)eps));
        if (blockIdx.y == 0 && rl == 0) {
            stats[c + e] = mean;
            stats[C + c + e] = invstd;
            if (rmean) {
                const double unb = Rs > 1.0 ? var_d * Rs / (Rs - 1.0) : var_d;
                rmean[c + e] = (1.f - momentum) * rmean[c + e] + momentum * mean;
                rvar[c + e] = (1.f - momentum) * rvar[c + e] + momentum * (float)unb;
            }
        }
        g[e] = gamma[c + e] * invstd;
        b[e] = beta[c + e] - mean * g[e];
    }
    const int64_t r0 = (int64_t)blockIdx.y * RPB;
    const int64_t r1 = r0 + RPB < R ? r0 + RPB : R;
#pragma unroll 4
    for (int64_t r = r0 + rl; r < r1; r += 16) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + c) * g + b;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        *reinterpret_cast<f32x4*>(y + r * C + c) = v;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                            int64_t R, int C, int RPB, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float slope, float* __restrict__ stats,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                            const double* __restrict__ part, int RB, const double* __restrict__ wsd_local,
                                                            const double* __restrict__ rs_dev, int64_t Rs_host) {
    __shared__ double tot[2][64];
    bn_totals(part, RB, C, tot);
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    if (c >= C) return;
    const float Rs = rs_dev ? (float)rs_dev[0] : (float)Rs_host;
    const f32x4 mean = *reinterpret_cast<const f32x4*>(stats + c), invstd = *reinterpret_cast<const f32x4*>(stats + C + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 m1, m2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float sdy = (float)tot[0][cq * 4 + e], sdyx = (float)tot[1][cq * 4 + e];
        if (blockIdx.y == 0 && rl == 0) {
            stats[2 * C + c + e] = sdy;
            stats[3 * C + c + e] = sdyx;
            // parameter gradients are THIS rank's sums (the gradient all-reduce adds the ranks up); dx uses the global sums
            const float pdy = wsd_local ? (float)wsd_local[c + e] : sdy, pdyx = wsd_local ? (float)wsd_local[C + c + e] : sdyx;
            if (dgamma) dgamma[c + e] = accumulate ? dgamma[c + e] + pdyx : pdyx;
            if (dbeta) dbeta[c + e] = accumulate ? dbeta[c + e] + pdy : pdy;
        }
        m1[e] = sdy / Rs;
        m2[e] = sdyx / Rs;
    }
    const f32x4 k = g * invstd;
    const int64_t r0 = (int64_t)blockIdx.y * RPB;
    const int64_t r1 = r0 + RPB < R ? r0 + RPB : R;
#pragma unroll 4
    for (int64_t r = r0 + rl; r < r1; r += 16) {
        const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + r * C + c) - mean) * invstd;
        f32x4 d = *reinterpret_cast<const f32x4*>(dy + r * C + c);
        if (slope != 1.f) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!(xh[e] * g[e] + b[e] > 0.f)) d[e] *= slope;
        }
        *reinterpret_cast<f32x4*>(dx + r * C + c) = k * (d - m1 - xh * m2);
    }
}

// geometry of the 16-byte path: 128 rows per block while that gives <= 32 partial rows
struct Geom4 {
    int RB, RPB;
};
inline Geom4 geom4(int64_t R) {
    Geom4 g;
    g.RB = cdiv(R, 128) < 32 ? cdiv(R, 128) : 32;
    g.RPB = (cdiv(R, g.RB) + 15) / 16 * 16;
    g.RB = cdiv(R, g.RPB);
    return g;
}

// -> nullptr when the 16-byte path does not apply (C, alignment) or no workspace can be had (first use under hipGraph capture)
double* part_ws(hipStream_t s, int64_t R, int C, const void* a, const void* b, const void* c, const void* d) {
    if (C % 4 != 0 || ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                        reinterpret_cast<uintptr_t>(d)) & 15) != 0 || (aas_debug_flags_value() & 32768))
        return nullptr;
    const size_t bytes = sizeof(double) * 2 * (size_t)C * (size_t)geom4(R).RB;
    // per (device, stream); an outgrown block is retired, not freed (captured graphs keep its address): common.h
    return static_cast<double*>(aas_stream_workspace(AAS_WS_BN_PARTIALS, s, bytes, (size_t)2 << 20));
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
    if (double* part = part_ws(s, R, C, x, nullptr, nullptr, nullptr)) {
        const Geom4 g = geom4(R);
        hipLaunchKernelGGL(bn_part4_kernel<false>, dim3(cdiv(C, 64), g.RB), dim3(256), 0, s, x, (const float*)nullptr, R, C, g.RPB,
                           (const float*)nullptr, (const float*)nullptr, 1.f, (const float*)nullptr, part);
        hipLaunchKernelGGL(bn_finish4_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, s, part, g.RB, C, wsd);
        AAS_LAUNCH_CHECK("aas_bn_stats");
        return 0;
    }
    AAS_HIP(hipMemsetAsync(wsd, 0, sizeof(double) * 2 * C, s));
    dim3 grid(cdiv(C, 64), cdiv(R, RPB));
    hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, s, x, R, C, wsd);
    AAS_LAUNCH_CHECK("aas_bn_stats");
    return 0;
}

extern "C" int aas_bn_apply(aasStream_t stream, const float* x, float* y, int64_t R, int C, const float* gamma,
                            const float* beta, float eps, float slope, float* stats, float* running_mean,
                            float* running_var, float momentum, const double* wsd, const double* d_rows, long long* num_batches_tracked) {
    AAS_CHECK(x && y && gamma && beta && stats && wsd && R > 0 && C > 0, "aas_bn_apply: bad args");
    AAS_CHECK((running_mean == nullptr) == (running_var == nullptr), "aas_bn_apply: running stats must both be set or both NULL");
    if (part_ws((hipStream_t)stream, R, C, x, y, nullptr, nullptr)) {   // (the totals are the caller's: one "partial" row)
        const Geom4 g = geom4(R);
        hipLaunchKernelGGL(bn_apply4_kernel, dim3(cdiv(C, 64), g.RB), dim3(256), 0, (hipStream_t)stream, x, y, R, C, g.RPB, gamma, beta,
                           eps, slope, stats, running_mean, running_var, momentum, wsd, 1, d_rows, R, num_batches_tracked);
        AAS_LAUNCH_CHECK("aas_bn_apply");
        return 0;
    }
    dim3 grid(cdiv(C, 64), cdiv(R, RPB));
    hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, R, C, gamma, beta, eps, slope, stats,
                       running_mean, running_var, momentum, wsd, d_rows, R, num_batches_tracked);
    AAS_LAUNCH_CHECK("aas_bn_apply");
    return 0;
}

extern "C" int aas_bn_bwd_reduce(aasStream_t stream, const float* x, const float* dy, int64_t R, int C, const float* gamma,
                                 const float* beta, float slope, const float* stats, double* wsd) {
    AAS_CHECK(x && dy && gamma && beta && stats && wsd && R > 0 && C > 0, "aas_bn_bwd_reduce: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (double* part = part_ws(s, R, C, x, dy, nullptr, nullptr)) {
        const Geom4 g = geom4(R);
        hipLaunchKernelGGL(bn_part4_kernel<true>, dim3(cdiv(C, 64), g.RB), dim3(256), 0, s, x, dy, R, C, g.RPB, gamma, beta, slope, stats,
                           part);
        hipLaunchKernelGGL(bn_finish4_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, s, part, g.RB, C, wsd);
        AAS_LAUNCH_CHECK("aas_bn_bwd_reduce");
        return 0;
    }
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
    if (part_ws((hipStream_t)stream, R, C, x, dy, dx, nullptr)) {
        const Geom4 g = geom4(R);
        hipLaunchKernelGGL(bn_bwd_apply4_kernel, dim3(cdiv(C, 64), g.RB), dim3(256), 0, (hipStream_t)stream, x, dy, dx, R, C, g.RPB, gamma,
                           beta, slope, stats, dgamma, dbeta, accumulate, wsd, 1, wsd_local, d_rows, R);
        AAS_LAUNCH_CHECK("aas_bn_bwd_apply");
        return 0;
    }
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
                          float* running_var, float momentum, double* wsd, long long* num_batches_tracked) {
    AAS_CHECK(x && y && gamma && beta && stats && wsd && R > 0 && C > 0, "aas_bn_fwd: bad args");
    AAS_CHECK((running_mean == nullptr) == (running_var == nullptr), "aas_bn_fwd: running stats must both be set or both NULL");
    hipStream_t s = (hipStream_t)stream;
    if (double* part = part_ws(s, R, C, x, y, nullptr, nullptr)) {   // two launches, no fill, no atomics (`wsd` stays untouched)
        const Geom4 g = geom4(R);
        const dim3 grid(cdiv(C, 64), g.RB);
        hipLaunchKernelGGL(bn_part4_kernel<false>, grid, dim3(256), 0, s, x, (const float*)nullptr, R, C, g.RPB, (const float*)nullptr,
                           (const float*)nullptr, 1.f, (const float*)nullptr, part);
        hipLaunchKernelGGL(bn_apply4_kernel, grid, dim3(256), 0, s, x, y, R, C, g.RPB, gamma, beta, eps, slope, stats, running_mean,
                           running_var, momentum, (const double*)part, g.RB, (const double*)nullptr, R, num_batches_tracked);
        AAS_LAUNCH_CHECK("aas_bn_fwd");
        return 0;
    }
    AAS_HIP(hipMemsetAsync(wsd, 0, sizeof(double) * 2 * C, s));
    dim3 grid(cdiv(C, 64), cdiv(R, RPB));
    hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, s, x, R, C, wsd);
    hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(256), 0, s, x, y, R, C, gamma, beta, eps, slope, stats, running_mean,
                       running_var, momentum, wsd, (const double*)nullptr, R, num_batches_tracked);
    AAS_LAUNCH_CHECK("aas_bn_fwd");
    return 0;
}

extern "C" int aas_bn_bwd(aasStream_t stream, const float* x, const float* dy, float* dx, int64_t R, int C,
                          const float* gamma, const float* beta, float slope, float* stats, float* dgamma,
                          float* dbeta, int accumulate, double* wsd) {
    AAS_CHECK(x && dy && dx && gamma && beta && stats && wsd && R > 0 && C > 0, "aas_bn_bwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (double* part = part_ws(s, R, C, x, dy, dx, nullptr)) {
        const Geom4 g = geom4(R);
        const dim3 grid(cdiv(C, 64), g.RB);
        hipLaunchKernelGGL(bn_part4_kernel<true>, grid, dim3(256), 0, s, x, dy, R, C, g.RPB, gamma, beta, slope, (const float*)stats, part);
        hipLaunchKernelGGL(bn_bwd_apply4_kernel, grid, dim3(256), 0, s, x, dy, dx, R, C, g.RPB, gamma, beta, slope, stats, dgamma, dbeta,
                           accumulate, (const double*)part, g.RB, (const double*)nullptr, (const double*)nullptr, R);
        AAS_LAUNCH_CHECK("aas_bn_bwd");
        return 0;
    }
    AAS_HIP(hipMemsetAsync(wsd, 0, sizeof(double) * 2 * C, s));
    dim3 grid(cdiv(C, 64), cdiv(R, RPB));
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(256), 0, s, x, dy, R, C, gamma, beta, slope, stats, wsd);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, s, x, dy, dx, R, C, gamma, beta, slope, stats, dgamma,
                       dbeta, accumulate, wsd, (const double*)nullptr, (const double*)nullptr, R);
    AAS_LAUNCH_CHECK("aas_bn_bwd");
    return 0;
}
