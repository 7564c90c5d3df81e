// HBM-bound elementwise / layout / reduction kernels (gfx950): 16-byte vector accesses, 64-wide
// wave reductions, grid-stride loops capped at ~8 blocks per CU.
#include "common.h"

namespace {

constexpr int MAXB = 2048;

// ---- tiled transpose: out[b, c, r] = in[b, r, c] ------------------------------------------------
template <bool ACC>
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        int R, int C, int64_t isb, int64_t isr, int64_t osb,
                                                        int64_t osc) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    const float* ip = in + (int64_t)b * isb;
    float* op = out + (int64_t)b * osb;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int r = r0 + ty + i * 4, c = c0 + tx;
        if (r < R && c < C) tile[ty + i * 4][tx] = ip[(int64_t)r * isr + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int c = c0 + ty + i * 4, r = r0 + tx;
        if (r < R && c < C) {
            if (ACC) op[(int64_t)c * osc + r] += tile[tx][ty + i * 4];
            else op[(int64_t)c * osc + r] = tile[tx][ty + i * 4];
        }
    }
}

__global__ __launch_bounds__(256) void add3_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                   const float* __restrict__ b, const float* __restrict__ c,
                                                   int64_t n4, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n4; i += stride) {
        f32x4 va = reinterpret_cast<const f32x4*>(a)[i];
        f32x4 vb = reinterpret_cast<const f32x4*>(b)[i];
        f32x4 v = va + vb;
        if (c) v += reinterpret_cast<const f32x4*>(c)[i];
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
    // scalar tail (n not a multiple of 4, or unaligned pointers: n4 == 0)
    for (int64_t t = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += stride)
        out[t] = a[t] + b[t] + (c ? c[t] : 0.f);
}

__global__ __launch_bounds__(256) void axpby_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha,
                                                    float beta, int64_t n, const float* __restrict__ d_alpha) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (d_alpha) alpha *= d_alpha[0];
    for (; i < n; i += stride) {
        float v = alpha * x[i];
        if (beta != 0.f) v += beta * y[i];
        y[i] = v;
    }
}

// out[c] += sum over a chunk of rows; grid (col blocks of 64, row chunks)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t R, int C, int64_t ld,
                                                     float* __restrict__ out, int rows_per_block) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    float s = 0.f;
    if (c < C)
        for (int64_t r = r0 + w; r < r1; r += 4) s += x[r * ld + c];
    part[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < C) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void sqsum_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ acc) {
    __shared__ double part[4];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float s = 0.f;
    for (; i < n; i += stride) { float v = x[i]; s += v * v; }
    double d = wave_sum_d((double)s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void l1_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     int64_t n, double* __restrict__ acc) {
    __shared__ double part[4];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float s = 0.f;
    for (; i < n; i += stride) s += fabsf(a[i] - b[i]);
    double d = wave_sum_d((double)s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     int64_t n, float scale, const float* __restrict__ d_scale,
                                                     float* __restrict__ ga, float* __restrict__ gb, int accumulate) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (d_scale) scale *= d_scale[0];
    for (; i < n; i += stride) {
        float d = a[i] - b[i];
        float g = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);  // torch.abs backward: sign(d), 0 at 0
        if (ga) ga[i] = accumulate ? ga[i] + g : g;
        if (gb) gb[i] = accumulate ? gb[i] - g : -g;
    }
}

// Row-strided forms of the two L1 kernels: `rows` rows of `cols` valid elements, a / b / ga / gb with their own row pitches - the
// loss of one utterance class of a batched pass over a ragged pair (each class's own frames of a tensor padded to the longer class's
// length).  The backward kernel also ZEROES columns [cols, ga_cols) of ga: no gradient reaches the padding.
__global__ __launch_bounds__(256) void l1_fwd2d_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                       int64_t rows, int cols, double* __restrict__ acc) {
    __shared__ double part[4];
    const int64_t n = rows * cols;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float s = 0.f;
    for (; i < n; i += stride) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        s += fabsf(a[r * lda + c] - b[r * ldb + c]);
    }
    double d = wave_sum_d((double)s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void l1_bwd2d_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                       int64_t rows, int cols, float scale, const float* __restrict__ d_scale,
                                                       float* __restrict__ ga, int64_t ldga, int ga_cols, float* __restrict__ gb, int64_t ldgb) {
    const int width = ga ? (ga_cols > cols ? ga_cols : cols) : cols;
    const int64_t n = rows * width;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (d_scale) scale *= d_scale[0];
    for (; i < n; i += stride) {
        const int64_t r = i / width;
        const int c = (int)(i - r * width);
        if (c < cols) {
            const float d = a[r * lda + c] - b[r * ldb + c];
            const float g = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
            if (ga) ga[r * ldga + c] = g;
            if (gb) gb[r * ldgb + c] = -g;
        } else {
            ga[r * ldga + c] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ vmax, int64_t n, float b1, float b2, float eps,
                                                   float step_size, float bc2_sqrt, int amsgrad, float gscale,
                                                   const float* __restrict__ d_hyper) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (d_hyper) {  // step-dependent scalars kept on the device (hipGraph replay: no host-side bias correction)
        step_size = d_hyper[0];
        bc2_sqrt = d_hyper[1];
    }
    for (; i < n; i += stride) {
        float gi = g[i] * gscale;
        float mo = m[i];
        float mi = mo + (1.f - b1) * (gi - mo);  // torch: exp_avg.lerp_(grad, 1-beta1)
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        float vv = vi;
        if (amsgrad) {
            vv = fmaxf(vmax[i], vi);
            vmax[i] = vv;
        }
        float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] -= step_size * (mi / denom);
    }
}

// torch.optim.SGD(momentum, nesterov=True, dampening 0): buf = mu buf + g (buf starts at zero, which gives the first step's
// "buf = g" exactly); p -= lr (g + mu buf)
__global__ __launch_bounds__(256) void sgd_nesterov_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                           int64_t n, float lr, float mu, float gscale) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const float gi = g[i] * gscale;
        const float b = mu * buf[i] + gi;
        buf[i] = b;
        p[i] -= lr * (gi + mu * b);
    }
}

// dx[n,t,f] = sum_kk dcol[n, (t-kk)/s, kk, f]
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int N,
                                                     int T, int T1, int F, int KW, int s) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tot = (int64_t)N * T * F;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < tot; i += stride) {
        int f = (int)(i % F);
        int t = (int)((i / F) % T);
        int n = (int)(i / ((int64_t)F * T));
        float acc = 0.f;
        for (int kk = 0; kk < KW; ++kk) {
            int tt = t - kk;
            if (tt < 0) break;
            if (tt % s) continue;
            int t1 = tt / s;
            if (t1 < T1) acc += dcol[(((int64_t)n * T1 + t1) * KW + kk) * F + f];
        }
        dx[i] = acc;
    }
}

__global__ __launch_bounds__(256) void swap01_kernel(const float* __restrict__ in, float* __restrict__ out, int A, int B,
                                                     int C) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tot = (int64_t)A * B * C;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < tot; i += stride) {
        const int c = (int)(i % C);
        const int b = (int)((i / C) % B);
        const int a = (int)(i / ((int64_t)C * B));
        out[((int64_t)b * A + a) * C + c] = in[i];
    }
}

// out[(t,n), c] = in[(t,n), c] * scale[n]   (rows are time-major: row = t*Nb + n)
__global__ __launch_bounds__(256) void scale_rows_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                         const float* __restrict__ scale, int64_t rows, int Nb, int C) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tot = rows * C;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < tot; i += stride) {
        const int64_t r = i / C;
        out[i] = in[i] * scale[(int)(r % Nb)];
    }
}

inline int grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > MAXB ? MAXB : b));
}

}  // namespace

extern "C" int aas_transpose_f32(aasStream_t stream, const float* in, float* out, int B, int R, int C, int64_t isb,
                                 int64_t isr, int64_t osb, int64_t osc) {
    AAS_CHECK(in && out && B > 0 && R > 0 && C > 0, "aas_transpose_f32: bad args");
    AAS_CHECK(B <= 65535 && cdiv(R, 64) <= 65535, "aas_transpose_f32: grid too large");
    dim3 grid(cdiv(C, 64), cdiv(R, 64), B);
    hipLaunchKernelGGL(transpose_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in, out, R, C, isb, isr, osb, osc);
    AAS_LAUNCH_CHECK("aas_transpose_f32");
    return 0;
}

extern "C" int aas_transpose_add_f32(aasStream_t stream, const float* in, float* out, int B, int R, int C, int64_t isb,
                                     int64_t isr, int64_t osb, int64_t osc) {
    AAS_CHECK(in && out && B > 0 && R > 0 && C > 0, "aas_transpose_add_f32: bad args");
    AAS_CHECK(B <= 65535 && cdiv(R, 64) <= 65535, "aas_transpose_add_f32: grid too large");
    dim3 grid(cdiv(C, 64), cdiv(R, 64), B);
    hipLaunchKernelGGL(transpose_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in, out, R, C, isb, isr, osb, osc);
    AAS_LAUNCH_CHECK("aas_transpose_add_f32");
    return 0;
}

extern "C" int aas_swap01_f32(aasStream_t stream, const float* in, float* out, int A, int B, int C) {
    AAS_CHECK(in && out && A > 0 && B > 0 && C > 0, "aas_swap01_f32: bad args");
    hipLaunchKernelGGL(swap01_kernel, dim3(grid_for((int64_t)A * B * C)), dim3(256), 0, (hipStream_t)stream, in, out, A, B, C);
    AAS_LAUNCH_CHECK("aas_swap01_f32");
    return 0;
}

extern "C" int aas_scale_rows_f32(aasStream_t stream, float* out, const float* in, const float* scale, int64_t rows,
                                  int Nb, int C) {
    AAS_CHECK(out && in && scale && rows >= 0 && Nb > 0 && C > 0, "aas_scale_rows_f32: bad args");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, out, in, scale, rows, Nb, C);
    AAS_LAUNCH_CHECK("aas_scale_rows_f32");
    return 0;
}

extern "C" int aas_add3_f32(aasStream_t stream, float* out, const float* a, const float* b, const float* c, int64_t n) {
    AAS_CHECK(out && a && b && n >= 0, "aas_add3_f32: bad args");
    if (n == 0) return 0;
    bool al = ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) |
                reinterpret_cast<uintptr_t>(c)) & 15) == 0;
    int64_t n4 = al ? n / 4 : 0;
    int g = grid_for(n4 > 0 ? n4 : n);
    hipLaunchKernelGGL(add3_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, out, a, b, c, n4, n);
    AAS_LAUNCH_CHECK("aas_add3_f32");
    return 0;
}

extern "C" int aas_axpby_f32(aasStream_t stream, float* y, const float* x, float alpha, float beta, int64_t n) {
    AAS_CHECK(y && x && n >= 0, "aas_axpby_f32: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, x, alpha, beta, n, (const float*)nullptr);
    AAS_LAUNCH_CHECK("aas_axpby_f32");
    return 0;
}

// y = (ref >= 0 ? 1 : slope) * x: the forward pass with ref == x, the backward pass with x = dy and ref = the forward input
__global__ __launch_bounds__(256) void leaky_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ ref,
                                                    float slope, int64_t n4, int64_t n) {
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = i0; i < n4; i += st) {
        const float4 a = reinterpret_cast<const float4*>(x)[i], r = reinterpret_cast<const float4*>(ref)[i];
        float4 o;
        o.x = r.x >= 0.f ? a.x : slope * a.x;
        o.y = r.y >= 0.f ? a.y : slope * a.y;
        o.z = r.z >= 0.f ? a.z : slope * a.z;
        o.w = r.w >= 0.f ? a.w : slope * a.w;
        reinterpret_cast<float4*>(y)[i] = o;
    }
    for (int64_t i = 4 * n4 + i0; i < n; i += st) y[i] = ref[i] >= 0.f ? x[i] : slope * x[i];
}

extern "C" int aas_leaky_relu_f32(aasStream_t stream, float* y, const float* x, const float* ref, float slope, int64_t n) {
    AAS_CHECK(y && x && ref && n >= 0, "aas_leaky_relu_f32: bad args");
    if (n == 0) return 0;
    const bool al = ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ref)) & 15) == 0;
    const int64_t n4 = al ? n / 4 : 0;
    hipLaunchKernelGGL(leaky_kernel, dim3(grid_for(n4 > 0 ? n4 : n)), dim3(256), 0, (hipStream_t)stream, y, x, ref, slope, n4, n);
    AAS_LAUNCH_CHECK("aas_leaky_relu_f32");
    return 0;
}

extern "C" int aas_scale_dev_f32(aasStream_t stream, float* y, const float* x, const float* d_alpha, float alpha, int64_t n) {
    AAS_CHECK(y && x && d_alpha && n >= 0, "aas_scale_dev_f32: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, x, alpha, 0.f, n, d_alpha);
    AAS_LAUNCH_CHECK("aas_scale_dev_f32");
    return 0;
}

extern "C" int aas_colsum_f32(aasStream_t stream, const float* x, int64_t R, int C, int64_t ld, float* out,
                              int accumulate) {
    AAS_CHECK(x && out && R >= 0 && C > 0, "aas_colsum_f32: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) AAS_HIP(hipMemsetAsync(out, 0, sizeof(float) * C, s));
    if (R == 0) return 0;
    int rpb = 256;
    dim3 grid(cdiv(C, 64), cdiv(R, rpb));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, x, R, C, ld, out, rpb);
    AAS_LAUNCH_CHECK("aas_colsum_f32");
    return 0;
}

extern "C" int aas_sqsum_f32(aasStream_t stream, const float* x, int64_t n, double* acc) {
    AAS_CHECK(x && acc && n >= 0, "aas_sqsum_f32: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(sqsum_kernel, dim3(grid_for(n) > 512 ? 512 : grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, acc);
    AAS_LAUNCH_CHECK("aas_sqsum_f32");
    return 0;
}

extern "C" int aas_l1_fwd(aasStream_t stream, const float* a, const float* b, int64_t n, double* loss_sum) {
    AAS_CHECK(a && b && loss_sum && n >= 0, "aas_l1_fwd: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(l1_fwd_kernel, dim3(grid_for(n) > 512 ? 512 : grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, loss_sum);
    AAS_LAUNCH_CHECK("aas_l1_fwd");
    return 0;
}

extern "C" int aas_l1_bwd(aasStream_t stream, const float* a, const float* b, int64_t n, float scale,
                          const float* d_scale, float* ga, float* gb, int accumulate) {
    AAS_CHECK(a && b && n >= 0, "aas_l1_bwd: bad args");
    if (n == 0 || (!ga && !gb)) return 0;
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, scale, d_scale, ga, gb, accumulate);
    AAS_LAUNCH_CHECK("aas_l1_bwd");
    return 0;
}

extern "C" int aas_l1_fwd2d(aasStream_t stream, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int cols,
                            double* loss_sum) {
    AAS_CHECK(a && b && loss_sum && rows >= 0 && cols >= 0 && lda >= cols && ldb >= cols, "aas_l1_fwd2d: bad args");
    if (rows == 0 || cols == 0) return 0;
    const int g = grid_for(rows * cols);
    hipLaunchKernelGGL(l1_fwd2d_kernel, dim3(g > 512 ? 512 : g), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, rows, cols, loss_sum);
    AAS_LAUNCH_CHECK("aas_l1_fwd2d");
    return 0;
}

extern "C" int aas_l1_bwd2d(aasStream_t stream, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int cols, float scale,
                            const float* d_scale, float* ga, int64_t ldga, int ga_cols, float* gb, int64_t ldgb) {
    AAS_CHECK(a && b && rows >= 0 && cols >= 0 && lda >= cols && ldb >= cols, "aas_l1_bwd2d: bad args");
    AAS_CHECK(!ga || (ldga >= cols && ldga >= ga_cols), "aas_l1_bwd2d: ga pitch");
    AAS_CHECK(!gb || ldgb >= cols, "aas_l1_bwd2d: gb pitch");
    if (rows == 0 || (!ga && !gb)) return 0;
    const int width = ga ? (ga_cols > cols ? ga_cols : cols) : cols;
    if (width == 0) return 0;
    hipLaunchKernelGGL(l1_bwd2d_kernel, dim3(grid_for(rows * width)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, rows, cols, scale, d_scale,
                       ga, ldga, ga_cols, gb, ldgb);
    AAS_LAUNCH_CHECK("aas_l1_bwd2d");
    return 0;
}

extern "C" int aas_adam_f32(aasStream_t stream, float* p, const float* g, float* m, float* v, float* vmax, int64_t n,
                            float lr, float beta1, float beta2, float eps, int step, int amsgrad, float grad_scale) {
    AAS_CHECK(p && g && m && v && (vmax || !amsgrad) && n >= 0 && step >= 1, "aas_adam_f32: bad args");
    if (n == 0) return 0;
    double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    float step_size = (float)(lr / bc1);
    float bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vmax, n, beta1,
                       beta2, eps, step_size, bc2_sqrt, amsgrad, grad_scale, (const float*)nullptr);
    AAS_LAUNCH_CHECK("aas_adam_f32");
    return 0;
}

extern "C" int aas_sgd_nesterov_f32(aasStream_t stream, float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                                    float grad_scale) {
    AAS_CHECK(p && g && buf && n >= 0, "aas_sgd_nesterov_f32: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, lr, momentum, grad_scale);
    AAS_LAUNCH_CHECK("aas_sgd_nesterov_f32");
    return 0;
}

namespace {
// one thread: advance the device-resident step counter and derive the step-dependent Adam scalars from it
__global__ void adam_tick_kernel(double* __restrict__ t, double lr, double b1, double b2, float* __restrict__ hyper) {
    const double tt = t[0] + 1.0;
    t[0] = tt;
    hyper[0] = (float)(lr / (1.0 - pow(b1, tt)));
    hyper[1] = (float)sqrt(1.0 - pow(b2, tt));
}

// BEGAN proportional controller on the device (trainer_AAS.py:190-194) + the scalars a later log line needs:
// out = [l_adv_ny_G, l_adv_cl, l_ctc, kt, running sum of l_ctc * n, running sum of n]
__global__ void began_step_kernel(const float* __restrict__ l_ny, const float* __restrict__ l_cl, const float* __restrict__ l_ctc,
                                  double* __restrict__ kt, double* __restrict__ out, double gamma, double lambda_k, double n_batch) {
    const double a = (double)l_ny[0], b = (double)l_cl[0], c = (double)l_ctc[0];
    double k = kt[0] + lambda_k * (gamma * b - a);
    k = k < 0.0 ? 0.0 : (k > 1.0 ? 1.0 : k);
    kt[0] = k;
    out[0] = a; out[1] = b; out[2] = c; out[3] = k;
    out[4] += c * n_batch;
    out[5] += n_batch;
}
// ---- step prologue: every buffer of the step that starts from zero, and the per-utterance weights of the batched discriminator
// pass, in ONE launch (was: one fill launch per gradient buffer / loss accumulator, neg + two copies + fill for the weights)
struct ZeroList {
    void* p[8];
    unsigned long long bytes[8];        // multiples of 16 except possibly the last chunk of each buffer (handled bytewise)
    unsigned long long first[9];        // first 4 KB chunk of each buffer in the launch's chunk numbering
    int n;
};
__global__ __launch_bounds__(256) void step_prologue_kernel(ZeroList z, float* __restrict__ rs, int n_neg, int n_one,
                                                            const double* __restrict__ kt) {
    const unsigned long long total = z.first[z.n];
    for (unsigned long long c = blockIdx.x; c < total; c += gridDim.x) {
        int b = 0;
#pragma unroll
        for (int i = 1; i < 8; ++i)
            if (i < z.n && c >= z.first[i]) b = i;
        const unsigned long long off = (c - z.first[b]) * 4096ull + (unsigned long long)threadIdx.x * 16ull;
        char* q = static_cast<char*>(z.p[b]) + off;
        if (off + 16 <= z.bytes[b]) {
            *reinterpret_cast<f32x4*>(q) = (f32x4){0.f, 0.f, 0.f, 0.f};
        } else if (off < z.bytes[b]) {
            for (unsigned long long i = off; i < z.bytes[b]; ++i) static_cast<char*>(z.p[b])[i] = 0;
        }
    }
    if (rs != nullptr && blockIdx.x == 0) {
        const float neg = -(float)kt[0];
        for (int i = threadIdx.x; i < n_neg + n_one; i += 256) rs[i] = i < n_neg ? neg : 1.0f;
    }
}

// BEGAN controller from the RAW sums of the step (no scaling / summing launches in between): L_ny = s_ny * sum|D(E(x)) - E(x)|,
// L_cl = s_cl * sum|D(c) - c|, L_ctc = s_ctc * sum of the per-utterance CTC costs
__global__ void began_step_raw_kernel(const double* __restrict__ l1_sums, double s_ny, double s_cl, const float* __restrict__ costs, int n_costs,
                                      double s_ctc, double* __restrict__ kt, double* __restrict__ out, double gamma, double lambda_k,
                                      double n_batch) {
    float cs = 0.f;                                   // (fp32 running sum in utterance order: what the colsum launch did)
    for (int i = 0; i < n_costs; ++i) cs += costs[i];
    const double a = (double)(float)(l1_sums[0] * s_ny), b = (double)(float)(l1_sums[1] * s_cl), c = (double)(float)((double)cs * s_ctc);
    double k = kt[0] + lambda_k * (gamma * b - a);
    k = k < 0.0 ? 0.0 : (k > 1.0 ? 1.0 : k);
    kt[0] = k;
    out[0] = a; out[1] = b; out[2] = c; out[3] = k;
    out[4] += c * n_batch;
    out[5] += n_batch;
}
// The controller from THREE raw device sums - sums[0..1] and third[0] - (FSEGAN: the third logged loss is the DCE sum, not CTC costs; data parallel: the sums are
// all-reduced first and the normalisers are device scalars): L_i = s_i * (d_scales ? d_scales[i] : 1) * sums[i]
__global__ void began_step_sums_kernel(const double* __restrict__ sums, const double* __restrict__ third, double s0, double s1, double s2,
                                       const float* __restrict__ d_scales,
                                       double* __restrict__ kt, double* __restrict__ out, double gamma, double lambda_k, double n_batch,
                                       const double* __restrict__ d_n_batch) {
    const double f0 = d_scales ? (double)d_scales[0] : 1.0, f1 = d_scales ? (double)d_scales[1] : 1.0, f2 = d_scales ? (double)d_scales[2] : 1.0;
    const double a = (double)(float)(sums[0] * s0 * f0), b = (double)(float)(sums[1] * s1 * f1), c = (double)(float)(third[0] * s2 * f2);
    double k = kt[0] + lambda_k * (gamma * b - a);
    k = k < 0.0 ? 0.0 : (k > 1.0 ? 1.0 : k);
    kt[0] = k;
    const double nb = d_n_batch ? d_n_batch[0] : n_batch;
    out[0] = a; out[1] = b; out[2] = c; out[3] = k;
    out[4] += c * nb;
    out[5] += nb;
}
// [sum|D(E(x)) - E(x)|, sum|D(c) - c|, sum of the per-utterance CTC costs] as three doubles: what a data-parallel step all-reduces
__global__ void loss_pack_kernel(const double* __restrict__ l1_sums, const float* __restrict__ costs, int n_costs, double* __restrict__ out3) {
    float cs = 0.f;                                   // (fp32 running sum in utterance order, as began_step_raw_kernel)
    for (int i = 0; i < n_costs; ++i) cs += costs[i];
    out3[0] = l1_sums ? l1_sums[0] : 0.0;
    out3[1] = l1_sums ? l1_sums[1] : 0.0;
    out3[2] = (double)cs;
}
__global__ void sums_pack_kernel(const double* __restrict__ a2, const double* __restrict__ b1, double* __restrict__ out3) {
    out3[0] = a2 ? a2[0] : 0.0;
    out3[1] = a2 ? a2[1] : 0.0;
    out3[2] = b1 ? b1[0] : 0.0;
}
__global__ void scales_from_counts_kernel(const double* __restrict__ counts, int n, double w0, double w1, double w2, double w3, int i0, int i1,
                                          int i2, int i3, float* __restrict__ out) {
    const double w[4] = {w0, w1, w2, w3};
    const int idx[4] = {i0, i1, i2, i3};
    for (int i = 0; i < n; ++i) out[i] = (float)(w[i] / counts[idx[i]]);
}
}  // namespace

extern "C" int aas_step_prologue(aasStream_t stream, int n, void* const* bufs, const size_t* bytes, float* rs, int n_neg, int n_one,
                                 const double* d_kt) {
    AAS_CHECK(n >= 0 && n <= 8 && (n == 0 || (bufs && bytes)), "aas_step_prologue: at most 8 buffers");
    AAS_CHECK(rs == nullptr || (d_kt != nullptr && n_neg >= 0 && n_one >= 0), "aas_step_prologue: the weight vector needs d_kt");
    ZeroList z;
    memset(&z, 0, sizeof(z));
    z.n = n;
    unsigned long long chunks = 0;
    for (int i = 0; i < n; ++i) {
        AAS_CHECK(bufs[i] != nullptr && (reinterpret_cast<uintptr_t>(bufs[i]) & 15) == 0, "aas_step_prologue: buffer %d is null or not 16-byte aligned", i);
        z.p[i] = bufs[i];
        z.bytes[i] = bytes[i];
        z.first[i] = chunks;
        chunks += (bytes[i] + 4095) / 4096;
    }
    z.first[n] = chunks;
    if (chunks == 0 && rs == nullptr) return 0;
    const unsigned long long g = chunks < 1 ? 1 : (chunks > 8192 ? 8192 : chunks);
    hipLaunchKernelGGL(step_prologue_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, z, rs, n_neg, n_one, d_kt);
    AAS_LAUNCH_CHECK("aas_step_prologue");
    return 0;
}

extern "C" int aas_began_step_raw(aasStream_t stream, const double* d_l1_sums, double scale_ny, double scale_cl, const float* d_ctc_costs,
                                  int n_costs, double scale_ctc, double* d_kt, double* d_out6, double gamma, double lambda_k, double n_batch) {
    AAS_CHECK(d_l1_sums && d_ctc_costs && n_costs >= 0 && d_kt && d_out6, "aas_began_step_raw: null pointer");
    hipLaunchKernelGGL(began_step_raw_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_l1_sums, scale_ny, scale_cl, d_ctc_costs, n_costs,
                       scale_ctc, d_kt, d_out6, gamma, lambda_k, n_batch);
    AAS_LAUNCH_CHECK("aas_began_step_raw");
    return 0;
}

extern "C" int aas_began_step_sums(aasStream_t stream, const double* d_l1_sums, const double* d_third, double s0, double s1, double s2,
                                   const float* d_scales3, double* d_kt, double* d_out6, double gamma, double lambda_k, double n_batch,
                                   const double* d_n_batch) {
    AAS_CHECK(d_l1_sums && d_third && d_kt && d_out6, "aas_began_step_sums: null pointer");
    hipLaunchKernelGGL(began_step_sums_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_l1_sums, d_third, s0, s1, s2, d_scales3, d_kt, d_out6, gamma, lambda_k,
                       n_batch, d_n_batch);
    AAS_LAUNCH_CHECK("aas_began_step_sums");
    return 0;
}

extern "C" int aas_loss_pack(aasStream_t stream, const double* d_l1_sums, const float* d_ctc_costs, int n_costs, double* d_out3) {
    AAS_CHECK(d_out3 && n_costs >= 0 && (n_costs == 0 || d_ctc_costs), "aas_loss_pack: null pointer");
    hipLaunchKernelGGL(loss_pack_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_l1_sums, d_ctc_costs, n_costs, d_out3);
    AAS_LAUNCH_CHECK("aas_loss_pack");
    return 0;
}

extern "C" int aas_sums_pack(aasStream_t stream, const double* d_a2, const double* d_b1, double* d_out3) {
    AAS_CHECK(d_out3 && (d_a2 || d_b1), "aas_sums_pack: null pointer");
    hipLaunchKernelGGL(sums_pack_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_a2, d_b1, d_out3);
    AAS_LAUNCH_CHECK("aas_sums_pack");
    return 0;
}

extern "C" int aas_scales_from_counts(aasStream_t stream, const double* d_counts, int n, const double* weights, const int* index, float* d_out) {
    AAS_CHECK(d_counts && weights && index && d_out && n >= 1 && n <= 4, "aas_scales_from_counts: 1..4 scales");
    double w[4] = {0, 0, 0, 0};
    int ix[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        AAS_CHECK(index[i] >= 0, "aas_scales_from_counts: negative index");
        w[i] = weights[i];
        ix[i] = index[i];
    }
    hipLaunchKernelGGL(scales_from_counts_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_counts, n, w[0], w[1], w[2], w[3], ix[0], ix[1], ix[2],
                       ix[3], d_out);
    AAS_LAUNCH_CHECK("aas_scales_from_counts");
    return 0;
}

extern "C" int aas_adam_tick(aasStream_t stream, double* d_step, double lr, double beta1, double beta2, float* d_hyper) {
    AAS_CHECK(d_step && d_hyper, "aas_adam_tick: null pointer");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_step, lr, beta1, beta2, d_hyper);
    AAS_LAUNCH_CHECK("aas_adam_tick");
    return 0;
}

extern "C" int aas_began_step(aasStream_t stream, const float* d_l_adv_ny_G, const float* d_l_adv_cl, const float* d_l_ctc,
                              double* d_kt, double* d_out6, double gamma, double lambda_k, double n_batch) {
    AAS_CHECK(d_l_adv_ny_G && d_l_adv_cl && d_l_ctc && d_kt && d_out6, "aas_began_step: null pointer");
    hipLaunchKernelGGL(began_step_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_l_adv_ny_G, d_l_adv_cl, d_l_ctc, d_kt, d_out6, gamma,
                       lambda_k, n_batch);
    AAS_LAUNCH_CHECK("aas_began_step");
    return 0;
}

extern "C" int aas_adam_dev_f32(aasStream_t stream, float* p, const float* g, float* m, float* v, float* vmax, int64_t n,
                                float beta1, float beta2, float eps, const float* d_hyper, int amsgrad, float grad_scale) {
    AAS_CHECK(p && g && m && v && (vmax || !amsgrad) && d_hyper && n >= 0, "aas_adam_dev_f32: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vmax, n, beta1,
                       beta2, eps, 0.f, 1.f, amsgrad, grad_scale, d_hyper);
    AAS_LAUNCH_CHECK("aas_adam_dev_f32");
    return 0;
}

extern "C" int aas_col2im_f32(aasStream_t stream, const float* dcol, float* dx, int N, int T, int T1, int F, int KW,
                              int stride) {
    AAS_CHECK(dcol && dx && N > 0 && T > 0 && T1 > 0 && F > 0 && KW > 0 && stride > 0, "aas_col2im_f32: bad args");
    int64_t tot = (int64_t)N * T * F;
    hipLaunchKernelGGL(col2im_kernel, dim3(grid_for(tot)), dim3(256), 0, (hipStream_t)stream, dcol, dx, N, T, T1, F, KW, stride);
    AAS_LAUNCH_CHECK("aas_col2im_f32");
    return 0;
}
