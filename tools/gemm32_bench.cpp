// Stand-alone check + timing of aas_gemm_f32 / aas_gemm_f32_multi (fp32 arithmetic): the LDS-DMA kernel (variant 0) against the
// register-staged one (variant 1) and against an fp64 reference, on the shapes of the AAS step and on edge shapes.
//   hipcc --offload-arch=gfx950 -O2 tools/gemm32_bench.cpp -Iinclude -Laas_enhancement_amd/lib -laas_hip -o /tmp/gemm32_bench
//   LD_LIBRARY_PATH=aas_enhancement_amd/lib /tmp/gemm32_bench [check|time|all]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "aas_hip.h"

#define HIPC(x)                                                                         \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(2);                                                                    \
        }                                                                               \
    } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed * 40503u + 12345u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f;
}

__device__ __forceinline__ long long krow64(int r, int kdiv, long long kouter, long long ld) {
    return kdiv > 0 ? (long long)(r / kdiv) * kouter + (long long)(r % kdiv) * ld : (long long)r * ld;
}

// reference: one thread per output element, fp64 accumulation; also returns sum |a b| for the error scale
__global__ void ref_kernel(int mode, int M, int N, int K, const float* A, long long lda, const float* B, long long ldb, double* R, double* S,
                           int kdivA, long long kouterA, int kdivB, long long kouterB) {
    int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    double acc = 0, sab = 0;
    for (int k = 0; k < K; ++k) {
        float a = mode == 2 ? A[krow64(k, kdivA, kouterA, lda) + m] : A[(long long)m * lda + k];
        float b = mode == 0 ? B[(long long)n * ldb + k] : B[krow64(k, kdivB, kouterB, ldb) + n];
        acc += (double)a * b;
        sab += fabs((double)a * b);
    }
    R[(long long)m * N + n] = acc;
    S[(long long)m * N + n] = sab;
}

__global__ void cmp_kernel(int M, int N, const float* C, long long ldc, const double* R, const double* S, const float* C0, const float* bias,
                           const float* addend, long long ldd, double* out) {
    int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    double want = R[(long long)m * N + n];
    if (C0) want += C0[(long long)m * ldc + n];
    if (bias) want += bias[n];
    if (addend) want += addend[(long long)m * ldd + n];
    double err = fabs((double)C[(long long)m * ldc + n] - want) / (S[(long long)m * N + n] + 1e-30);
    // max via atomic on the bit pattern of a non-negative double
    atomicMax((unsigned long long*)out, (unsigned long long)__double_as_longlong(err));
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// the box's fp32 MFMA ceiling: back-to-back v_mfma_f32_32x32x2_f32 on four accumulators, `wps` waves per SIMD on every CU
__global__ __launch_bounds__(512) void mfma_peak_kernel(float* out, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 123.456f) out[0] = s;
}

static void mfma_peak() {
    float* out;
    HIPC(hipMalloc(&out, 64));
    for (int threads = 256; threads <= 512; threads += 256) {
        const int iters = 20000;
        mfma_peak_kernel<<<256, threads>>>(out, 100);
        HIPC(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
        HIPC(hipEventRecord(e0, nullptr));
        mfma_peak_kernel<<<256, threads>>>(out, iters);
        HIPC(hipEventRecord(e1, nullptr));
        HIPC(hipEventSynchronize(e1));
        float ms;
        HIPC(hipEventElapsedTime(&ms, e0, e1));
        double fl = 256.0 * (threads / 64) * iters * 4 * 4096.0;
        printf("fp32 MFMA peak (32x32x2, %d waves per SIMD, 256 workgroups): %.1f TFLOP/s over %.2f ms\n", threads / 256, fl / ms / 1e9, ms);
    }
    HIPC(hipFree(out));
}

struct Case {
    const char* name;
    int mode, M, N, K;
    long long lda, ldb, ldc;
    int bias, addend, accumulate, batch;
    long long sA, sB, sC;
    int kdivA;
    long long kouterA;
    int kdivB;
    long long kouterB;
    size_t a_elems, b_elems, c_elems;
};

static float* dalloc(size_t n, unsigned seed) {
    float* p;
    HIPC(hipMalloc(&p, n * sizeof(float) + 256));
    fill_kernel<<<(unsigned)((n + 255) / 256), 256>>>(p, n, seed);
    return p;
}

static size_t opnd_elems(bool kc, int rows, int K, long long ld, int kdiv, long long kouter) {
    if (kc) return (size_t)(rows - 1) * ld + K;
    // row-contiguous: K rows of `rows` columns
    long long last = kdiv > 0 ? (long long)((K - 1) / kdiv) * kouter + (long long)((K - 1) % kdiv) * ld : (long long)(K - 1) * ld;
    return (size_t)last + rows;
}

static int run(const Case& c, int variant, float* A, float* B, float* C, float* bias, float* addend) {
    aas_set_gemm_variant(variant);
    return aas_gemm_f32(nullptr, c.mode, c.M, c.N, c.K, A, c.lda, B, c.ldb, C, c.ldc, c.bias ? bias : nullptr, c.addend ? addend : nullptr,
                        c.ldc, c.accumulate, c.batch, c.sA, c.sB, c.sC, c.kdivA, c.kouterA, c.kdivB, c.kouterB);
}

static double check_case(Case c, int variant) {
    const bool akc = c.mode != 2, bkc = c.mode == 0;
    size_t ae = opnd_elems(akc, c.M, c.K, c.lda, c.kdivA, c.kouterA) + (size_t)(c.batch - 1) * c.sA;
    size_t be = opnd_elems(bkc, c.N, c.K, c.ldb, c.kdivB, c.kouterB) + (size_t)(c.batch - 1) * c.sB;
    size_t ce = (size_t)(c.M - 1) * c.ldc + c.N + (size_t)(c.batch - 1) * c.sC;
    float *A = dalloc(ae, 1), *B = dalloc(be, 2), *C = dalloc(ce, 3), *C0 = dalloc(ce, 3), *bias = dalloc(c.N, 4), *add = dalloc(ce, 5);
    double *R, *S, *out;
    HIPC(hipMalloc(&R, sizeof(double) * (size_t)c.M * c.N));
    HIPC(hipMalloc(&S, sizeof(double) * (size_t)c.M * c.N));
    HIPC(hipMalloc(&out, 8));
    HIPC(hipMemset(out, 0, 8));
    if (run(c, variant, A, B, C, bias, add) != 0) {
        fprintf(stderr, "%s: %s\n", c.name, aas_last_error());
        exit(3);
    }
    for (int b = 0; b < c.batch; ++b) {
        dim3 g((c.N + 63) / 64, c.M);
        ref_kernel<<<g, 64>>>(c.mode, c.M, c.N, c.K, A + b * c.sA, c.lda, B + b * c.sB, c.ldb, R, S, c.kdivA, c.kouterA, c.kdivB, c.kouterB);
        cmp_kernel<<<g, 64>>>(c.M, c.N, C + b * c.sC, c.ldc, R, S, c.accumulate ? C0 + b * c.sC : nullptr, c.bias ? bias : nullptr,
                              c.addend ? add + b * c.sC : nullptr, c.ldc, out);
    }
    double err;
    HIPC(hipMemcpy(&err, out, 8, hipMemcpyDeviceToHost));
    for (void* q : {(void*)A, (void*)B, (void*)C, (void*)C0, (void*)bias, (void*)add, (void*)R, (void*)S, (void*)out}) HIPC(hipFree(q));
    return err;
}

static double time_case(Case c, int variant, int iters) {
    const bool akc = c.mode != 2, bkc = c.mode == 0;
    size_t ae = opnd_elems(akc, c.M, c.K, c.lda, c.kdivA, c.kouterA) + (size_t)(c.batch - 1) * c.sA;
    size_t be = opnd_elems(bkc, c.N, c.K, c.ldb, c.kdivB, c.kouterB) + (size_t)(c.batch - 1) * c.sB;
    size_t ce = (size_t)(c.M - 1) * c.ldc + c.N + (size_t)(c.batch - 1) * c.sC;
    float *A = dalloc(ae, 1), *B = dalloc(be, 2), *C = dalloc(ce, 3), *bias = dalloc(c.N, 4), *add = dalloc(ce, 5);
    for (int i = 0; i < 3; ++i) run(c, variant, A, B, C, bias, add);
    HIPC(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0));
    HIPC(hipEventCreate(&e1));
    HIPC(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) run(c, variant, A, B, C, bias, add);
    HIPC(hipEventRecord(e1, nullptr));
    HIPC(hipEventSynchronize(e1));
    float ms;
    HIPC(hipEventElapsedTime(&ms, e0, e1));
    for (void* q : {(void*)A, (void*)B, (void*)C, (void*)bias, (void*)add}) HIPC(hipFree(q));
    return ms / iters;
}

static Case mk(const char* name, int mode, int M, int N, int K, int bias = 0, int addend = 0, int acc = 0) {
    Case c;
    memset(&c, 0, sizeof(c));
    c.name = name; c.mode = mode; c.M = M; c.N = N; c.K = K;
    c.lda = mode == 2 ? M : K;
    c.ldb = mode == 0 ? K : N;
    c.ldc = N;
    c.bias = bias; c.addend = addend; c.accumulate = acc; c.batch = 1;
    return c;
}

// the four weight-gradient products of one layer as one aas_gemm_f32_multi launch vs four aas_gemm_f32 launches
static void time_wgrad(int R, int GH, int I, int iters) {
    float* dg = dalloc((size_t)R * 2 * GH, 1);
    float* x = dalloc((size_t)R * I, 2);
    float* h = dalloc((size_t)2 * R * I, 3);
    float* out[4];
    for (int i = 0; i < 4; ++i) out[i] = dalloc((size_t)GH * I, 10 + i);
    const float* Am[4] = {dg, dg + GH, dg, dg + GH};
    const float* Bm[4] = {x, x, h, h + (size_t)R * I};
    const int Km[4] = {R, R, R - 30, R - 30};
    for (int variant = 1; variant >= 0; --variant) {
        aas_set_gemm_variant(variant);
        for (int multi = 0; multi < 2; ++multi) {
            auto go = [&]() {
                if (multi) {
                    if (aas_gemm_f32_multi(nullptr, 2, 4, GH, I, Km, Am, 2 * GH, Bm, I, out, I, 1, 0, 0, 0, nullptr)) { fprintf(stderr, "%s\n", aas_last_error()); exit(3); }
                } else {
                    for (int i = 0; i < 4; ++i)
                        if (aas_gemm_f32(nullptr, 2, GH, I, R, Am[i], 2 * GH, Bm[i], I, out[i], I, nullptr, nullptr, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0)) {
                            fprintf(stderr, "%s\n", aas_last_error()); exit(3);
                        }
                }
            };
            for (int i = 0; i < 3; ++i) go();
            HIPC(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
            HIPC(hipEventRecord(e0, nullptr));
            for (int i = 0; i < iters; ++i) go();
            HIPC(hipEventRecord(e1, nullptr));
            HIPC(hipEventSynchronize(e1));
            float ms;
            HIPC(hipEventElapsedTime(&ms, e0, e1));
            ms /= iters;
            printf("wgrad layer R=%d GH=%d I=%d  %s %s: %.3f ms  %.1f TFLOP/s\n", R, GH, I, variant ? "legacy" : "lds-dma", multi ? "one multi launch" : "four launches   ",
                   ms, 4 * 2.0 * GH * I * R / ms / 1e9);
        }
    }
    HIPC(hipFree(dg)); HIPC(hipFree(x)); HIPC(hipFree(h));
    for (int i = 0; i < 4; ++i) HIPC(hipFree(out[i]));
}

__global__ void maxdiff_kernel(const float* a, const float* b, size_t n, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float d = fabsf(a[i] - b[i]) / (fabsf(b[i]) + 1.0f);
    atomicMax((unsigned*)out, __float_as_uint(d));
}

// one multi launch (LDS-DMA kernel, whatever split it picks) against four register-staged launches
static int check_multi(int R, int GH, int I) {
    float* dg = dalloc((size_t)R * 2 * GH, 1);
    float* x = dalloc((size_t)R * I, 2);
    float* h = dalloc((size_t)2 * R * I, 3);
    float *o0[4], *o1[4];
    for (int i = 0; i < 4; ++i) { o0[i] = dalloc((size_t)GH * I, 10 + i); o1[i] = dalloc((size_t)GH * I, 10 + i); }
    const float* Am[4] = {dg, dg + GH, dg + 30 * 2 * GH, dg + GH};
    const float* Bm[4] = {x, x, h, h + (size_t)R * I + 30 * I};
    const int Km[4] = {R, R, R - 30, R - 30};
    aas_set_gemm_variant(0);
    if (aas_gemm_f32_multi(nullptr, 2, 4, GH, I, Km, Am, 2 * GH, Bm, I, o0, I, 1, 0, 0, 0, nullptr)) { fprintf(stderr, "%s\n", aas_last_error()); exit(3); }
    aas_set_gemm_variant(1);
    for (int i = 0; i < 4; ++i)
        if (aas_gemm_f32(nullptr, 2, GH, I, Km[i], Am[i], 2 * GH, Bm[i], I, o1[i], I, nullptr, nullptr, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0)) exit(3);
    float* out;
    HIPC(hipMalloc(&out, 4));
    HIPC(hipMemset(out, 0, 4));
    for (int i = 0; i < 4; ++i) maxdiff_kernel<<<(unsigned)(((size_t)GH * I + 255) / 256), 256>>>(o0[i], o1[i], (size_t)GH * I, out);
    float d;
    HIPC(hipMemcpy(&d, out, 4, hipMemcpyDeviceToHost));
    const bool ok = d < 2e-4f;
    printf("check multi wgrad R=%d GH=%d I=%d: max |multi - four launches| / (|ref| + 1) = %.2e %s\n", R, GH, I, d, ok ? "ok" : "FAIL");
    HIPC(hipFree(dg)); HIPC(hipFree(x)); HIPC(hipFree(h)); HIPC(hipFree(out));
    for (int i = 0; i < 4; ++i) { HIPC(hipFree(o0[i])); HIPC(hipFree(o1[i])); }
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "all";
    if (getenv("GEMM32_FLAGS")) aas_set_debug_flags(atoi(getenv("GEMM32_FLAGS")));
    int bad = 0;
    if (!strcmp(what, "check") || !strcmp(what, "all")) {
        std::vector<Case> cs;
        cs.push_back(mk("nt 6000x2000x500 bias", 0, 6000, 2000, 500, 1));
        cs.push_back(mk("nn 6000x500x4000 addend", 1, 6000, 500, 4000, 0, 1));
        cs.push_back(mk("tn 2000x500x6000 acc", 2, 2000, 500, 6000, 0, 0, 1));
        cs.push_back(mk("tn 2000x500x5970", 2, 2000, 500, 5970));
        cs.push_back(mk("nt 130x68x100", 0, 130, 68, 100));
        cs.push_back(mk("nt 257x129x36 (ragged: general kernel)", 0, 257, 129, 36));
        cs.push_back(mk("nt 1000x500x80 bias", 0, 1000, 500, 80, 1));
        cs.push_back(mk("nn 300x132x260 bias addend acc", 1, 300, 132, 260, 1, 1, 1));
        cs.push_back(mk("nn 128x128x64", 1, 128, 128, 64));
        cs.push_back(mk("tn 132x260x1300 (split-K)", 2, 132, 260, 1300));
        cs.push_back(mk("tn 64x64x4096 acc (split-K)", 2, 64, 64, 4096, 0, 0, 1));
        cs.push_back(mk("nt 2550x3000x1000", 0, 2550, 3000, 1000));
        {   // both directions' projections in one batched launch: same A, B / C strided
            Case c = mk("nt batch2 6000x2000x500 (strided B, C halves)", 0, 6000, 2000, 500);
            c.batch = 2; c.sA = 0; c.sB = (long long)2000 * 500 + 2000 * 500; c.sC = 2000; c.ldc = 4000;
            cs.push_back(c);
        }
        {   // dx = [dg_f | dg_r] x [W ; W_r]: B rows two-level
            Case c = mk("nn 6000x500x4000 two-level B rows", 1, 6000, 500, 4000, 0, 1);
            c.kdivB = 2000; c.kouterB = (long long)2000 * 500 + 2000 * 500;
            cs.push_back(c);
        }
        {   // conv weight gradient: B rows two-level (kdivB = T1, kouterB = T*F), overlapping rows
            Case c = mk("tn conv wgrad 128x880x(30*95)", 2, 128, 880, 30 * 95);
            c.ldb = 160; c.kdivB = 95; c.kouterB = 200 * 80;
            cs.push_back(c);
        }
        {   // implicit im2col forward: A rows overlap (lda < K)
            Case c = mk("nt conv fwd batch 30: 95x128x880", 0, 95, 128, 880, 1);
            c.lda = 160; c.batch = 30; c.sA = 200 * 80; c.sB = 0; c.sC = 95 * 128;
            cs.push_back(c);
        }
        for (auto& c : cs)
            for (int v = 1; v >= 0; --v) {
                double e = check_case(c, v);
                const bool ok = e < 4e-6;
                printf("check %-52s %-8s max |err| / sum|ab| = %.2e %s\n", c.name, v ? "legacy" : "lds-dma", e, ok ? "ok" : "FAIL");
                bad += !ok;
            }
        bad += check_multi(6000, 2000, 500);
        bad += check_multi(2550, 3000, 1000);
        bad += check_multi(700, 512, 256);
    }
    if (!strcmp(what, "time") || !strcmp(what, "all")) {
        mfma_peak();
        std::vector<Case> cs;
        cs.push_back(mk("nt 6000x2000x500 (E projection, one direction)", 0, 6000, 2000, 500));
        {
            Case c = mk("nt batch2 6000x2000x500 (E projection)", 0, 6000, 2000, 500);
            c.batch = 2; c.sB = (long long)2 * 2000 * 500; c.sC = 2000; c.ldc = 4000;
            cs.push_back(c);
        }
        {
            Case c = mk("nt batch2 12000x2000x500 (D projection)", 0, 12000, 2000, 500);
            c.batch = 2; c.sB = (long long)2 * 2000 * 500; c.sC = 2000; c.ldc = 4000;
            cs.push_back(c);
        }
        cs.push_back(mk("nn 6000x500x2000", 1, 6000, 500, 2000));
        cs.push_back(mk("nn 6000x500x4000 (E input gradient)", 1, 6000, 500, 4000, 0, 1));
        cs.push_back(mk("nn 12000x500x4000 (D input gradient)", 1, 12000, 500, 4000, 0, 1));
        cs.push_back(mk("tn 2000x500x6000 (E weight gradient)", 2, 2000, 500, 6000, 0, 0, 1));
        cs.push_back(mk("tn 2000x500x12000 (D weight gradient)", 2, 2000, 500, 12000, 0, 0, 1));
        cs.push_back(mk("tn 3000x1000x2550 (A weight gradient)", 2, 3000, 1000, 2550, 0, 0, 1));
        cs.push_back(mk("nt 2550x6000x1000 (A projection both dirs)", 0, 2550, 6000, 1000));
        cs.push_back(mk("nn 2550x1000x6000 (A input gradient)", 1, 2550, 1000, 6000));
        cs.push_back(mk("nt 8192x8192x1024", 0, 8192, 8192, 1024));
        cs.push_back(mk("nt 4096x4096x4096", 0, 4096, 4096, 4096));
        for (auto& c : cs) {
            double t1 = getenv("GEMM32_SKIP_LEGACY") ? 1e9 : time_case(c, 1, 20), t0 = time_case(c, 0, 20);
            double fl = 2.0 * c.M * c.N * c.K * c.batch;
            printf("time  %-48s legacy %.3f ms %6.1f TFLOP/s | lds-dma %.3f ms %6.1f TFLOP/s\n", c.name, t1, fl / t1 / 1e9, t0, fl / t0 / 1e9);
        }
        if (!getenv("GEMM32_SKIP_LEGACY")) {
            time_wgrad(6000, 2000, 500, 20);
            time_wgrad(12000, 2000, 500, 20);
            time_wgrad(2550, 3000, 1000, 20);
        }
    }
    if (bad) printf("%d FAILED\n", bad);
    return bad ? 1 : 0;
}
