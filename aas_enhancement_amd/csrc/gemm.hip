// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32): 128x128x16 block tile, 4 waves (2x2),
// each wave 64x64 = 2x2 MFMA tiles; operands staged k-major in LDS (conflict-free ds_read_b32),
// register-prefetch double buffering.  Exact fp32 (k-ordered fma chain).
// Replaces the cuBLAS/cuDNN GEMMs below nn.LSTM/nn.GRU/nn.Conv1d/nn.Linear
// (reference Speech_enhancement_by_AAS/model.py:73-74,94-95,216-217,289,297,317).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDT = BM + 4;

struct GemmP {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* addend;
    int M, N, K;
    int64_t lda, ldb, ldc, ldd;
    int accumulate;
    int batch;
    int64_t sA, sB, sC;
    int kdivA;
    int64_t kouterA;
    int kdivB;
    int64_t kouterB;
    int splitk;
    int flags;  // ablation bits (aas_set_debug_flags): 16 skip MFMA, 32 skip LDS staging, 64 skip global loads
    const float* kscale;  // TN, fp32 kernel: reduction row r of A is scaled by kscale[r % knb] while it is staged (per-utterance
    int knb;              // weights of a weight-gradient product: no separate scaling pass over d(gates)); null = none
};

__device__ __forceinline__ int64_t krow_addr(int r, int kdiv, int64_t kouter, int64_t ld) {
    return kdiv > 0 ? (int64_t)(r / kdiv) * kouter + (int64_t)(r % kdiv) * ld : (int64_t)r * ld;
}

// k-contiguous operand: tile [128 rows][16 k]; 512 / NTH float4 per thread (NTH = threads of the workgroup).
template <bool VEC, int NTH = 256>
__device__ __forceinline__ void load_kcontig(const float* __restrict__ base, int64_t ld, int row0, int rmax,
                                             int k0, int kend, int tid, f32x4 (&r)[512 / NTH]) {
#pragma unroll
    for (int i = 0; i < 512 / NTH; ++i) {
        int idx = tid + i * NTH;
        int row = idx >> 2, kq = idx & 3;
        int gr = row0 + row, gk = k0 + kq * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gr < rmax) {
            const float* p = base + (int64_t)gr * ld + gk;
            if (VEC) {
                if (gk < kend) v = *reinterpret_cast<const f32x4*>(p);
            } else {
                if (gk + 0 < kend) v.x = p[0];
                if (gk + 1 < kend) v.y = p[1];
                if (gk + 2 < kend) v.z = p[2];
                if (gk + 3 < kend) v.w = p[3];
            }
        }
        r[i] = v;
    }
}
template <int NTH = 256>
__device__ __forceinline__ void store_kcontig(float (*S)[LDT], int tid, const f32x4 (&r)[512 / NTH]) {
#pragma unroll
    for (int i = 0; i < 512 / NTH; ++i) {
        int idx = tid + i * NTH;
        int row = idx >> 2, kq = idx & 3;
        S[kq * 4 + 0][row] = r[i].x;
        S[kq * 4 + 1][row] = r[i].y;
        S[kq * 4 + 2][row] = r[i].z;
        S[kq * 4 + 3][row] = r[i].w;
    }
}
// row-contiguous operand ([K, cols], cols contiguous): tile [16 k][128 cols]
template <bool VEC, int NTH = 256>
__device__ __forceinline__ void load_rcontig(const float* __restrict__ base, int64_t ld, int kdiv, int64_t kouter,
                                             int c0, int cmax, int k0, int kend, int tid, f32x4 (&r)[512 / NTH],
                                             const float* __restrict__ kscale = nullptr, int knb = 1) {
#pragma unroll
    for (int i = 0; i < 512 / NTH; ++i) {
        int idx = tid + i * NTH;
        int kr = idx >> 5, cq = idx & 31;
        int gk = k0 + kr, gc = c0 + cq * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gk < kend) {
            const float* p = base + krow_addr(gk, kdiv, kouter, ld) + gc;
            if (VEC) {
                if (gc < cmax) v = *reinterpret_cast<const f32x4*>(p);
            } else {
                if (gc + 0 < cmax) v.x = p[0];
                if (gc + 1 < cmax) v.y = p[1];
                if (gc + 2 < cmax) v.z = p[2];
                if (gc + 3 < cmax) v.w = p[3];
            }
            if (kscale) {
                const float sc = kscale[gk % knb];
                v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
            }
        }
        r[i] = v;
    }
}
template <int NTH = 256>
__device__ __forceinline__ void store_rcontig(float (*S)[LDT], int tid, const f32x4 (&r)[512 / NTH]) {
#pragma unroll
    for (int i = 0; i < 512 / NTH; ++i) {
        int idx = tid + i * NTH;
        int kr = idx >> 5, cq = idx & 31;
        *reinterpret_cast<f32x4*>(&S[kr][cq * 4]) = r[i];
    }
}

// W8: eight waves per workgroup (4 x 2, wave tile 32 x 64) instead of four (2 x 2, 64 x 64) - two waves per SIMD from ONE
// workgroup and 65 registers per lane instead of 184-204, so up to four workgroups fit a CU: a single wave per SIMD had nothing to
// cover its LDS and barrier waits (the plane GEMMs do the same for sparse grids, gemm_planes.hip).
template <bool A_KC, bool B_KC, bool VECA, bool VECB, bool W8 = false>
__global__ __launch_bounds__(W8 ? 512 : 256) void gemm_f32_kernel(GemmP p) {
    constexpr int NTH = W8 ? 512 : 256, NLD = 512 / NTH, MI = W8 ? 1 : 2;
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int mrow = wm * (W8 ? 32 : 64);       // first tile row of this wave
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    int z = blockIdx.z;
    const float* A = p.A;
    const float* B = p.B;
    float* C = p.C;
    const float* addend = p.addend;
    int kbeg = 0, kend = p.K;
    if (p.splitk > 1) {
        int ktiles = (p.K + BK - 1) / BK;
        int per = (ktiles + p.splitk - 1) / p.splitk;
        kbeg = z * per * BK;
        kend = min(p.K, (z + 1) * per * BK);
        if (kbeg >= kend) return;
    } else if (p.batch > 1) {
        A += (int64_t)z * p.sA;
        B += (int64_t)z * p.sB;
        C += (int64_t)z * p.sC;
        if (addend) addend += (int64_t)z * p.sC;
    }
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[NLD], rb[NLD];
    auto gload = [&](int k0) {
        if (A_KC) load_kcontig<VECA, NTH>(A, p.lda, m0, p.M, k0, kend, tid, ra);
        else load_rcontig<VECA, NTH>(A, p.lda, p.kdivA, p.kouterA, m0, p.M, k0, kend, tid, ra, p.kscale, p.knb);
        if (B_KC) load_kcontig<VECB, NTH>(B, p.ldb, n0, p.N, k0, kend, tid, rb);
        else load_rcontig<VECB, NTH>(B, p.ldb, p.kdivB, p.kouterB, n0, p.N, k0, kend, tid, rb);
    };
    auto sstore = [&](int buf) {
        if (A_KC) store_kcontig<NTH>(As[buf], tid, ra); else store_rcontig<NTH>(As[buf], tid, ra);
        if (B_KC) store_kcontig<NTH>(Bs[buf], tid, rb); else store_rcontig<NTH>(Bs[buf], tid, rb);
    };
    gload(kbeg);
    sstore(0);
    __syncthreads();
    int buf = 0;
    const int l31 = lane & 31, lh = lane >> 5;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = (k0 + BK) < kend;
        if (more) gload(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a0 = As[buf][kk + lh][mrow + l31];
            float b0 = Bs[buf][kk + lh][wn * 64 + l31];
            float b1 = Bs[buf][kk + lh][wn * 64 + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            if constexpr (!W8) {
                float a1 = As[buf][kk + lh][mrow + 32 + l31];
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool first = (p.splitk <= 1) || (z == 0);
#pragma unroll
    for (int mt = 0; mt < MI; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            int n = n0 + wn * 64 + nt * 32 + l31;
            if (n >= p.N) continue;
            float bv = (p.bias && first) ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + mrow + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= p.M) continue;
                float v = acc[mt][nt][r] + bv;
                if (addend && first) v += addend[(int64_t)m * p.ldd + n];
                float* cp = C + (int64_t)m * p.ldc + n;
                if (p.splitk > 1) atomicAdd(cp, v);
                else if (p.accumulate) *cp += v;
                else *cp = v;
            }
        }
}


// =================================================================================================
// Split-bf16 ("bf16x3") GEMM: every fp32 operand x is staged in LDS as hi = bf16(x) (RNE) and
// lo = bf16(x - hi); C += Ah*Bh + Al*Bh + Ah*Bl on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
// Dropped term Al*Bl ~ 2^-16 relative: fp32-class accuracy (~1e-5 rel) at 3/16 of the fp32-MFMA
// issue time.  Tile 128x128x32, 4 waves (2x2) of 64x64, LDS rows of 32 bf16 padded to 80 B
// (conflict-free ds_read_b128), register-prefetched double buffering.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SBK = 32;          // k per stage
constexpr int SROW = 80;         // bytes per LDS row (32 bf16 + 16 B pad)
constexpr int SARR = 128 * SROW; // bytes per (operand, hi|lo) array

// split 4 floats -> packed hi (2 dwords) and lo (2 dwords)
__device__ __forceinline__ void split4(const f32x4& v, u32x2& hi, u32x2& lo) {
    // hi = rne_bf16(x), lo = rne_bf16(x - hi): |x - hi - lo| <= 2^-18 |x| (round-to-nearest on both halves gains
    // two bits over truncation).  NB: copy the lanes to scalars first - __builtin_bit_cast applied directly to an
    // ext_vector element (v.y ...) is miscompiled by ROCm 7.2 clang into a read of element 0.
    const float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
    const bf16x2 h01 = {(__bf16)x0, (__bf16)x1}, h23 = {(__bf16)x2, (__bf16)x3};
    const unsigned w01 = __builtin_bit_cast(unsigned, h01), w23 = __builtin_bit_cast(unsigned, h23);
    hi.x = w01;
    hi.y = w23;
    const bf16x2 l01 = {(__bf16)(x0 - __uint_as_float(w01 << 16)), (__bf16)(x1 - __uint_as_float(w01 & 0xFFFF0000u))};
    const bf16x2 l23 = {(__bf16)(x2 - __uint_as_float(w23 << 16)), (__bf16)(x3 - __uint_as_float(w23 & 0xFFFF0000u))};
    lo.x = __builtin_bit_cast(unsigned, l01);
    lo.y = __builtin_bit_cast(unsigned, l23);
}

// k-contiguous operand: tile [128 rows][32 k]; 4 float4 per thread (row = idx>>3, kq = idx&7)
template <bool VEC>
__device__ __forceinline__ void sload_kc(const float* __restrict__ base, int64_t ld, int row0, int rmax, int k0, int kend,
                                         int tid, f32x4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256;
        const int row = idx >> 3, kq = idx & 7;
        const int gr = row0 + row, gk = k0 + kq * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gr < rmax) {
            const float* p = base + (int64_t)gr * ld + gk;
            if (VEC) {
                if (gk < kend) v = *reinterpret_cast<const f32x4*>(p);
            } else {
                if (gk + 0 < kend) v.x = p[0];
                if (gk + 1 < kend) v.y = p[1];
                if (gk + 2 < kend) v.z = p[2];
                if (gk + 3 < kend) v.w = p[3];
            }
        }
        r[i] = v;
    }
}
__device__ __forceinline__ void sstore_kc(char* hi, char* lo, int tid, const f32x4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256;
        const int row = idx >> 3, kq = idx & 7;
        u32x2 h, l;
        split4(r[i], h, l);
        *reinterpret_cast<u32x2*>(hi + row * SROW + kq * 8) = h;
        *reinterpret_cast<u32x2*>(lo + row * SROW + kq * 8) = l;
    }
}
// row-contiguous operand ([K, cols]): each thread loads a 4(k) x 4(col) micro-tile (4 float4 from 4
// consecutive k rows), transposes it in registers and stores, per column, 4 consecutive k as 8 B.
template <bool VEC>
__device__ __forceinline__ void sload_rc(const float* __restrict__ base, int64_t ld, int kdiv, int64_t kouter, int c0,
                                         int cmax, int k0, int kend, int tid, f32x4 (&r)[4]) {
    const int kg = tid >> 5, cq = tid & 31;  // 8 k-groups x 32 column quads
    const int gc = c0 + cq * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gk = k0 + kg * 4 + i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gk < kend) {
            const float* p = base + krow_addr(gk, kdiv, kouter, ld) + gc;
            if (VEC) {
                if (gc < cmax) v = *reinterpret_cast<const f32x4*>(p);
            } else {
                if (gc + 0 < cmax) v.x = p[0];
                if (gc + 1 < cmax) v.y = p[1];
                if (gc + 2 < cmax) v.z = p[2];
                if (gc + 3 < cmax) v.w = p[3];
            }
        }
        r[i] = v;
    }
}
__device__ __forceinline__ void sstore_rc(char* hi, char* lo, int tid, const f32x4 (&r)[4]) {
    const int kg = tid >> 5, cq = tid & 31;
    // lanes of a wave write rows 4 apart (80-B rows): un-swizzled that is a 16-way LDS bank conflict.  XOR the
    // 16-byte pair index with ((row >> 3) & 3): a wave64 8-byte store then takes its 4-cycle floor; readers of a
    // row-contiguous operand apply the same XOR (rc_swz below).
    const int kgs = kg ^ (((cq >> 1) & 3) << 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // column cq*4 + j gets k = kg*4 .. kg*4+3
        const f32x4 col = {r[0][j], r[1][j], r[2][j], r[3][j]};
        u32x2 h, l;
        split4(col, h, l);
        const int row = cq * 4 + j;
        *reinterpret_cast<u32x2*>(hi + row * SROW + kgs * 8) = h;
        *reinterpret_cast<u32x2*>(lo + row * SROW + kgs * 8) = l;
    }
}

template <bool A_KC, bool B_KC, bool VECA, bool VECB>
__global__ __launch_bounds__(256) void gemm_split_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * SARR];  // [A_hi, A_lo, B_hi, B_lo][128][80 B] = 40 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    int z = blockIdx.z;
    const float* A = p.A;
    const float* B = p.B;
    float* C = p.C;
    const float* addend = p.addend;
    int kbeg = 0, kend = p.K;
    if (p.splitk > 1) {
        int ktiles = (p.K + SBK - 1) / SBK;
        int per = (ktiles + p.splitk - 1) / p.splitk;
        kbeg = z * per * SBK;
        kend = min(p.K, (z + 1) * per * SBK);
        if (kbeg >= kend) return;
    } else if (p.batch > 1) {
        A += (int64_t)z * p.sA;
        B += (int64_t)z * p.sB;
        C += (int64_t)z * p.sC;
        if (addend) addend += (int64_t)z * p.sC;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // two register sets: the global loads of tile i+2 are issued right after tile i has been written to LDS,
    // so every load has two full k-steps (MFMA phases) to land - the small-grid shapes (wgrad, dgrad with
    // ~1 block per CU) are latency- not bandwidth-bound
    f32x4 ra0[4], rb0[4], ra1[4], rb1[4];
    auto gload = [&](int k0, f32x4 (&ra)[4], f32x4 (&rb)[4]) {
        if (A_KC) sload_kc<VECA>(A, p.lda, m0, p.M, k0, kend, tid, ra);
        else sload_rc<VECA>(A, p.lda, p.kdivA, p.kouterA, m0, p.M, k0, kend, tid, ra);
        if (B_KC) sload_kc<VECB>(B, p.ldb, n0, p.N, k0, kend, tid, rb);
        else sload_rc<VECB>(B, p.ldb, p.kdivB, p.kouterB, n0, p.N, k0, kend, tid, rb);
    };
    auto sstore = [&](const f32x4 (&ra)[4], const f32x4 (&rb)[4]) {
        char* st = smem;
        if (A_KC) sstore_kc(st, st + SARR, tid, ra); else sstore_rc(st, st + SARR, tid, ra);
        if (B_KC) sstore_kc(st + 2 * SARR, st + 3 * SARR, tid, rb); else sstore_rc(st + 2 * SARR, st + 3 * SARR, tid, rb);
    };
    const int l31 = lane & 31, lh = lane >> 5;
    auto compute = [&]() {
        const char* st = smem;
#pragma unroll
        for (int kk = 0; kk < SBK; kk += 16) {
            // A operand of 32x32x16: lane (row l&31, half l>>5) holds k = 8*half + j; same map for B columns
            const int px = kk / 8 + lh;  // 16-byte pair index within the 64-byte row of 32 k
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int rowa = wm * 64 + t * 32 + l31, rowb = wn * 64 + t * 32 + l31;
                const int ra_ = rowa * SROW + (A_KC ? px : (px ^ ((rowa >> 3) & 3))) * 16;
                const int rb_ = rowb * SROW + (B_KC ? px : (px ^ ((rowb >> 3) & 3))) * 16;
                ah[t] = *reinterpret_cast<const bf16x8*>(st + ra_);
                al[t] = *reinterpret_cast<const bf16x8*>(st + SARR + ra_);
                bh[t] = *reinterpret_cast<const bf16x8*>(st + 2 * SARR + rb_);
                bl[t] = *reinterpret_cast<const bf16x8*>(st + 3 * SARR + rb_);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                }
        }
    };
    const bool do_mma = !(p.flags & 16), do_st = !(p.flags & 32), do_ld = !(p.flags & 64);
    if (p.flags & 128) kend = kbeg;  // ablation: epilogue only
#pragma unroll
    for (int i = 0; i < 4; ++i) ra0[i] = rb0[i] = ra1[i] = rb1[i] = (f32x4){1.f, 2.f, 3.f, 4.f};
    if (do_ld) gload(kbeg, ra0, rb0);
    if (do_ld && kbeg + SBK < kend) gload(kbeg + SBK, ra1, rb1);
    for (int k0 = kbeg; k0 < kend; k0 += 2 * SBK) {
        if (do_st) sstore(ra0, rb0);
        __syncthreads();
        if (do_ld && k0 + 2 * SBK < kend) gload(k0 + 2 * SBK, ra0, rb0);
        if (do_mma) compute();
        __syncthreads();
        if (k0 + SBK < kend) {
            if (do_st) sstore(ra1, rb1);
            __syncthreads();
            if (do_ld && k0 + 3 * SBK < kend) gload(k0 + 3 * SBK, ra1, rb1);
            if (do_mma) compute();
            __syncthreads();
        }
    }
    const bool first = (p.splitk <= 1) || (z == 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            int n = n0 + wn * 64 + nt * 32 + l31;
            if (n >= p.N) continue;
            float bv = (p.bias && first) ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= p.M) continue;
                float v = acc[mt][nt][r] + bv;
                if (addend && first) v += addend[(int64_t)m * p.ldd + n];
                float* cp = C + (int64_t)m * p.ldc + n;
                if (p.splitk > 1) atomicAdd(cp, v);
                else if (p.accumulate) *cp += v;
                else *cp = v;
            }
        }
}

template <bool A_KC, bool B_KC>
int launch_split(const GemmP& p, bool va, bool vb, dim3 grid, hipStream_t s) {
    if (va && vb) hipLaunchKernelGGL((gemm_split_kernel<A_KC, B_KC, true, true>), grid, dim3(256), 0, s, p);
    else if (va) hipLaunchKernelGGL((gemm_split_kernel<A_KC, B_KC, true, false>), grid, dim3(256), 0, s, p);
    else if (vb) hipLaunchKernelGGL((gemm_split_kernel<A_KC, B_KC, false, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_split_kernel<A_KC, B_KC, false, false>), grid, dim3(256), 0, s, p);
    return 0;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <bool A_KC, bool B_KC>
void launch(const GemmP& p, bool va, bool vb, dim3 grid, hipStream_t s) {
    // eight waves per workgroup whenever both operands take 16-byte loads (debug bit 1073741824: always four; AAS_GEMM_W8_MAX=n: only
    // for grids below n workgroups).  Same box, config-2 step: four waves 31.7-31.8 ms, eight waves below 512 workgroups 30.9-31.1,
    // always 30.8; alone on the chip the 2000 x 500 x 6000 weight-gradient product 0.176 -> 0.149 ms, the GRU's 0.255 -> 0.204.
    static const int64_t w8_max = aas_ablation_env("AAS_GEMM_W8_MAX") ? atoll(aas_ablation_env("AAS_GEMM_W8_MAX")) : (int64_t)1 << 40;
    const bool w8 = va && vb && (int64_t)grid.x * grid.y * grid.z < w8_max && !(p.flags & 1073741824);
    if (w8) hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true, true, true>), grid, dim3(512), 0, s, p);
    else if (va && vb) hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true, true>), grid, dim3(256), 0, s, p);
    else if (va) hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true, false>), grid, dim3(256), 0, s, p);
    else if (vb) hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, false, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, false, false>), grid, dim3(256), 0, s, p);
}

}  // namespace

static int gemm_f32_impl(aasStream_t stream, int mode, int M, int N, int K, const float* A, int64_t lda,
                         const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                         const float* addend, int64_t ldd, int accumulate, int batch, int64_t strideA,
                         int64_t strideB, int64_t strideC, int kdivA, int64_t kouterA, int kdivB,
                         int64_t kouterB, const float* kscale, int knb) {
    AAS_CHECK(mode >= 0 && mode <= 2, "aas_gemm_f32: bad mode %d", mode);
    AAS_CHECK(M >= 0 && N >= 0 && K >= 0 && batch >= 1, "aas_gemm_f32: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, batch);
    AAS_CHECK(A && B && C, "aas_gemm_f32: null operand");
    AAS_CHECK(!(batch > 1 && mode == AAS_GEMM_TN), "aas_gemm_f32: batch>1 unsupported for TN");
    if (M == 0 || N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (aas_precision_value() != 1 && kscale == nullptr) {
        // fp32 arithmetic: the LDS-DMA kernel (gemm32.hip) wherever both operands take 16-byte chunks
        const int rc32 = aas_gemm32_try(s, mode, M, N, K, A, lda, B, ldb, C, ldc, bias, addend, ldd, accumulate, batch, strideA, strideB,
                                        strideC, kdivA, kouterA, kdivB, kouterB, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (rc32 >= 0) return rc32;
    }
    GemmP p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.addend = addend;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldd = ldd;
    p.accumulate = accumulate; p.batch = batch; p.sA = strideA; p.sB = strideB; p.sC = strideC;
    p.kdivA = kdivA; p.kouterA = kouterA; p.kdivB = kdivB; p.kouterB = kouterB;
    p.splitk = 1;
    p.flags = aas_debug_flags_value();
    p.kscale = kscale; p.knb = knb > 0 ? knb : 1;
    dim3 grid(cdiv(N, BN), cdiv(M, BM), batch);
    // split-K (atomic epilogue) when the MxN grid cannot fill the 256 CUs and K is deep
    int blocks = grid.x * grid.y;
    const bool split_prec = aas_precision_value() == 1;
    // TN (weight gradients: tiny MxN, deep K) wants >= 2 blocks per CU to hide the k-step latency; its atomic
    // epilogue traffic (splitk x MxN x 4 B at ~1.3 TB/s) stays far below the time saved
    // fp32 kernel (eight light waves per workgroup, up to four workgroups per CU): two workgroups per CU for every mode
    // (config-2 step, same box: target 256: 30.9 ms, 384: 30.7-30.9, 512: 30.55-30.6, 640: 30.65-30.7, 1024: 31.2, 128: 32.6-33.1)
    static const int sk_target = aas_ablation_env("AAS_GEMM_SK_TARGET") ? atoi(aas_ablation_env("AAS_GEMM_SK_TARGET")) : 0;   // experiment switch
    const int target = sk_target > 0 ? sk_target : (!split_prec || mode == AAS_GEMM_TN) ? 512 : 256;
    if (batch == 1 && blocks < (target * 3) / 4 && K >= 1024) {
        int want = (target + blocks - 1) / blocks;
        int maxs = K / 256;
        int sk = want < maxs ? want : maxs;
        if (sk > 16) sk = 16;
        if (sk > 1) {
            p.splitk = sk;
            grid.z = sk;
            if (!accumulate) {
                if (ldc == N) {
                    AAS_HIP(hipMemsetAsync(C, 0, sizeof(float) * (size_t)M * N, s));
                } else {
                    AAS_HIP(hipMemset2DAsync(C, sizeof(float) * ldc, 0, sizeof(float) * N, M, s));
                }
            }
        }
    }
    bool va, vb;
    const bool split = aas_precision_value() == 1;
    int rc = 0;
    if (mode == AAS_GEMM_TN) {
        va = al16(A) && lda % 4 == 0 && M % 4 == 0 && (kdivA == 0 || kouterA % 4 == 0);
        vb = al16(B) && ldb % 4 == 0 && N % 4 == 0 && (kdivB == 0 || kouterB % 4 == 0);
        if (split) rc = launch_split<false, false>(p, va, vb, grid, s);
        else launch<false, false>(p, va, vb, grid, s);
    } else {
        va = al16(A) && lda % 4 == 0 && K % 4 == 0 && strideA % 4 == 0;
        if (mode == AAS_GEMM_NT) {
            vb = al16(B) && ldb % 4 == 0 && K % 4 == 0 && strideB % 4 == 0;
            if (split) rc = launch_split<true, true>(p, va, vb, grid, s);
            else launch<true, true>(p, va, vb, grid, s);
        } else {
            vb = al16(B) && ldb % 4 == 0 && N % 4 == 0 && strideB % 4 == 0 && (kdivB == 0 || kouterB % 4 == 0);
            if (split) rc = launch_split<true, false>(p, va, vb, grid, s);
            else launch<true, false>(p, va, vb, grid, s);
        }
    }
    AAS_CHECK(rc == 0, "aas_gemm_f32: could not raise the dynamic LDS limit for the split-bf16 kernel");
    AAS_LAUNCH_CHECK("aas_gemm_f32");
    return 0;
}

extern "C" int aas_gemm_f32(aasStream_t stream, int mode, int M, int N, int K, const float* A, int64_t lda,
                            const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                            const float* addend, int64_t ldd, int accumulate, int batch, int64_t strideA,
                            int64_t strideB, int64_t strideC, int kdivA, int64_t kouterA, int kdivB,
                            int64_t kouterB) {
    return gemm_f32_impl(stream, mode, M, N, K, A, lda, B, ldb, C, ldc, bias, addend, ldd, accumulate, batch, strideA, strideB, strideC,
                         kdivA, kouterA, kdivB, kouterB, nullptr, 0);
}

extern "C" int aas_gemm_f32_multi(aasStream_t stream, int mode, int n, int M, int N, const int* K, const float* const* A, int64_t lda,
                                  const float* const* B, int64_t ldb, float* const* C, int64_t ldc, int accumulate, int kdiv, int64_t kouterA,
                                  int64_t kouterB, const float* d_alpha) {
    AAS_CHECK(mode >= 0 && mode <= 2 && n >= 1 && n <= 4, "aas_gemm_f32_multi: bad mode %d / problem count %d", mode, n);
    AAS_CHECK(M >= 0 && N >= 0 && K && A && B && C, "aas_gemm_f32_multi: bad sizes M=%d N=%d or null arrays", M, N);
    AAS_CHECK(kdiv >= 0 && (kdiv == 0 || mode == AAS_GEMM_TN), "aas_gemm_f32_multi: two-level reduction rows (kdiv=%d) are a TN feature", kdiv);
    AAS_CHECK(!(d_alpha && aas_precision_value() == 1), "aas_gemm_f32_multi: alpha is not supported in the split-bf16 mode (the plane GEMMs carry it)");
    int kmax = 0;
    for (int i = 0; i < n; ++i) {
        AAS_CHECK(K[i] >= 0 && A[i] && B[i] && C[i], "aas_gemm_f32_multi: problem %d: K=%d or a null operand", i, K[i]);
        kmax = K[i] > kmax ? K[i] : kmax;
    }
    if (M == 0 || N == 0) return 0;
    if (aas_precision_value() != 1) {
        const int rc32 = aas_gemm32_try((hipStream_t)stream, mode, M, N, kmax, nullptr, lda, nullptr, ldb, nullptr, ldc, nullptr, nullptr, 0,
                                        accumulate, 1, 0, 0, 0, kdiv, kouterA, kdiv, kouterB, n, A, B, C, K, d_alpha);
        if (rc32 >= 0) return rc32;
    }
    for (int i = 0; i < n; ++i) {   // operands that do not take 16-byte chunks (or the fast mode): one general launch per problem
        // (alpha rides on the reduction rows of A there: kscale with one entry)
        const int rc = gemm_f32_impl(stream, mode, M, N, K[i], A[i], lda, B[i], ldb, C[i], ldc, nullptr, nullptr, 0, accumulate, 1, 0, 0, 0, kdiv,
                                     kouterA, kdiv, kouterB, d_alpha, d_alpha ? 1 : 0);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int aas_gemm_tn_rowscaled_f32(aasStream_t stream, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                                         float* C, int64_t ldc, int accumulate, const float* d_kscale, int knb) {
    AAS_CHECK(d_kscale && knb > 0, "aas_gemm_tn_rowscaled_f32: scale vector missing");
    AAS_CHECK(aas_precision_value() != 1, "aas_gemm_tn_rowscaled_f32: fp32 mode only (the split-bf16 path scales through its own operands)");
    return gemm_f32_impl(stream, AAS_GEMM_TN, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, accumulate, 1, 0, 0, 0, 0, 0, 0, 0,
                         d_kscale, knb);
}
