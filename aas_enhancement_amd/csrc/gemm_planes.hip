// Split-bf16 GEMM on PRE-SPLIT operand planes (gfx950).
//
// gemm.hip's split kernel converts fp32 tiles to bf16 hi/lo while staging them (VALU work and VGPR staging inside
// the k-loop, fp32 bytes through the L1/L2 path).  Here every operand already lives in HBM as two bf16 planes,
// hi = rne(x) and lo = rne(x - hi), laid out k-contiguous [rows][Kp] with Kp a multiple of 32 and the pad zero-filled
// (aas_split_planes / aas_split_planes_t write them, with the transpose and the per-utterance row weights folded
// into that HBM-bound pass).  The product is then a plain NT bf16 GEMM with three MFMAs per fragment pair,
//     C[m][n] (+)= sum_k  A_hi*B_hi + A_lo*B_hi + A_hi*B_lo        (fp32 accumulate, dropped term lo*lo ~ 2^-18),
// whose k-loop contains nothing but LDS-DMA loads (global_load_lds_dwordx4), ds_read_b128 and MFMA.
//
// Tile 128 x 128 x 32, 256 threads (2 x 2 waves, 64 x 64 per wave as 4 x 4 v_mfma_f32_16x16x32_bf16 tiles), two LDS
// stages of 4 planes x 128 rows x 64 B (64 KB), one workgroup barrier per k-step.  The LDS image is lane-linear
// (LDS-DMA: wave base + 16*lane), so the bank swizzle is applied to the SOURCE chunk a lane fetches and undone by
// the fragment reads: slot = chunk ^ g[(row >> 2) & 3], g = {0,2,3,1}, which makes every ds_read_b128 lane group of
// the 16x16x32 operand (MI355X_MICROARCH.md LDS table) hit 16 distinct 16-byte slots.  The MFMA operands are
// swapped (B fragment first), so a lane ends up with 4 consecutive columns of one C row: 16-byte epilogue stores.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PG {
    const char *A, *B;      // interleaved planes: row r, k-block b (32 k): 64 B of bf16 hi then 64 B of bf16 lo at r*ld*4 + b*128
    float* C;
    const float* bias;      // [N] or null
    const float* addend;    // [M][ldd] or null
    int M, N, K;            // K: multiple of 32 (planes are zero-padded)
    int64_t lda, ldb, ldc, ldd;
    int accumulate, splitk, batch;
    int64_t sA, sB, sC;     // batch strides (elements)
    int flags;
    int gx, gy, tiles, per; // XCD-aware launch (per > 0): 1-D grid of 8*per workgroups, workgroup L -> tile (L%8)*per + L/8
    int multi;              // > 0: `multi` independent problems with their own operand / result pointers (Am/Bm/Cm)
    const char *Am[4], *Bm[4];
    float* Cm[4];
};

constexpr int TK = 32;

__device__ __forceinline__ int swz(int r4) { return (0x78 >> (2 * r4)) & 3; }  // g = {0, 2, 3, 1}

__device__ __forceinline__ void glds16(const void* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// BM x BN tile, WM x WN waves (wave tile (BM/WM) x (BN/WN) as MI x NJ 16x16 MFMA tiles).  SWAP: MFMA operands swapped
// so that a lane owns 4 consecutive columns of a C row (16-byte stores); !SWAP: a lane owns one column of 4 rows,
// the form whose 4-byte atomics coalesce (split-K epilogue).
template <int BM, int BN, int WM, int WN, bool SWAP>
__global__ __launch_bounds__(64 * WM * WN) void gemm_planes_kernel(PG p) {
    constexpr int NW = WM * WN, THREADS = 64 * NW;
    constexpr int MI = BM / WM / 16, NJ = BN / WN / 16;
    constexpr int A_B = BM * 128, B_B = BN * 128;             // bytes of one A / B tile (rows x 32 k x (hi + lo))
    constexpr int STAGE_B = A_B + B_B;
    static_assert(BM % (16 * NW) == 0 && BN % (16 * NW) == 0, "tile rows must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (A_hi, A_lo, B_hi, B_lo)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): with the XCD-aware launch, XCD x takes a
    // CONTIGUOUS run of tiles (n fastest, then m, then problem), so an A row block is fetched by one XCD only (its n-tiles
    // are neighbours in the run) and a B tile once per XCD instead of once per m-tile.
    int bx = blockIdx.x, by = blockIdx.y, z = blockIdx.z;
    if (p.per > 0) {
        const int L = blockIdx.x, q = (L & 7) * p.per + (L >> 3);
        if (q >= p.tiles) return;
        bx = q % p.gx;
        const int t = q / p.gx;
        by = t % p.gy;
        z = t / p.gy;
    }
    const int m0 = by * BM, n0 = bx * BN;
    int kbeg = 0, kend = p.K;
    const int bz = z / p.splitk, kz = z - bz * p.splitk;
    if (p.splitk > 1) {
        const int ktiles = p.K / TK;
        const int per = (ktiles + p.splitk - 1) / p.splitk;
        kbeg = kz * per * TK;
        kend = min(p.K, (kz + 1) * per * TK);
        if (kbeg >= kend) return;
    }
    const char* Ap = p.multi ? p.Am[bz] : p.A + bz * p.sA * 4;
    const char* Bp = p.multi ? p.Bm[bz] : p.B + bz * p.sB * 4;
    float* C = p.multi ? p.Cm[bz] : p.C + bz * p.sC;
    const float* addend = p.addend ? p.addend + bz * p.sC : nullptr;
    if (p.flags & 128) kend = kbeg;  // ablation: epilogue only

    // ---- LDS-DMA: one instruction moves 8 rows x 128 B (hi | lo of one 32-k block: full lines); a 16-row piece of a
    // tile is two instructions, piece pc = j*NW + wave.  lane -> (row = lane>>3, LDS slot = lane&7) and the lane fetches
    // source slot (slot ^ f(row)), f(row) = (row>>1)&7 over the 16 rows of the piece: the LDS image is lane-linear, the
    // XOR makes every ds_read_b128 lane group of the 16x16x32 operand hit 16 distinct 16-byte slots.
    constexpr int PA = BM / (16 * NW), PB = BN / (16 * NW);
    const int r8 = lane >> 3, sl = lane & 7;
    int64_t aoff[PA][2], boff[PB][2];
#pragma unroll
    for (int j = 0; j < PA; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = h * 8 + r8;
            aoff[j][h] = (int64_t)min(m0 + (j * NW + wave) * 16 + r, p.M - 1) * p.lda * 4 + ((sl ^ ((r >> 1) & 7)) << 4);
        }
#pragma unroll
    for (int j = 0; j < PB; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = h * 8 + r8;
            boff[j][h] = (int64_t)min(n0 + (j * NW + wave) * 16 + r, p.N - 1) * p.ldb * 4 + ((sl ^ ((r >> 1) & 7)) << 4);
        }
    auto stage = [&](int st, int k0) {
        char* base = smem + st * STAGE_B + wave * 2048;
        const int64_t kb = (int64_t)k0 * 4;   // byte offset of the k-block inside a row
#pragma unroll
        for (int j = 0; j < PA; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) glds16(Ap + aoff[j][h] + kb, base + j * NW * 2048 + h * 1024);
#pragma unroll
        for (int j = 0; j < PB; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) glds16(Bp + boff[j][h] + kb, base + A_B + j * NW * 2048 + h * 1024);
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets inside a tile: row (lane&15) of a 16-row piece, hi chunk q = lane>>4, lo chunk 4+q
    const int fr = lane & 15, q = lane >> 4;
    const int fx = (fr >> 1) & 7;
    const int fhi = fr * 128 + ((q ^ fx) << 4), flo = fr * 128 + (((4 + q) ^ fx) << 4);
    const int ab = (wm * MI) * 2048, bb = A_B + (wn * NJ) * 2048;

    if (kbeg < kend) stage(0, kbeg);
    int st = 0;
    for (int k0 = kbeg; k0 < kend; k0 += TK, st ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // stage st has landed for every wave; everyone is done reading stage st^1
        if (k0 + TK < kend && !(p.flags & 64)) stage(st ^ 1, k0 + TK);
        if (!(p.flags & 16)) {
            const char* sb = smem + st * STAGE_B;
            bf16x8 ah[MI], al[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(sb + ab + i * 2048 + fhi);
                al[i] = *reinterpret_cast<const bf16x8*>(sb + ab + i * 2048 + flo);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(sb + bb + j * 2048 + fhi);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(sb + bb + j * 2048 + flo);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if (SWAP) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[i], acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    }

    const bool first = (p.splitk <= 1) || (kz == 0);
    if (SWAP) {
        // lane holds C[m][n .. n+3], m = tile row (lane&15), n = 4*(lane>>4)
        const bool vec = ((p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                         (!p.bias || ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0)) &&
                         (!addend || (((p.ldd & 3) == 0) && ((reinterpret_cast<uintptr_t>(addend) & 15) == 0)));
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + (wm * MI + i) * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = n0 + (wn * NJ + j) * 16 + q * 4;
                if (n >= p.N) continue;
                f32x4 v = acc[i][j];
                float* cp = C + (int64_t)m * p.ldc + n;
                if (n + 3 < p.N && vec) {
                    if (p.bias) { const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n); v += b; }
                    if (addend) { const f32x4 a = *reinterpret_cast<const f32x4*>(addend + (int64_t)m * p.ldd + n); v += a; }
                    if (p.accumulate) {
                        const f32x4 o = *reinterpret_cast<const f32x4*>(cp);
                        *reinterpret_cast<f32x4*>(cp) = o + v;
                    } else {
                        *reinterpret_cast<f32x4*>(cp) = v;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (n + r >= p.N) break;
                        float x = v[r];
                        if (p.bias) x += p.bias[n + r];
                        if (addend) x += addend[(int64_t)m * p.ldd + n + r];
                        if (p.accumulate) cp[r] += x;
                        else cp[r] = x;
                    }
                }
            }
        }
    } else {
        // lane holds C[m .. m+3][n], n = tile column (lane&15), m = 4*(lane>>4): 64-byte runs per atomic instruction
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + (wn * NJ + j) * 16 + fr;
            if (n >= p.N) continue;
            const float bv = (p.bias && first) ? p.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + (wm * MI + i) * 16 + q * 4 + r;
                    if (m >= p.M) continue;
                    float x = acc[i][j][r] + bv;
                    if (addend && first) x += addend[(int64_t)m * p.ldd + n];
                    float* cp = C + (int64_t)m * p.ldc + n;
                    if (p.splitk > 1) atomicAdd(cp, x);
                    else if (p.accumulate) *cp += x;
                    else *cp = x;
                }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool SWAP>
int launch_planes(PG p, dim3 grid, hipStream_t s) {
    constexpr int LDS = 2 * (BM * 128 + BN * 128);
    p.per = 0;
    if (p.splitk == 1 && !(p.flags & 4096)) {   // XCD-aware tile order (debug bit 4096: plain 3-D grid)
        p.gx = grid.x; p.gy = grid.y; p.tiles = grid.x * grid.y * grid.z;
        p.per = (p.tiles + 7) / 8;
        grid = dim3(8 * p.per);
    }
    static unsigned char attr_done[AAS_MAX_DEV];
    if (aas_raise_dynamic_lds_once(attr_done, reinterpret_cast<const void*>(&gemm_planes_kernel<BM, BN, WM, WN, SWAP>), LDS)) return -1;
    hipLaunchKernelGGL((gemm_planes_kernel<BM, BN, WM, WN, SWAP>), grid, dim3(64 * WM * WN), LDS, s, p);
    return 0;
}

__device__ __forceinline__ void split2(float x, unsigned short& h, unsigned short& l) {
    const __bf16 hb = (__bf16)x;
    h = __builtin_bit_cast(unsigned short, hb);
    const __bf16 lb = (__bf16)(x - __uint_as_float((unsigned)h << 16));
    l = __builtin_bit_cast(unsigned short, lb);
}

// ---- three-term operands for the fp32-EQUIVALENT products (aas_set_precision(2)) ----------------------------------------------
// x = h + m + l EXACTLY, h = rne_bf16(x), m = rne_bf16(x - h), l = rne_bf16(x - h - m) (24 significant bits in three 8-bit terms).
// The six products hh' + hm' + mh' + mm' + hl' + lh' (dropped: ml' + lm' + ll' <= 2^-25 |x y|, below fp32's rounding unit) are two
// passes of the UNCHANGED three-product kernels (X_hi Y_hi + X_lo Y_hi + X_hi Y_lo) over two plane sets of every operand:
//     set Q1 = (m | h):  m m' + h m' + m h'          set Q2 = (h | l):  h h' + l h' + h l'
// the second pass accumulating into the first one's result (the small terms are summed first).
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    const __bf16 hb = (__bf16)x;
    h = __builtin_bit_cast(unsigned short, hb);
    const float r1 = x - __uint_as_float((unsigned)h << 16);
    const __bf16 mb = (__bf16)r1;
    m = __builtin_bit_cast(unsigned short, mb);
    const float r2 = r1 - __uint_as_float((unsigned)m << 16);
    const __bf16 lb = (__bf16)r2;
    l = __builtin_bit_cast(unsigned short, lb);
}

typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pu32x4 pack8(const unsigned short (&v)[8]) {
    return (pu32x4){v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16), v[6] | ((unsigned)v[7] << 16)};
}
// 8 consecutive k of one row -> the two plane sets (16-byte stores into the hi / lo slots of the k-block's 128-byte line)
__device__ __forceinline__ void store_sets(const float (&v)[8], char* q1, char* q2, int64_t row_bytes_off, int k) {
    unsigned short h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split3(v[e], h[e], m[e], l[e]);
    const int64_t o = row_bytes_off + (k >> 5) * 128 + (k & 31) * 2;
    const pu32x4 hv = pack8(h), mv = pack8(m), lv = pack8(l);
    *reinterpret_cast<pu32x4*>(q1 + o) = mv;
    *reinterpret_cast<pu32x4*>(q1 + o + 64) = hv;
    *reinterpret_cast<pu32x4*>(q2 + o) = hv;
    *reinterpret_cast<pu32x4*>(q2 + o + 64) = lv;
}

__global__ __launch_bounds__(256) void split_rows3_kernel(const float* __restrict__ src, int64_t ld, int64_t R, int K, int Kp,
                                                          char* __restrict__ q1, char* __restrict__ q2, int64_t pitch) {
    const int cpr = Kp / 8;
    const int64_t total = R * cpr;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int k = (int)(i - r * cpr) * 8;
        float v[8];
        const float* s = src + r * ld + k;
        if (vec && k + 8 <= K) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (k + e < K) ? s[e] : 0.f;
        }
        store_sets(v, q1, q2, r * pitch, k);
    }
}

__global__ __launch_bounds__(256) void add3_planes3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                           float* __restrict__ out, int64_t R, int K, int Kp, char* __restrict__ q1,
                                                           char* __restrict__ q2, int64_t pitch) {
    const int cpr = Kp / 8;
    const int64_t total = R * cpr;
    const bool vec = (K & 3) == 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int k = (int)(i - r * cpr) * 8;
        float v[8];
        const int64_t o = r * K + k;
        if (vec && k + 8 <= K) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 x = *reinterpret_cast<const f32x4*>(a + o + 4 * h) + *reinterpret_cast<const f32x4*>(b + o + 4 * h);
                if (c) x += *reinterpret_cast<const f32x4*>(c + o + 4 * h);
                *reinterpret_cast<f32x4*>(out + o + 4 * h) = x;
                v[4 * h] = x[0]; v[4 * h + 1] = x[1]; v[4 * h + 2] = x[2]; v[4 * h + 3] = x[3];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = 0.f;
                if (k + e < K) {
                    v[e] = a[o + e] + b[o + e] + (c ? c[o + e] : 0.f);
                    out[o + e] = v[e];
                }
            }
        }
        store_sets(v, q1, q2, r * pitch, k);
    }
}

// transposing form (the weights' transposed operand of the input-gradient product): see split_rows_t_kernel
__global__ __launch_bounds__(256) void split_rows_t3_kernel(const float* __restrict__ src, int64_t ld, int T, int nb, int nbp, int Cc,
                                                            int64_t Kp, char* __restrict__ q1, char* __restrict__ q2, int64_t tstride, int64_t pitch) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64;
    const int64_t k0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int rr = ty; rr < 64; rr += 4) {
        const int64_t kpos = k0 + rr;
        const int t = (int)(kpos / nbp), n = (int)(kpos - (int64_t)t * nbp);
        const int c = c0 + tx;
        float v = 0.f;
        if (t < T && n < nb && c < Cc) v = src[(int64_t)t * tstride + (int64_t)n * ld + c];
        tile[rr][tx] = v;
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        const int c = c0 + cc;
        const int64_t kpos = k0 + tx;
        if (c < Cc && kpos < Kp) {
            unsigned short h, m, l;
            split3(tile[tx][cc], h, m, l);
            const int64_t o = (int64_t)c * pitch + (kpos >> 5) * 128 + (kpos & 31) * 2;
            *reinterpret_cast<unsigned short*>(q1 + o) = m;
            *reinterpret_cast<unsigned short*>(q1 + o + 64) = h;
            *reinterpret_cast<unsigned short*>(q2 + o) = h;
            *reinterpret_cast<unsigned short*>(q2 + o + 64) = l;
        }
    }
}

// planes[r][k] = split(src[r*ld + k] * (rs ? rs[r % nb] : 1)), k < K; zero for K <= k < Kp.  8 elements per thread.
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ src, int64_t ld, int64_t R, int K, int Kp,
                                                         char* __restrict__ planes, const float* __restrict__ rs, int nb) {
    const int cpr = Kp / 8;  // 16-byte output chunks per row
    const int64_t total = R * cpr;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int k = (int)(i - r * cpr) * 8;
        float v[8];
        const float* s = src + r * ld + k;
        if (vec && k + 8 <= K) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (k + e < K) ? s[e] : 0.f;
        }
        const float sc = rs ? rs[r % nb] : 1.f;
        unsigned short h[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split2(v[e] * sc, h[e], l[e]);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 hv = {h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16)};
        const u32x4 lv = {l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16), l[4] | ((unsigned)l[5] << 16), l[6] | ((unsigned)l[7] << 16)};
        char* o = planes + r * (int64_t)Kp * 4 + (k >> 5) * 128 + (k & 31) * 2;
        *reinterpret_cast<u32x4*>(o) = hv;
        *reinterpret_cast<u32x4*>(o + 64) = lv;
    }
}

// out[r][k] = a[r][k] + b[r][k] (+ c[r][k]) AND its operand planes in one pass: the direction sum + residual of a recurrent
// layer (model.py:85,104,223-226) is the next layer's input, i.e. the A operand of its input projection.
__global__ __launch_bounds__(256) void add3_planes_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                          float* __restrict__ out, int64_t R, int K, int Kp, char* __restrict__ planes) {
    const int cpr = Kp / 8;
    const int64_t total = R * cpr;
    const bool vec = (K & 3) == 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int k = (int)(i - r * cpr) * 8;
        float v[8];
        const int64_t o = r * K + k;
        if (vec && k + 8 <= K) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 x = *reinterpret_cast<const f32x4*>(a + o + 4 * h) + *reinterpret_cast<const f32x4*>(b + o + 4 * h);
                if (c) x += *reinterpret_cast<const f32x4*>(c + o + 4 * h);
                *reinterpret_cast<f32x4*>(out + o + 4 * h) = x;
                v[4 * h] = x[0]; v[4 * h + 1] = x[1]; v[4 * h + 2] = x[2]; v[4 * h + 3] = x[3];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = 0.f;
                if (k + e < K) {
                    v[e] = a[o + e] + b[o + e] + (c ? c[o + e] : 0.f);
                    out[o + e] = v[e];
                }
            }
        }
        unsigned short h[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split2(v[e], h[e], l[e]);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 hv = {h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16)};
        const u32x4 lv = {l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16), l[4] | ((unsigned)l[5] << 16), l[6] | ((unsigned)l[7] << 16)};
        char* po = planes + r * (int64_t)Kp * 4 + (k >> 5) * 128 + (k & 31) * 2;
        *reinterpret_cast<u32x4*>(po) = hv;
        *reinterpret_cast<u32x4*>(po + 64) = lv;
    }
}

// Transposing split of a time-major matrix: src[(t*nb + n)*ld + c] -> planes[c][t*nbp + n] for c < Cc (plane rows, pitch
// Kp); the pad columns (n >= nb inside a time block, and everything from T*nbp to Kp) are written as zeros.  One
// workgroup transposes a 64 (k positions) x 64 (source columns) tile through LDS, so both the fp32 reads and the bf16
// writes are full 128-byte lines; rs[n] scales source row (t, n).
__global__ __launch_bounds__(256) void split_rows_t_kernel(const float* __restrict__ src, int64_t ld, int T, int nb, int nbp, int Cc,
                                                           int64_t Kp, char* __restrict__ planes, const float* __restrict__ rs,
                                                           int64_t tstride) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64;
    const int64_t k0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int rr = ty; rr < 64; rr += 4) {
        const int64_t kpos = k0 + rr;
        const int t = (int)(kpos / nbp), n = (int)(kpos - (int64_t)t * nbp);
        const int c = c0 + tx;
        float v = 0.f;
        if (t < T && n < nb && c < Cc) v = src[(int64_t)t * tstride + (int64_t)n * ld + c] * (rs ? rs[n] : 1.f);
        tile[rr][tx] = v;
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        const int c = c0 + cc;
        const int64_t kpos = k0 + tx;
        if (c < Cc && kpos < Kp) {
            unsigned short h, l;
            split2(tile[tx][cc], h, l);
            char* o = planes + (int64_t)c * Kp * 4 + (kpos >> 5) * 128 + (kpos & 31) * 2;
            *reinterpret_cast<unsigned short*>(o) = h;
            *reinterpret_cast<unsigned short*>(o + 64) = l;
        }
    }
}

// The same transposition with PLANES as the source (row-major operand planes written by the BPTT kernels): element
// (row (t, n), column c) = hi + lo (exact in fp32), times rs[n], re-split into planes[c][t*nbp + n].
__global__ __launch_bounds__(256) void planes_t_kernel(const char* __restrict__ src, int64_t ldp, int T, int nb, int nbp, int Cc,
                                                       int64_t Kp, char* __restrict__ planes, const float* __restrict__ rs) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64;
    const int64_t k0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int rr = ty; rr < 64; rr += 4) {
        const int64_t kpos = k0 + rr;
        const int t = (int)(kpos / nbp), n = (int)(kpos - (int64_t)t * nbp);
        const int c = c0 + tx;
        float v = 0.f;
        if (t < T && n < nb && c < Cc) {
            const char* e = src + ((int64_t)t * nb + n) * ldp * 4 + (c >> 5) * 128 + (c & 31) * 2;
            const unsigned h = *reinterpret_cast<const unsigned short*>(e), l = *reinterpret_cast<const unsigned short*>(e + 64);
            v = (__uint_as_float(h << 16) + __uint_as_float(l << 16)) * (rs ? rs[n] : 1.f);
        }
        tile[rr][tx] = v;
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        const int c = c0 + cc;
        const int64_t kpos = k0 + tx;
        if (c < Cc && kpos < Kp) {
            unsigned short h, l;
            split2(tile[tx][cc], h, l);
            char* o = planes + (int64_t)c * Kp * 4 + (kpos >> 5) * 128 + (kpos & 31) * 2;
            *reinterpret_cast<unsigned short*>(o) = h;
            *reinterpret_cast<unsigned short*>(o + 64) = l;
        }
    }
}

}  // namespace

extern "C" int aas_add3_planes_f32(aasStream_t stream, float* out, const float* a, const float* b, const float* c, int64_t rows, int K, int Kp,
                                   void* planes) {
    AAS_CHECK(out && a && b && planes && rows >= 0 && K >= 1 && Kp >= K && Kp % 32 == 0, "aas_add3_planes_f32: bad arguments (K=%d Kp=%d)", K, Kp);
    AAS_CHECK(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0,
              "aas_add3_planes_f32: operands must be 16-byte aligned");
    if (rows == 0) return 0;
    const int64_t total = rows * (Kp / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(add3_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, out, rows, K, Kp, (char*)planes);
    AAS_LAUNCH_CHECK("aas_add3_planes_f32");
    return 0;
}

extern "C" int aas_planes_transpose(aasStream_t stream, const void* src_planes, int64_t src_ld, int T, int nb, int nbp, int C, int64_t Kp,
                                    void* planes, const float* row_scale) {
    AAS_CHECK(src_planes && planes && T >= 1 && nb >= 1 && nbp >= nb && nbp % 8 == 0 && C >= 1 && src_ld % 32 == 0 && src_ld >= C &&
                  Kp >= (int64_t)T * nbp && Kp % 32 == 0,
              "aas_planes_transpose: bad arguments (T=%d nb=%d nbp=%d C=%d ld=%lld Kp=%lld)", T, nb, nbp, C, (long long)src_ld, (long long)Kp);
    hipLaunchKernelGGL(planes_t_kernel, dim3(cdiv(C, 64), (unsigned)((Kp + 63) / 64)), dim3(256), 0, (hipStream_t)stream, (const char*)src_planes,
                       src_ld, T, nb, nbp, C, Kp, (char*)planes, row_scale);
    AAS_LAUNCH_CHECK("aas_planes_transpose");
    return 0;
}

extern "C" int aas_gemm_planes(aasStream_t stream, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb,
                               float* C, int64_t ldc, const float* bias, const float* addend, int64_t ldd, int accumulate, int batch,
                               int64_t strideA, int64_t strideB, int64_t strideC) {
    AAS_CHECK(M >= 0 && N >= 0 && K >= 0 && batch >= 1, "aas_gemm_planes: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, batch);
    AAS_CHECK(A && B && C, "aas_gemm_planes: null operand");
    AAS_CHECK(K % 32 == 0 && lda % 32 == 0 && ldb % 32 == 0 && strideA % 32 == 0 && strideB % 32 == 0,
              "aas_gemm_planes: K, lda, ldb and the batch strides must be multiples of 32 (K=%d lda=%lld ldb=%lld)", K,
              (long long)lda, (long long)ldb);
    AAS_CHECK(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 127) == 0, "aas_gemm_planes: planes must be 128-byte aligned");
    if (M == 0 || N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    PG p;
    p.A = (const char*)A; p.B = (const char*)B;
    p.C = C; p.bias = bias; p.addend = addend; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldd = ldd;
    p.accumulate = accumulate; p.batch = batch; p.sA = strideA; p.sB = strideB; p.sC = strideC; p.splitk = 1;
    p.flags = aas_debug_flags_value();
    p.multi = 0;
    // 256 x 256 tiles (8 waves) halve the operand bytes per flop that the 128 x 128 kernel pulls through L2 (its bound);
    // they are used when they still give every CU about a tile, else 128 x 128 with split-K when K is deep
    const int64_t big_tiles = (int64_t)cdiv(M, 256) * cdiv(N, 256) * batch;
    const bool big = big_tiles >= 192 && !(p.flags & 1024);
    const int bm = big ? 256 : 128, bn = big ? 256 : 128;
    dim3 grid(cdiv(N, bn), cdiv(M, bm), batch);
    const int blocks = grid.x * grid.y * batch;
    // split K only when the tiles alone cannot occupy even the half of the chip a lane of the step leaves free: the atomic
    // epilogue (+ the memset in front of it) costs more than a second round of whole tiles
    if (!big && blocks < 128 && K >= 1024) {
        int want = (512 + blocks - 1) / blocks;
        const int maxs = K / 256;
        int sk = want < maxs ? want : maxs;
        if (sk > 16) sk = 16;
        if (sk > 1) {
            p.splitk = sk;
            grid.z = batch * sk;
            if (!accumulate) {
                for (int b = 0; b < batch; ++b) {
                    float* cb = C + (int64_t)b * strideC;
                    if (ldc == N) { AAS_HIP(hipMemsetAsync(cb, 0, sizeof(float) * (size_t)M * N, s)); }
                    else { AAS_HIP(hipMemset2DAsync(cb, sizeof(float) * ldc, 0, sizeof(float) * N, M, s)); }
                }
            }
        }
    }
    int rc;
    // 128 x 256 tiles (debug bit 2048; OFF by default): half the re-reads of the tall operand and one round of tiles instead of
    // two on half a chip, but measured SLOWER inside the step (19.7 vs 19.0 ms: 98 KB of LDS = one workgroup per CU)
    const bool wide = !big && p.splitk == 1 && N >= 256 && (int64_t)cdiv(M, 128) * cdiv(N, 256) * batch >= 90 && (p.flags & 2048);
    if (wide) { grid = dim3(cdiv(N, 256), cdiv(M, 128), batch); rc = launch_planes<128, 256, 2, 2, true>(p, grid, s); }
    else if (big) rc = launch_planes<256, 256, 4, 2, true>(p, grid, s);
    else if (p.splitk > 1) rc = launch_planes<128, 128, 2, 2, false>(p, grid, s);
    // fewer than two tiles per CU: eight waves per workgroup (two per SIMD from ONE workgroup, wave tile 32 x 64) - with a
    // single wave per SIMD nothing covers that wave's LDS and barrier waits (debug bit 131072: the four-wave form)
    else if (blocks < 512 && !(p.flags & 131072)) rc = launch_planes<128, 128, 4, 2, true>(p, grid, s);
    else rc = launch_planes<128, 128, 2, 2, true>(p, grid, s);
    AAS_CHECK(rc == 0, "aas_gemm_planes: could not raise the dynamic LDS limit");
    AAS_LAUNCH_CHECK("aas_gemm_planes");
    return 0;
}

// `count` (<= 4) independent products of one shape in ONE launch, each with its own operand / result pointers: the four
// weight-gradient products of a bidirectional recurrent layer (dW_ih, dW_hh per direction) fill the chip together
// instead of four quarter-size launches.  Always accumulates into C (weight gradients add into the flat buffers).
extern "C" int aas_gemm_planes_multi(aasStream_t stream, int M, int N, int K, int count, const void* const* h_A,
                                     const void* const* h_B, float* const* h_C, int64_t lda, int64_t ldb, int64_t ldc) {
    AAS_CHECK(M >= 0 && N >= 0 && K >= 0 && count >= 1 && count <= 4, "aas_gemm_planes_multi: bad sizes M=%d N=%d K=%d count=%d", M, N, K, count);
    AAS_CHECK(h_A && h_B && h_C, "aas_gemm_planes_multi: null pointer table");
    AAS_CHECK(K % 32 == 0 && lda % 32 == 0 && ldb % 32 == 0, "aas_gemm_planes_multi: K, lda, ldb must be multiples of 32");
    if (M == 0 || N == 0 || K == 0) return 0;
    PG p = {};
    for (int i = 0; i < count; ++i) {
        AAS_CHECK(h_A[i] && h_B[i] && h_C[i], "aas_gemm_planes_multi: null operand %d", i);
        AAS_CHECK(((reinterpret_cast<uintptr_t>(h_A[i]) | reinterpret_cast<uintptr_t>(h_B[i])) & 127) == 0, "aas_gemm_planes_multi: planes must be 128-byte aligned");
        p.Am[i] = (const char*)h_A[i]; p.Bm[i] = (const char*)h_B[i]; p.Cm[i] = h_C[i];
    }
    p.multi = count;
    p.A = p.Am[0]; p.B = p.Bm[0]; p.C = p.Cm[0];
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldd = 0; p.accumulate = 1; p.batch = count; p.splitk = 1;
    p.flags = aas_debug_flags_value();
    hipStream_t s = (hipStream_t)stream;
    const int64_t big_tiles = (int64_t)cdiv(M, 256) * cdiv(N, 256) * count;
    const bool big = big_tiles >= 192 && !(p.flags & 1024);
    const int bm = big ? 256 : 128, bn = big ? 256 : 128;
    dim3 grid(cdiv(N, bn), cdiv(M, bm), count);
    const int blocks = grid.x * grid.y * count;
    if (!big && blocks < 192 && K >= 1024) {     // too few tiles to fill the chip: split K (atomic accumulation)
        int want = (384 + blocks - 1) / blocks;
        const int maxs = K / 512;
        int sk = want < maxs ? want : maxs;
        if (sk > 16) sk = 16;
        if (sk > 1) { p.splitk = sk; grid.z = count * sk; }
    }
    int rc;
    const bool wide = !big && p.splitk == 1 && N >= 256 && (int64_t)cdiv(M, 128) * cdiv(N, 256) * count >= 90 && (p.flags & 2048);
    if (wide) { grid = dim3(cdiv(N, 256), cdiv(M, 128), count); rc = launch_planes<128, 256, 2, 2, true>(p, grid, s); }
    else if (big) rc = launch_planes<256, 256, 4, 2, true>(p, grid, s);
    else if (p.splitk > 1) rc = launch_planes<128, 128, 2, 2, false>(p, grid, s);
    else rc = launch_planes<128, 128, 2, 2, true>(p, grid, s);
    AAS_CHECK(rc == 0, "aas_gemm_planes_multi: could not raise the dynamic LDS limit");
    AAS_LAUNCH_CHECK("aas_gemm_planes_multi");
    return 0;
}

extern "C" int aas_split_planes(aasStream_t stream, const float* src, int64_t ld, int64_t rows, int K, int Kp, void* planes,
                                const float* row_scale, int nb) {
    AAS_CHECK(src && planes && rows >= 0 && K >= 0 && Kp >= K && Kp % 32 == 0, "aas_split_planes: bad arguments (K=%d Kp=%d)", K, Kp);
    AAS_CHECK(!row_scale || nb > 0, "aas_split_planes: row_scale needs nb > 0");
    if (rows == 0 || Kp == 0) return 0;
    const int64_t total = rows * (Kp / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, ld, rows, K, Kp,
                       (char*)planes, row_scale, nb > 0 ? nb : 1);
    AAS_LAUNCH_CHECK("aas_split_planes");
    return 0;
}

static int split_planes_t_impl(aasStream_t stream, const float* src, int64_t ld, int T, int nb, int nbp, int C, int64_t Kp, void* planes,
                               const float* row_scale, int64_t tstride) {
    AAS_CHECK(src && planes && T >= 1 && nb >= 1 && nbp >= nb && nbp % 8 == 0 && C >= 1 && Kp >= (int64_t)T * nbp && Kp % 32 == 0,
              "aas_split_planes_t: bad arguments (T=%d nb=%d nbp=%d C=%d Kp=%lld)", T, nb, nbp, C, (long long)Kp);
    hipLaunchKernelGGL(split_rows_t_kernel, dim3(cdiv(C, 64), (unsigned)((Kp + 63) / 64)), dim3(256), 0, (hipStream_t)stream, src, ld, T, nb, nbp,
                       C, Kp, (char*)planes, row_scale, tstride != 0 ? tstride : (int64_t)nb * ld);
    AAS_LAUNCH_CHECK("aas_split_planes_t");
    return 0;
}

extern "C" int aas_split_planes_t(aasStream_t stream, const float* src, int64_t ld, int T, int nb, int nbp, int C, int64_t Kp, void* planes,
                                  const float* row_scale) {
    return split_planes_t_impl(stream, src, ld, T, nb, nbp, C, Kp, planes, row_scale, 0);
}

// same, with the T blocks of nb rows `tstride` elements apart (blocks that are separate tensors of one flat buffer:
// the two directions' W_ih as the transposed operand of the input-gradient product)
extern "C" int aas_split_planes_t2(aasStream_t stream, const float* src, int64_t ld, int64_t tstride, int T, int nb, int nbp, int C,
                                   int64_t Kp, void* planes, const float* row_scale) {
    AAS_CHECK(tstride != 0, "aas_split_planes_t2: tstride must be non-zero (it may be negative)");
    return split_planes_t_impl(stream, src, ld, T, nb, nbp, C, Kp, planes, row_scale, tstride);
}

// ---- producers of the three-term plane sets (aas_set_precision(2): see split3 above) --------------------------------------------
extern "C" int aas_split_planes3(aasStream_t stream, const float* src, int64_t ld, int64_t rows, int K, int Kp, void* planes_q1, void* planes_q2,
                                 int64_t row_pitch_bytes) {
    AAS_CHECK(src && planes_q1 && planes_q2 && rows >= 0 && K >= 0 && Kp >= K && Kp % 32 == 0, "aas_split_planes3: bad arguments (K=%d Kp=%d)", K, Kp);
    const int64_t pitch = row_pitch_bytes > 0 ? row_pitch_bytes : (int64_t)Kp * 4;
    AAS_CHECK(pitch >= (int64_t)Kp * 4 && pitch % 128 == 0, "aas_split_planes3: row pitch must be a multiple of 128 bytes and hold Kp columns");
    if (rows == 0 || Kp == 0) return 0;
    const int64_t total = rows * (Kp / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_rows3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, ld, rows, K, Kp, (char*)planes_q1,
                       (char*)planes_q2, pitch);
    AAS_LAUNCH_CHECK("aas_split_planes3");
    return 0;
}

extern "C" int aas_add3_planes3_f32(aasStream_t stream, float* out, const float* a, const float* b, const float* c, int64_t rows, int K, int Kp,
                                    void* planes_q1, void* planes_q2, int64_t row_pitch_bytes) {
    const int64_t pitch = row_pitch_bytes > 0 ? row_pitch_bytes : (int64_t)Kp * 4;
    AAS_CHECK(pitch >= (int64_t)Kp * 4 && pitch % 128 == 0, "aas_add3_planes3_f32: row pitch must be a multiple of 128 bytes and hold Kp columns");
    AAS_CHECK(out && a && b && planes_q1 && planes_q2 && rows >= 0 && K >= 1 && Kp >= K && Kp % 32 == 0, "aas_add3_planes3_f32: bad arguments (K=%d Kp=%d)", K, Kp);
    AAS_CHECK(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0,
              "aas_add3_planes3_f32: operands must be 16-byte aligned");
    if (rows == 0) return 0;
    const int64_t total = rows * (Kp / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(add3_planes3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, out, rows, K, Kp,
                       (char*)planes_q1, (char*)planes_q2, pitch);
    AAS_LAUNCH_CHECK("aas_add3_planes3_f32");
    return 0;
}

extern "C" int aas_split_planes_t3(aasStream_t stream, const float* src, int64_t ld, int64_t tstride, int T, int nb, int nbp, int C, int64_t Kp,
                                   void* planes_q1, void* planes_q2, int64_t row_pitch_bytes) {
    const int64_t pitch = row_pitch_bytes > 0 ? row_pitch_bytes : Kp * 4;
    AAS_CHECK(pitch >= Kp * 4 && pitch % 128 == 0, "aas_split_planes_t3: row pitch must be a multiple of 128 bytes and hold Kp columns");
    AAS_CHECK(src && planes_q1 && planes_q2 && T >= 1 && nb >= 1 && nbp >= nb && nbp % 8 == 0 && C >= 1 && Kp >= (int64_t)T * nbp && Kp % 32 == 0,
              "aas_split_planes_t3: bad arguments (T=%d nb=%d nbp=%d C=%d Kp=%lld)", T, nb, nbp, C, (long long)Kp);
    hipLaunchKernelGGL(split_rows_t3_kernel, dim3(cdiv(C, 64), (unsigned)((Kp + 63) / 64)), dim3(256), 0, (hipStream_t)stream, src, ld, T, nb, nbp,
                       C, Kp, (char*)planes_q1, (char*)planes_q2, tstride != 0 ? tstride : (int64_t)nb * ld, pitch);
    AAS_LAUNCH_CHECK("aas_split_planes_t3");
    return 0;
}
