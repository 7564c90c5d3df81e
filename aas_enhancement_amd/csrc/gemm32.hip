// fp32 MFMA GEMM, LDS-DMA form (gfx950): the fast path of aas_gemm_f32 for operands that take 16-byte loads.
//
// gemm.hip's kernel stages every k-tile through VGPRs (global_load -> ds_write, four 4-byte LDS stores per 16-byte load for
// a k-contiguous operand), reads one ds_read_b32 per MFMA operand, meets at a barrier every 16 k and finishes split-K with
// one atomicAdd per element: 80-90 TFLOP/s alone on the step's shapes, 0.36 of the fp32 MFMA roof inside the step.  Here:
//   * tile 128 x 128 x 32, 256 threads (2 x 2 waves, 64 x 64 per wave as 2 x 2 v_mfma_f32_32x32x2_f32 tiles: 64 MFMAs per
//     wave and k-step, 4096 matrix-pipe cycles between two barriers);
//   * both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no LDS stores, and the next
//     k-step's 32 KB stay in flight across the whole compute phase of the current one (two stages, 64 KB: two workgroups per CU);
//   * a k-contiguous operand ([rows][K]) lands as [row][8 chunks of 4 k] with the chunk index XOR-swizzled on the SOURCE side
//     (the LDS image of an LDS-DMA is lane-linear), so one conflict-free ds_read_b128 hands a lane 4 k of its row: one LDS
//     read per FOUR MFMA operands; a row-contiguous operand ([K][cols]) lands as it lies, [32 k][128 cols], and a lane reads
//     two neighbouring columns with one ds_read_b64 - its two MFMA tiles interleave columns (col = 2 j + u), which also
//     makes the epilogue stores 8 bytes wide;
//   * edges cost nothing in the loop: a lane whose chunk lies outside the matrix (row / column tail, k tail, split-K end)
//     fetches from a zero block instead;
//   * split-K writes fp32 partial slabs with plain stores and a second launch sums them (no atomics: at 1.3 TB/s of atomic
//     bytes the old epilogue cost a fifth of a weight-gradient launch);
//   * XCD-aware tile order: XCD x (workgroups x, x+8, ...) takes a contiguous run of tiles, n fastest, so an A row panel is
//     fetched by one XCD and B stays in that XCD's L2.
// The k order inside a 32-k step is permuted identically for both operands (k = 8 g + 4 (lane>>5) + j): a sum-order change
// at fp32 rounding level against the k-ordered chain of gemm.hip, like any other tiling.
// Replaces cuBLAS / cuDNN GEMMs of the reference (Speech_enhancement_by_AAS/model.py:73-74,94-95,216-217,289,297,317).
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "common.h"

int aas_gemm_max_steps_value();

namespace {

constexpr int TN_ = 128, TK = 32;
constexpr int OPB = 128 * TK * 4;      // bytes of the B image (16 KB); the A image is BM rows (or columns) of 128 bytes
constexpr int NSTAGE = 2;

struct G32 {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* addend;
    int M, N, K;
    int64_t lda, ldb, ldc, ldd;
    int accumulate, splitk, batch;
    int64_t sA, sB, sC;
    int kdivA;
    int64_t kouterA;
    int kdivB;
    int64_t kouterB;
    float* ws;            // split-K: slab z at ws + z * M * N ([M][N], ld = N)
    const float* zero;    // >= 16 bytes of zeros
    int gx, gy, per, tiles;
    int flags;
    int multi;            // > 0: `multi` problems of equal shape, operands / results from the arrays below (z = problem)
    const float* Am[4];
    const float* Bm[4];
    float* Cm[4];
    int Km[4];            // multi: reduction extent per problem (<= K)
    const float* alpha;   // device scalar: the product is multiplied by it before bias / addend / accumulate (null: 1)
};

// LDS-DMA as inline assembly: hipcc does not then know that the instruction writes LDS.  With the builtin it orders every LDS
// read whose memory operand it cannot tell apart from the DMA's target behind `s_waitcnt vmcnt(0)` - the merged ds_read2st64_b64
// reads of a row-contiguous operand got that wait right behind the issue of the NEXT stage's loads, i.e. the loads never overlapped
// the MFMAs (2000 x 500 x 6000 TN, one workgroup per CU: 2.7 us per k-step against 1.9 without loads).  Ordering is explicit here:
// `s_waitcnt vmcnt(0)` + workgroup barrier at the head of every k-step.  (M0 = LDS byte address of the wave's 1-KB piece.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16(const void* g, unsigned lds) {
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

// One source position of an LDS-DMA instruction, advanced by one k-step (32 k) per call of next().
//  k-contiguous operand: a 16-byte chunk (4 k) of one row; row-contiguous: 4 columns of one k row, the k rows two-level addressed
//  (row(r) = (r / kdiv) * kouter + (r % kdiv) * ld) without a division per step: the remainder is carried along (kdiv >= 32).
struct Src {
    const float* p;   // current source (may be anywhere when !ok)
    int k;            // k index of the chunk's first element
    int rem;          // row-contiguous, two-level: k % kdiv
    bool ok;          // inside the matrix in the non-k dimension
};

// AKC / BKC: the operand is k-contiguous ([rows][K]); otherwise row-contiguous ([K][cols], the k rows two-level addressed).
// TMW: rows of a wave's tile (64: 128-row workgroup tile, 64 KB of LDS, two workgroups per CU; 32: 64-row tile, 48 KB, three per
// CU - twice the tiles for products whose 128-row tiles would fill the chip's workgroup slots 1.5 times, i.e. half of them idle in
// the second round).
template <bool AKC, bool BKC, int TMW>
__global__ __launch_bounds__(256, TMW == 64 ? 2 : 3) void gemm32_kernel(G32 p) {
    constexpr int TM = 2 * TMW, MT = TMW / 32;          // workgroup tile rows; 32-row MFMA tiles per wave along M
    constexpr int AB = TM * 128, STAGE = AB + OPB;      // A image bytes; stage = A | B
    constexpr int AI = TM / 32;                         // LDS-DMA instructions per thread for the A image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x, by = blockIdx.y, z = blockIdx.z;
    if (p.per > 0) {
        const int L = blockIdx.x, q = (L & 7) * p.per + (L >> 3);
        if (q >= p.tiles) return;
        bx = q % p.gx;
        const int t = q / p.gx;
        by = t % p.gy;
        z = t / p.gy;
    }
    const int m0 = by * TM, n0 = bx * TN_;
    int kbeg = 0, kend = p.K;
    const float* A = p.A;
    const float* B = p.B;
    float* C = p.C;
    const float* addend = p.addend;
    int kz = 0, slab = 0;
    if (p.multi > 0) {     // z = problem * splitk + k slice
        const int pr = z / p.splitk;
        kz = z - pr * p.splitk;
        A = p.Am[pr]; B = p.Bm[pr]; C = p.Cm[pr];
        kend = p.Km[pr];
    } else if (p.splitk > 1) {
        kz = z;
    }
    slab = z;
    if (p.splitk > 1) {
        const int ktiles = (p.K + TK - 1) / TK;     // (slices are cut on the longest problem's k extent: equal work per slice)
        const int per = (ktiles + p.splitk - 1) / p.splitk;
        kbeg = kz * per * TK;
        kend = min(kend, (kz + 1) * per * TK);
    } else if (p.multi > 0) {
    } else if (p.batch > 1) {
        A += (int64_t)z * p.sA;
        B += (int64_t)z * p.sB;
        C += (int64_t)z * p.sC;
        if (addend) addend += (int64_t)z * p.sC;
    }

    // ---- LDS-DMA sources.  Instruction i of a thread fills 16-byte position q = i*256 + tid of an operand image.
    //  k-contiguous:  q -> (row = q>>3, slot = q&7), the lane fetches source chunk c = slot ^ ((row>>1)&7) of that row (4 k each)
    //  row-contiguous: q -> (k row = q / chunks per row, column chunk = q % chunks per row)
    const int kdA = p.kdivA > 0 ? p.kdivA : 0x3fffffff, kdB = p.kdivB > 0 ? p.kdivB : 0x3fffffff;
    const int64_t wrapA = p.kouterA - (int64_t)kdA * p.lda, wrapB = p.kouterB - (int64_t)kdB * p.ldb;
    auto init = [&](Src& s, bool kc, const float* X, int64_t ld, int x0, int xmax, int q, int cpr, int kdiv, int64_t kouter) {
        if (kc) {
            const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
            s.ok = x0 + row < xmax;
            s.k = kbeg + 4 * c;
            s.rem = 0;
            s.p = X + (int64_t)(x0 + row) * ld + s.k;
        } else {
            const int ch = q % cpr;
            s.ok = x0 + 4 * ch < xmax;
            s.k = kbeg + q / cpr;
            s.rem = s.k % kdiv;
            s.p = X + (int64_t)(s.k / kdiv) * kouter + (int64_t)s.rem * ld + x0 + 4 * ch;
        }
    };
    Src sa_[AI], sb_[4];
#pragma unroll
    for (int i = 0; i < AI; ++i) init(sa_[i], AKC, A, p.lda, m0, p.M, i * 256 + tid, TM / 4, kdA, p.kouterA);
#pragma unroll
    for (int i = 0; i < 4; ++i) init(sb_[i], BKC, B, p.ldb, n0, p.N, i * 256 + tid, 32, kdB, p.kouterB);
    const float* zero = p.zero;
    const unsigned lds0 = (unsigned)(size_t)smem + wave * 1024;
    auto fetch = [&](Src& s, bool kc, unsigned dst, int64_t ld, int kdiv, int64_t wrap) {
        const float* g = (s.ok && s.k < kend) ? s.p : zero;
        glds16(g, dst);
        if (p.flags & 1) return;      // ablation: every k-step refetches the tile's first one (sources stay L2-resident)
        s.k += TK;
        if (kc) {
            s.p += TK;
        } else {
            s.p += TK * ld;
            s.rem += TK;
            while (s.rem >= kdiv) { s.rem -= kdiv; s.p += wrap; }      // (kdiv >= 8: at most four turns)
        }
    };
    auto stage = [&](int st) {   // the next k-step of both operands into stage st
        const unsigned base = lds0 + st * STAGE;
#pragma unroll
        for (int i = 0; i < AI; ++i) fetch(sa_[i], AKC, base + i * 4096, p.lda, kdA, wrapA);
#pragma unroll
        for (int i = 0; i < 4; ++i) fetch(sb_[i], BKC, base + AB + i * 4096, p.ldb, kdB, wrapB);
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    // fragment read offsets inside an operand image
    //  k-contiguous: tile t rows w*TMW + t*32 + l31; chunk (2 g + lh) ^ f(row)  -> a lane's 4 k: 8 g + 4 lh + j
    //  row-contiguous: k row 8 g + 4 lh + j, columns w*64 + 2 l31 (+ u)
    int aoff[MT], axr[MT], boff[2], bxr[2];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int ra = wm * TMW + t * 32 + l31;
        aoff[t] = ra * 128; axr[t] = (ra >> 1) & 7;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int rb = wn * 64 + t * 32 + l31;
        boff[t] = rb * 128; bxr[t] = (rb >> 1) & 7;
    }
    const int arc = (4 * lh) * (TM * 4) + (wm * TMW + MT * l31) * 4;     // (one tile: consecutive rows, 4-byte reads)
    const int brc = (4 * lh) * 512 + (wn * 64 + 2 * l31) * 4;

    if (p.flags & 128) kend = kbeg;  // ablation: epilogue only
    if (kbeg < kend) stage(0);
    int st = 0;
    for (int k0 = kbeg; k0 < kend; k0 += TK, st ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // stage st has landed for every wave; everyone is done reading stage st^1
        if (k0 + TK < kend && !(p.flags & 64)) stage(st ^ 1);
        if (p.flags & 16) continue;
        const char* sa = smem + st * STAGE;
        const char* sb = sa + AB;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float a[MT][4], b[2][4];
            if (AKC) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(sa + aoff[t] + (((2 * g + lh) ^ axr[t]) << 4));
                    a[t][0] = v.x; a[t][1] = v.y; a[t][2] = v.z; a[t][3] = v.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (MT == 2) {
                        const float2 v = *reinterpret_cast<const float2*>(sa + arc + (8 * g + j) * (TM * 4));
                        a[0][j] = v.x; a[1][j] = v.y;
                    } else {
                        a[0][j] = *reinterpret_cast<const float*>(sa + arc + (8 * g + j) * (TM * 4));
                    }
                }
            }
            if (BKC) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(sb + boff[u] + (((2 * g + lh) ^ bxr[u]) << 4));
                    b[u][0] = v.x; b[u][1] = v.y; b[u][2] = v.z; b[u][3] = v.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float2 v = *reinterpret_cast<const float2*>(sb + brc + (8 * g + j) * 512);
                    b[0][j] = v.x; b[1][j] = v.y;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], b[u][j], acc[t][u], 0, 0, 0);
        }
    }

    // ---- epilogue through LDS: every wave lays its TMW x 64 sub-tile down row-major in a private 256-byte-pitch region (the
    // stage buffers are free now), then stores whole rows: 16 lanes x 16 bytes = the 256 contiguous bytes of a row per store
    // instruction quarter, bias / addend / accumulate read the same way.
    //  32x32 C map: column index j = lane&31 (B side), row index i = (r&3) + 8 (r>>2) + 4 (lane>>5) (A side);
    //  tile (t, u) covers rows  AKC ? t*32 + i : MT i + t   and columns  BKC ? u*32 + j : 2 j + u   of the wave's TMW x 64.
    __syncthreads();
    char* ew = smem + wave * (TMW * 256);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int row = AKC ? t * 32 + i : MT * i + t;
            if (BKC) {
                *reinterpret_cast<float*>(ew + row * 256 + l31 * 4) = acc[t][0][r];
                *reinterpret_cast<float*>(ew + row * 256 + (32 + l31) * 4) = acc[t][1][r];
            } else {
                *reinterpret_cast<float2*>(ew + row * 256 + l31 * 8) = make_float2(acc[t][0][r], acc[t][1][r]);
            }
        }
    // (a wave reads back only what it wrote itself: no barrier, the LDS queue of a wave is in order)
    const bool part = p.splitk > 1;
    float* Cw = part ? p.ws + (int64_t)slab * p.M * p.N : C;
    const int64_t ldc = part ? p.N : p.ldc;
    const int c4 = (lane & 15) * 4, rq = lane >> 4;
    const int n = n0 + wn * 64 + c4;
    if (n < p.N) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (!part && p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
        const float al = (!part && p.alpha) ? *p.alpha : 1.0f;
#pragma unroll 4
        for (int rr = 0; rr < TMW; rr += 4) {
            const int row = rr + rq, m = m0 + wm * TMW + row;
            if (m >= p.M) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(ew + row * 256 + c4 * 4);
            float* cp = Cw + (int64_t)m * ldc + n;
            if (!part) {
                v = v * al + bv;
                if (addend) v += *reinterpret_cast<const f32x4*>(addend + (int64_t)m * p.ldd + n);
                if (p.accumulate) v += *reinterpret_cast<const f32x4*>(cp);
            }
            *reinterpret_cast<f32x4*>(cp) = v;
        }
    }
}

// C (+)= sum of `ns` slabs [M][N] (+ bias + addend): the second launch of a split-K product
struct R4 {
    float* C[4];
};
__global__ __launch_bounds__(256) void gemm32_reduce_kernel(const float* __restrict__ ws, int ns, int M, int N, R4 cs,
                                                            int64_t ldc, const float* __restrict__ bias, const float* __restrict__ addend,
                                                            int64_t ldd, int accumulate, const float* __restrict__ alpha) {
    const int64_t q4 = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one float4 of the [M][N] slab
    const int n4 = N >> 2;
    if (q4 >= (int64_t)M * n4) return;
    const int m = (int)(q4 / n4), n = (int)(q4 % n4) * 4;
    const int64_t slab = (int64_t)M * N;
    float* __restrict__ C = cs.C[blockIdx.y];      // problem blockIdx.y: slabs [y * ns, (y + 1) * ns)
    ws += (int64_t)blockIdx.y * ns * slab;
    f32x4 s = *reinterpret_cast<const f32x4*>(ws + (int64_t)m * N + n);
    for (int z = 1; z < ns; ++z) s += *reinterpret_cast<const f32x4*>(ws + z * slab + (int64_t)m * N + n);
    if (alpha) s = s * alpha[0];
    // (16-byte accesses: aas_gemm32_try only takes products whose C / addend / bias rows are 16-byte aligned)
    if (bias) s += *reinterpret_cast<const f32x4*>(bias + n);
    float* cp = C + (int64_t)m * ldc + n;
    if (addend) s += *reinterpret_cast<const f32x4*>(addend + (int64_t)m * ldd + n);
    if (accumulate) s += *reinterpret_cast<const f32x4*>(cp);
    *reinterpret_cast<f32x4*>(cp) = s;
}

__device__ float g_zero_block[64];   // zero-initialised: where out-of-range lanes of an LDS-DMA fetch from

// one slab workspace per (device, stream): products on different streams overlap.  A block that is outgrown is retired, never freed
// under a caller's feet (a captured hipGraph of the training step has the block's address in its GEMM and reduce nodes); under
// capture the block cannot grow: the caller then runs the product without split-K.
float* workspace(hipStream_t s, size_t bytes) {
    return static_cast<float*>(aas_stream_workspace(AAS_WS_GEMM_SLABS, s, bytes, (size_t)64 << 20));
}

constexpr int MAX_DEV = 64;
std::mutex g_dev_mu;

// the device address of g_zero_block is a per-device fact (every device has its own copy of the module's globals)
const float* zero_block() {
    static const float* z[MAX_DEV] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (!z[dev]) {
        void* d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_zero_block)) != hipSuccess) return nullptr;
        z[dev] = (const float*)d;
    }
    return z[dev];
}

template <bool AKC, bool BKC, int TMW>
int launch32(G32& p, dim3 grid, hipStream_t s) {
    constexpr int LDS0 = NSTAGE * (2 * TMW * 128 + OPB);
    // experiment (AAS_GEMM32_WHOLE_CU=1): ask for more than half of a CU's LDS, so a workgroup has its CU to itself and hands the
    // WHOLE CU back when it retires - a persistent recurrent launch waiting for residency needs whole CUs
    static const bool whole = aas_ablation_env("AAS_GEMM32_WHOLE_CU") && atoi(aas_ablation_env("AAS_GEMM32_WHOLE_CU")) != 0;
    const int LDS = whole ? 84 * 1024 : LDS0;
    static bool attr_done[MAX_DEV] = {false};     // the raised dynamic-LDS limit is per device too
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return -1;
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        if (!attr_done[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm32_kernel<AKC, BKC, TMW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    84 * 1024) != hipSuccess)
                return -1;
            attr_done[dev] = true;
        }
    }
    p.per = 0;
    if (!(p.flags & 4096)) {   // XCD-aware tile order (debug bit 4096: plain 3-D grid)
        p.gx = grid.x; p.gy = grid.y; p.tiles = grid.x * grid.y * grid.z;
        p.per = (p.tiles + 7) / 8;
        grid = dim3(8 * p.per);
    }
    hipLaunchKernelGGL((gemm32_kernel<AKC, BKC, TMW>), grid, dim3(256), LDS, s, p);
    return 0;
}

// Tile height and split-K factor.  A launch runs in rounds of (CUs x workgroups per CU) workgroups that start and finish
// together; the last round costs a whole round however few workgroups it holds.  Model: time ~ rounds x (work of the workgroups a
// CU holds in a round) + the slab traffic of a split; pick the cheapest of 128- / 64-row tiles x split factors.
double cost_us(int M, int N, int K, int nz, int bm, int s, int* slabs = nullptr) {
    const int cus = aas_device_cus() > 0 ? aas_device_cus() : 256;
    const int ktiles = (K + TK - 1) / TK;
    const int per_cu = bm == 128 ? 2 : 3;
    const int64_t tiles = (int64_t)cdiv(N, TN_) * cdiv(M, bm) * nz;
    const int per = (ktiles + s - 1) / s;
    const int se = (ktiles + per - 1) / per;            // slabs that are not empty
    if (slabs) *slabs = se;
    const int64_t wgs = tiles * se;
    const int64_t slots = (int64_t)cus * per_cu;
    const int64_t rounds = (wgs + slots - 1) / slots;
    // workgroups the busiest CU holds over the launch (full rounds: per_cu each; the last round: dealt round-robin)
    const int64_t last = wgs - (rounds - 1) * slots;
    const int64_t on_cu = (rounds - 1) * per_cu + (last + cus - 1) / cus;
    // one k-step of a 128-row tile holds the CU's matrix pipes for ~2.05 us (4096 cycles at the ~2.0-2.1 GHz the chip holds
    // under fp32 MFMA load); a workgroup alone on its CU covers fewer of its own LDS / barrier waits (measured 0.93 of that
    // rate for 128-row tiles, 0.82 for 64-row ones; 0.97 / 0.90-0.95 with co-resident workgroups); + ~3 us of prologue /
    // epilogue per workgroup; a split adds the slab stores and the reduce launch
    const int co = (int)((wgs + cus - 1) / cus) < per_cu ? (int)((wgs + cus - 1) / cus) : per_cu;
    const double eff = bm == 128 ? (co >= 2 ? 0.97 : 0.93) : (co >= 3 ? 0.95 : co == 2 ? 0.90 : 0.82);
    const double wg_us = per * 2.05 * (bm / 128.0) / eff + 3.0;
    double t = on_cu * wg_us;
    if (se > 1) t += (double)(se + 1) * M * N * 4 / 4.0e6 + 3.0;
    return t;
}

double choose(int M, int N, int K, int nz, bool can_split, int& tmw, int& sk) {
    static const int f_bm = aas_ablation_env("AAS_GEMM32_BM") ? atoi(aas_ablation_env("AAS_GEMM32_BM")) : 0;
    static const int f_sk = aas_ablation_env("AAS_GEMM32_SK") ? atoi(aas_ablation_env("AAS_GEMM32_SK")) : 0;
    // longest life of a workgroup in k-steps (~2 us each): a persistent recurrent launch of the training step becomes resident only
    // when enough CUs are free AT ONCE, so a GEMM beside it must hand its CUs back soon (0 = no cap)
    const int max_steps = aas_gemm_max_steps_value();
    const int ktiles = (K + TK - 1) / TK;
    double best = 1e30;
    tmw = 64; sk = 1;
    for (int bm = 128; bm >= 64; bm -= 64) {
        if (f_bm && bm != f_bm) continue;
        for (int s = 1; s <= 16; ++s) {
            if (s > 1 && (!can_split || ktiles / s < 16)) break;
            if (f_sk && s != f_sk && can_split) continue;
            if (max_steps > 0 && can_split && (ktiles + s - 1) / s > max_steps && s < 16 && ktiles / (s + 1) >= 16) continue;
            int se;
            const double t = cost_us(M, N, K, nz, bm, s, &se);
            if (t < best) { best = t; tmw = bm / 2; sk = se; }
        }
    }
    return best;
}

int g_max_steps = -1;
int g_variant = -1;   // 0: LDS-DMA kernel where it applies (default); 1: always the register-staged kernel of gemm.hip

}  // namespace

extern "C" int aas_set_gemm_max_steps(int n) {
    g_max_steps = n < 0 ? 0 : n;
    return 0;
}

int aas_gemm_max_steps_value();
extern "C" int aas_get_gemm_max_steps(void) { return aas_gemm_max_steps_value(); }

extern "C" int aas_set_gemm_variant(int v) {
    g_variant = v ? 1 : 0;
    return 0;
}

int aas_gemm_max_steps_value() {
    const int scoped = aas_scope_gemm_max_steps();      // a launch scope installed for this thread (aasLaunch.gemm_max_steps)
    if (scoped >= 0) return scoped;
    if (g_max_steps < 0) g_max_steps = aas_ablation_env("AAS_GEMM32_MAXSTEPS") ? atoi(aas_ablation_env("AAS_GEMM32_MAXSTEPS")) : 0;
    return g_max_steps;
}

int aas_gemm_variant_value() {
    if (g_variant < 0) g_variant = (aas_ablation_env("AAS_GEMM32") && atoi(aas_ablation_env("AAS_GEMM32")) == 0) ? 1 : 0;
    return g_variant;
}

// -> 0 launched, 1 error (message set), -1 not applicable (the caller takes the general kernel)
int aas_gemm32_try(hipStream_t s, int mode, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                   int64_t ldc, const float* bias, const float* addend, int64_t ldd, int accumulate, int batch, int64_t strideA,
                   int64_t strideB, int64_t strideC, int kdivA, int64_t kouterA, int kdivB, int64_t kouterB, int nmulti,
                   const float* const* Am, const float* const* Bm, float* const* Cm, const int* Km, const float* d_alpha) {
    if (aas_gemm_variant_value() == 1) return -1;
    if (M < 64 || N < 64 || K < 64) return -1;   // thin products: the general kernel's tile edge handling is as good
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool akc = mode != AAS_GEMM_TN, bkc = mode == AAS_GEMM_NT;
    // 16-byte chunks: k-contiguous operands need K % 4 == 0 (the k tail is cut at chunk granularity), row-contiguous ones a
    // column count that is a multiple of 4; every row start 16-byte aligned
    auto ok_kc = [&](const float* X, int64_t ld, int64_t stride) { return al16(X) && ld % 4 == 0 && K % 4 == 0 && stride % 4 == 0; };
    auto ok_rc = [&](const float* X, int64_t ld, int cols, int kdiv, int64_t kouter, int64_t stride) {
        return al16(X) && ld % 4 == 0 && cols % 4 == 0 && (kdiv == 0 || kouter % 4 == 0) && stride % 4 == 0;
    };
    bool ok = true;
    const int np = nmulti > 0 ? nmulti : 1;
    for (int i = 0; i < np; ++i) {
        const float* Ai = nmulti > 0 ? Am[i] : A;
        const float* Bi = nmulti > 0 ? Bm[i] : B;
        ok = ok && (akc ? (kdivA == 0 && ok_kc(Ai, lda, strideA)) : ok_rc(Ai, lda, M, kdivA, kouterA, strideA));
        ok = ok && (bkc ? (kdivB == 0 && ok_kc(Bi, ldb, strideB)) : ok_rc(Bi, ldb, N, kdivB, kouterB, strideB));
        // a k-contiguous operand is fetched in chunks of 4 k guarded on the chunk's first element: EVERY problem's own reduction
        // extent must be whole chunks, or the products of k = K_i .. K_i | 3 would be added to C_i
        if (nmulti > 0 && Km && (akc || bkc)) ok = ok && Km[i] % 4 == 0;
    }
    // the 16-byte epilogue: whole float4 columns, 16-byte aligned rows of C / addend / bias; two-level k rows step without a division
    ok = ok && N % 4 == 0 && ldc % 4 == 0 && strideC % 4 == 0 && (!bias || al16(bias)) && (!addend || (al16(addend) && ldd % 4 == 0));
    for (int i = 0; i < np; ++i) ok = ok && al16(nmulti > 0 ? Cm[i] : C);
    ok = ok && (kdivA == 0 || kdivA >= 8) && (kdivB == 0 || kdivB >= 8);
    if (!ok) return -1;
    G32 p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.addend = addend;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldd = ldd;
    p.accumulate = accumulate; p.batch = batch; p.sA = strideA; p.sB = strideB; p.sC = strideC;
    p.kdivA = kdivA; p.kouterA = kouterA; p.kdivB = kdivB; p.kouterB = kouterB;
    p.splitk = 1;
    p.flags = aas_debug_flags_value();
    p.zero = zero_block();
    AAS_CHECK(p.zero != nullptr, "aas_gemm_f32: zero block symbol not found");
    p.alpha = d_alpha;
    p.multi = nmulti > 0 ? nmulti : 0;
    for (int i = 0; i < p.multi; ++i) { p.Am[i] = Am[i]; p.Bm[i] = Bm[i]; p.Cm[i] = Cm[i]; p.Km[i] = Km ? Km[i] : K; }
    const int nz = nmulti > 0 ? nmulti : batch;
    const int m_first = M;     // (a two-launch split of the rows - whole rounds of 128-row tiles, then 64-row tiles - was measured: no gain,
                               //  the hardware already deals a partial round one workgroup per CU)
    for (int part = 0; part < (m_first < M ? 2 : 1); ++part) {
        const int mo = part == 0 ? 0 : m_first, Mp = part == 0 ? m_first : M - m_first;
        G32 q = p;
        q.M = Mp;
        if (mo) {
            q.A = akc ? A + (int64_t)mo * lda : A + mo;
            q.C = C + (int64_t)mo * ldc;
            if (addend) q.addend = addend + (int64_t)mo * ldd;
        }
        int tmw, sk;
        if (m_first < M) { tmw = part == 0 ? 64 : 32; sk = 1; }
        else choose(Mp, N, K, nz, batch == 1, tmw, sk);
        dim3 grid(cdiv(N, TN_), cdiv(Mp, 2 * tmw), nz);
        if (sk > 1) {
            const size_t wsb = sizeof(float) * (size_t)sk * nz * Mp * N;
            q.ws = workspace(s, wsb);
            if (q.ws != nullptr) {
                q.splitk = sk;
                grid.z = sk * nz;
            } else {
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                (void)hipStreamIsCapturing(s, &cs);
                AAS_CHECK(cs != hipStreamCaptureStatusNone, "aas_gemm_f32: could not allocate the split-K workspace (%zu bytes)", wsb);
                sk = 1;     // (capturing, and no eager launch on this stream has sized the workspace yet: whole-K workgroups)
            }
        }
        int rc;
        if (tmw == 64) {
            if (akc && bkc) rc = launch32<true, true, 64>(q, grid, s);
            else if (akc) rc = launch32<true, false, 64>(q, grid, s);
            else rc = launch32<false, false, 64>(q, grid, s);
        } else {
            if (akc && bkc) rc = launch32<true, true, 32>(q, grid, s);
            else if (akc) rc = launch32<true, false, 32>(q, grid, s);
            else rc = launch32<false, false, 32>(q, grid, s);
        }
        AAS_CHECK(rc == 0, "aas_gemm_f32: could not raise the dynamic LDS limit");
        if (q.splitk > 1) {
            const int64_t q4 = (int64_t)Mp * (N / 4);
            R4 cs;
            for (int i = 0; i < 4; ++i) cs.C[i] = q.multi > 0 ? q.Cm[i < q.multi ? i : 0] : q.C;
            hipLaunchKernelGGL(gemm32_reduce_kernel, dim3((unsigned)((q4 + 255) / 256), q.multi > 0 ? q.multi : 1), dim3(256), 0, s, q.ws,
                               q.splitk, Mp, N, cs, ldc, bias, q.addend, ldd, accumulate, q.alpha);
        }
    }
    AAS_LAUNCH_CHECK("aas_gemm_f32");
    return 0;
}
