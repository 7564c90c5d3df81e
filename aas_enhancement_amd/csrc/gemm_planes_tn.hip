// Split-bf16 "TN" GEMM on ROW-MAJOR operand planes (gfx950): the weight-gradient products of a recurrent layer,
//     C[m][n] += alpha * sum_r  A[r][m] * B[r][n],      r = (time step, utterance) rows of the layer,
// straight from the planes the layer already holds - d(gates) as the BPTT kernel wrote it ([row][gate column], the A
// operand of the input-gradient GEMM), the layer input as the forward pass split it for the input projection, and h_t as
// the forward recurrent kernel published it in its exchange buffer - so nothing is transposed or re-split in HBM
// (gemm_planes.hip's NT kernel needs k-contiguous operands: planes_t_kernel + three split_rows_t passes per layer).
//
// The reduction index runs down the ROWS of both operands, so an LDS tile is [32 rows][128 columns] (hi | lo) and the
// MFMA fragments (8 consecutive k per lane) are gathered with gfx950's transposing LDS read ds_read_b64_tr_b16: per
// 16-lane group it reads 4 rows x 16 columns of 16-bit elements and hands lane i column i of the 4 rows - two such reads
// give the lane its 8 k values of one column.  Tiles are staged through registers (global_load_dwordx4 -> ds_write_b128);
// every lane computes its own source address, which is where all the generality lives:
//   * row map r' -> (t, n): rows of ONE utterance class (n0 <= n < n0 + Ns of the Nb rows of a time step) are gathered
//     into the reduction, so the two halves of a batched discriminator pass are two problems with their own alpha
//     (the per-utterance weights -kt / 1 of the reference's two D losses) and no operand is ever scaled;
//   * time shifts ta / tb: dW_hh pairs d(gates)[t] with h[t -+ 1];
//   * column base acol0 (multiple of 8): the reverse direction's gate columns start in the middle of a 32-column block;
//   * rows past K and columns past the planes are redirected to a block of zeros (never to the poisoned, i.e. NaN,
//     rows of the exchange buffer: 0 * NaN would poison the sum).
// Bank conflicts: LDS row k stores its 32 16-byte chunks XOR-ed with f(k) = 2*(k&3) ^ 8*((k>>3)&1); the 32 lanes of a
// half-wave read (4 rows x 32 B) x 2 groups = 16 distinct chunks of a 256-byte bank period.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int MAXP = 8;

struct PTN {
    int nprob;
    const char *A[MAXP], *B[MAXP];
    float *C0[MAXP], *C1[MAXP];
    const float* alpha[MAXP];
    int M[MAXP], N[MAXP], K[MAXP], msplit[MAXP], acol0[MAXP], n0[MAXP], ta[MAXP], tb[MAXP];
    int acols[MAXP], bcols[MAXP], gx[MAXP], tile0[MAXP + 1];
    int64_t lda[MAXP], ldb[MAXP], ldc[MAXP];   // lda / ldb: BYTES per plane row; ldc: elements
    int Ns, Nb;
    const char* zero;
    int accumulate, flags, per, tiles, maxwg;
    int ksplit;     // > 1: every tile's k extent is shared by `ksplit` workgroups that add their partial results with atomics
};

template <int PITCH>
__device__ __forceinline__ bf16x8 tr_frag(const char* a) {
    // rows 8g .. 8g+3 then 8g+4 .. 8g+7 (4 LDS rows further on): lane i of the group gets column i
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 4 * PITCH));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ int fsw(int k) { return (2 * (k & 3)) ^ (8 * ((k >> 3) & 1)); }

// BMC x BNC x 32 tile (columns of A / columns of B), WM x WN waves, wave tile (BMC/WM) x (BNC/WN) as MI x NJ MFMA tiles; two LDS
// slots of (A, B) x 32 rows.  128 x 128 with 4 x 2 waves (two waves per SIMD from ONE workgroup: with a single wave per SIMD
// nothing covers that wave's LDS / barrier waits, 0.30 -> 0.20 ms); 256 x 256 with 4 x 2 waves as an opt-in (debug bit
// 8388608): the kernel is bound by what a CU can stage per k-step (32 KB at 128 x 128, ~53 GB/s per CU), and a 256 x 256 tile
// stages 64 KB for four times the flops - but a layer then has only 64 long-lived workgroups (see the launch code).
template <int BMC, int BNC, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_planes_tn_kernel(PTN p) {
    constexpr int NW = WM * WN;
    constexpr int A_P = BMC * 4, B_P = BNC * 4;              // LDS bytes per k row (hi | lo of every column)
    constexpr int TILE_A = 32 * A_P, TILE_Bb = 32 * B_P, STAGE_B = TILE_A + TILE_Bb;
    constexpr int MI = BMC / WM / 16, NJ = BNC / WN / 16;
    constexpr int JA = TILE_A / 1024 / NW, JB = TILE_Bb / 1024 / NW;   // 1-KB staging loads per wave and k-step
    static_assert(TILE_A % (1024 * NW) == 0 && TILE_Bb % (1024 * NW) == 0, "tiles must split into whole wave loads");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD x (workgroups with blockIdx % 8 == x) takes the contiguous run of tiles [x*per, (x+1)*per); with a capped grid a
    // workgroup walks its XCD's run in strides of gridDim/8
    for (int slot = blockIdx.x >> 3; slot < p.per; slot += gridDim.x >> 3) {
    const int qw = (blockIdx.x & 7) * p.per + slot;          // work item = (tile, k share); the shares of a tile are neighbours
    if (qw >= p.tiles * p.ksplit) break;
    const int qt = qw / p.ksplit, kshare = qw - qt * p.ksplit;
    __syncthreads();             // (the previous tile's LDS reads are done before this tile's first writes)
    int pr = 0;
    while (pr + 1 < p.nprob && qt >= p.tile0[pr + 1]) ++pr;
    const int qq = qt - p.tile0[pr];
    const int gx = p.gx[pr];
    const int m0 = (qq / gx) * BMC, n0t = (qq % gx) * BNC;
    const int K = p.K[pr], Ns = p.Ns, Nb = p.Nb;
    // this workgroup's k-steps [ks0, ks1) of the problem's ceil(K / 32)
    const int nk_all = (K + 31) / 32, nk_per = (nk_all + p.ksplit - 1) / p.ksplit;
    const int ks0 = kshare * nk_per, ks1 = min(nk_all, ks0 + nk_per);
    if (ks0 >= ks1) continue;
    const int koff = ks0 * 32;
    const char* Ab = p.A[pr];
    const char* Bb = p.B[pr];
    const int64_t lda = p.lda[pr], ldb = p.ldb[pr];
    const int arow0 = p.ta[pr] * Nb + p.n0[pr], brow0 = p.tb[pr] * Nb + p.n0[pr];

    // ---- staging roles: load j of this wave fills 1 KB of a tile: LDS row kr, 16-byte chunk slot pc.  The rows walk down the
    // operands: per k-step a row index advances by 32 positions of the class, i.e. by 32 + (time-step wraps) * (Nb - Ns) plane
    // rows.  With 1-KB LDS rows (256-column tiles) a load covers ONE row: the row state is wave-uniform (scalar registers) and
    // a lane keeps just its column offset; with 512-byte rows (128 columns) the two halves of a wave are on different rows and
    // every lane walks its own 64-bit pointers.
    const char* zsrc = p.zero + (lane & 31) * 16;
    const int gap = Nb - Ns;
    u32x4 ra_[JA], rb_[JB];
    // per-lane state (512-byte rows)
    const char *pa[JA], *pb[JB];
    unsigned sa[JA], sb_[JB];     // row pitch in bytes, 0 for a column chunk outside the operand (pointer parked on the zero block)
    int na[JA], nb_[JB];
    // wave-uniform state (1-KB rows): plane row index and position inside the time step per load, column byte offset per lane
    int64_t rowa[JA], rowb[JB];
    int offa[JA], offb[JB];
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int byte = (wave * JA + j) * 1024 + lane * 16;
        const int kr = A_P == 1024 ? wave * JA + j : byte / A_P, pc = (byte % A_P) / 16;   // (1-KB rows: wave-uniform row)
        const int c = pc ^ fsw(kr);                        // logical chunk: piece (16 columns), hi / lo, 8-column half
        const int piece = c >> 2, hl = (c >> 1) & 1, half = c & 1;
        const int ca = p.acol0[pr] + m0 + 16 * piece + 8 * half;
        const bool va = (m0 + 16 * piece + 8 * half < p.M[pr]) && (ca + 8 <= p.acols[pr]);
        const int cbyte = (ca >> 5) * 128 + hl * 64 + (ca & 31) * 2;
        const int t0 = (kr + koff) / Ns;
        na[j] = (kr + koff) - t0 * Ns;
        if constexpr (A_P == 1024) {
            rowa[j] = (int64_t)t0 * Nb + na[j] + arow0;
            offa[j] = va ? cbyte : -1;
        } else {
            pa[j] = va ? Ab + ((int64_t)t0 * Nb + na[j] + arow0) * lda + cbyte : zsrc;
            sa[j] = va ? (unsigned)lda : 0u;
        }
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int byte = (wave * JB + j) * 1024 + lane * 16;
        const int kr = B_P == 1024 ? wave * JB + j : byte / B_P, pc = (byte % B_P) / 16;
        const int c = pc ^ fsw(kr);
        const int piece = c >> 2, hl = (c >> 1) & 1, half = c & 1;
        const int cb = n0t + 16 * piece + 8 * half;
        const bool vb = (cb < p.N[pr]) && (cb + 8 <= p.bcols[pr]);
        const int cbyte = (cb >> 5) * 128 + hl * 64 + (cb & 31) * 2;
        const int t0 = (kr + koff) / Ns;
        nb_[j] = (kr + koff) - t0 * Ns;
        if constexpr (B_P == 1024) {
            rowb[j] = (int64_t)t0 * Nb + nb_[j] + brow0;
            offb[j] = vb ? cbyte : -1;
        } else {
            pb[j] = vb ? Bb + ((int64_t)t0 * Nb + nb_[j] + brow0) * ldb + cbyte : zsrc;
            sb_[j] = vb ? (unsigned)ldb : 0u;
        }
    }
    // Register staging (global_load_dwordx4 -> ds_write_b128), one k-step ahead: the tile of step i+1 is written into the
    // other LDS slot right after the barrier of step i, and the loads of step i+2 are issued before the MFMAs of step i, so a
    // load has a whole k-step to land.  (LDS-DMA as in gemm_planes.hip was measured equal; behind a pending global_load_lds
    // the compiler also puts vmcnt(0) in front of the first transposing LDS read.)
    auto load = [&](int k0) {
#pragma unroll
        for (int j = 0; j < JA; ++j) {
            const int kr = A_P == 1024 ? wave * JA + j : ((wave * JA + j) * 1024 + lane * 16) / A_P;
            const bool rok = (k0 + kr) < K && !(p.flags & 64);
            unsigned adv = 32;
            na[j] += 32;                                   // the next k-step of this row
            while (na[j] >= Ns) { na[j] -= Ns; adv += gap; }
            if constexpr (A_P == 1024) {
                const char* rp = Ab + rowa[j] * lda;       // wave-uniform
                ra_[j] = *reinterpret_cast<const u32x4*>((rok && offa[j] >= 0) ? rp + offa[j] : zsrc);
                rowa[j] += adv;
            } else {
                ra_[j] = *reinterpret_cast<const u32x4*>(rok ? pa[j] : zsrc);
                pa[j] += (uint64_t)adv * sa[j];
            }
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int kr = B_P == 1024 ? wave * JB + j : ((wave * JB + j) * 1024 + lane * 16) / B_P;
            const bool rok = (k0 + kr) < K && !(p.flags & 64);
            unsigned adv = 32;
            nb_[j] += 32;
            while (nb_[j] >= Ns) { nb_[j] -= Ns; adv += gap; }
            if constexpr (B_P == 1024) {
                const char* rp = Bb + rowb[j] * ldb;
                rb_[j] = *reinterpret_cast<const u32x4*>((rok && offb[j] >= 0) ? rp + offb[j] : zsrc);
                rowb[j] += adv;
            } else {
                rb_[j] = *reinterpret_cast<const u32x4*>(rok ? pb[j] : zsrc);
                pb[j] += (uint64_t)adv * sb_[j];
            }
        }
    };
    auto put = [&](int st) {
        char* base = smem + st * STAGE_B + lane * 16;
#pragma unroll
        for (int j = 0; j < JA; ++j) *reinterpret_cast<u32x4*>(base + (wave * JA + j) * 1024) = ra_[j];
#pragma unroll
        for (int j = 0; j < JB; ++j) *reinterpret_cast<u32x4*>(base + TILE_A + (wave * JB + j) * 1024) = rb_[j];
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- fragment addresses: group g = lane >> 4 reads LDS rows 8g .. 8g+7; lane 4q'+p' supplies row 8g + q', bytes 8p' ----
    const int i16 = lane & 15, g = lane >> 4, qp = i16 >> 2, pp = i16 & 3;
    const int fq = (2 * qp) ^ (8 * (g & 1));
    int fa[MI][2], fb[NJ][2];
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i][hl] = (8 * g + qp) * A_P + 8 * (pp & 1) + ((((wm * MI + i) * 4 + 2 * hl) ^ fq) + (pp >> 1)) * 16;
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[j][hl] = TILE_A + (8 * g + qp) * B_P + 8 * (pp & 1) + ((((wn * NJ + j) * 4 + 2 * hl) ^ fq) + (pp >> 1)) * 16;
    }

    const int nk = ks1 - ks0;
    if (nk > 0) { load(koff); put(0); }
    if (nk > 1) load(koff + 32);
    for (int i = 0; i < nk; ++i) {
        __syncthreads();   // slot i&1 is complete; everyone is done reading slot (i+1)&1
        if (i + 1 < nk && !(p.flags & 32)) put((i + 1) & 1);
        if (i + 2 < nk && !(p.flags & 128)) load(koff + (i + 2) * 32);
        if (!(p.flags & 16)) {
            const char* sb = smem + (i & 1) * STAGE_B;
            bf16x8 ah[MI], al[MI];
#pragma unroll
            for (int ii = 0; ii < MI; ++ii) {
                ah[ii] = tr_frag<A_P>(sb + fa[ii][0]);
                al[ii] = tr_frag<A_P>(sb + fa[ii][1]);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bf16x8 bh = tr_frag<B_P>(sb + fb[j][0]);
                const bf16x8 bl = tr_frag<B_P>(sb + fb[j][1]);
#pragma unroll
                for (int ii = 0; ii < MI; ++ii) {   // operands swapped: a lane ends up with 4 consecutive columns of one C row
                    acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[ii], acc[ii][j], 0, 0, 0);
                    acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[ii], acc[ii][j], 0, 0, 0);
                    acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[ii], acc[ii][j], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: lane holds C[m][n .. n+3], m = tile row (lane & 15), n = 4 * (lane >> 4) -------------------------------
    const float alpha = p.alpha[pr] ? *p.alpha[pr] : 1.f;
    const int M = p.M[pr], N = p.N[pr], ms = p.msplit[pr];
    const int64_t ldc = p.ldc[pr];
    const bool vec = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.C0[pr]) & 15) == 0) &&
                     (!p.C1[pr] || (reinterpret_cast<uintptr_t>(p.C1[pr]) & 15) == 0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + (wm * MI + i) * 16 + i16;
        if (m >= M) continue;
        float* crow = (m < ms) ? p.C0[pr] + (int64_t)m * ldc : p.C1[pr] + (int64_t)(m - ms) * ldc;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0t + (wn * NJ + j) * 16 + g * 4;
            if (n >= N) continue;
            const f32x4 v = acc[i][j] * alpha;
            float* cp = crow + n;
            if (p.ksplit > 1) {          // shared tile (accumulating launch): partial results meet in atomics
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) atomicAdd(cp + r, v[r]);
            } else if (n + 3 < N && vec) {
                if (p.accumulate) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(cp);
                    *reinterpret_cast<f32x4*>(cp) = o + v;
                } else {
                    *reinterpret_cast<f32x4*>(cp) = v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r >= N) break;
                    if (p.accumulate) cp[r] += v[r];
                    else cp[r] = v[r];
                }
            }
        }
    }
    }   // tiles of this workgroup
}

template <int BMC, int BNC, int WM, int WN>
int launch_tn(PTN& p, const int* h_M, const int* h_N, int count, hipStream_t s) {
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        p.gx[i] = cdiv(h_N[i], BNC);
        p.tile0[i] = tiles;
        tiles += p.gx[i] * cdiv(h_M[i], BMC);
    }
    p.tile0[count] = tiles;
    p.tiles = tiles;
    // k shares per tile (AAS_TN_KSPLIT, default 1; accumulating launches only - the partial results are ADDED with atomics).
    // Two shares bring the fp32-equivalent mode's weight-gradient error from 1.2-1.5x the fp32 mode's down to it (shorter
    // accumulation chains, as the fp32 GEMM's split-K has), but every launch then pays ~40 us of atomics: config-2 step
    // 23.6 -> 25.5 ms (four shares 28.6), fast mode 16.0 -> 17.2 - so it stays an option.
    static const int ks_env = aas_ablation_env("AAS_TN_KSPLIT") ? atoi(aas_ablation_env("AAS_TN_KSPLIT")) : 1;
    int ksplit = ks_env;
    if (ksplit > 8) ksplit = 8;
    if (!p.accumulate || ksplit < 1) ksplit = 1;
    p.ksplit = ksplit;
    p.per = (tiles * ksplit + 7) / 8;
    constexpr int LDS = 2 * 32 * (BMC + BNC) * 4;
    static unsigned char attr_done[AAS_MAX_DEV];
    if (aas_raise_dynamic_lds_once(attr_done, reinterpret_cast<const void*>(&gemm_planes_tn_kernel<BMC, BNC, WM, WN>), LDS)) return -1;
    int grid = 8 * p.per;
    const int cap = aas_wgrad_wg_cap();     // > 0: at most this many workgroups (a multiple of 8), each walking several tiles
    if (cap > 0 && grid > cap) grid = cap < 8 ? 8 : cap / 8 * 8;
    hipLaunchKernelGGL((gemm_planes_tn_kernel<BMC, BNC, WM, WN>), dim3(grid), dim3(64 * WM * WN), LDS, s, p);
    return 0;
}

}  // namespace

extern "C" int aas_gemm_planes_tn(aasStream_t stream, int count, const void* const* h_A, const void* const* h_B, float* const* h_C0,
                                  float* const* h_C1, const float* const* h_alpha, const int* h_M, const int* h_N, const int* h_K,
                                  const int* h_msplit, const int* h_acol0, const int* h_n0, const int* h_ta, const int* h_tb,
                                  const int64_t* h_lda, const int* h_acols, const int64_t* h_ldb, const int* h_bcols,
                                  const int64_t* h_ldc, int Ns, int Nb, const void* zero512, int accumulate) {
    AAS_CHECK(count >= 1 && count <= MAXP, "aas_gemm_planes_tn: 1..%d problems per launch (got %d)", MAXP, count);
    AAS_CHECK(h_A && h_B && h_C0 && h_C1 && h_alpha && h_M && h_N && h_K && h_msplit && h_acol0 && h_n0 && h_ta && h_tb && h_lda &&
                  h_acols && h_ldb && h_bcols && h_ldc && zero512,
              "aas_gemm_planes_tn: null table");
    AAS_CHECK(Ns >= 1 && Nb >= Ns, "aas_gemm_planes_tn: bad row map Ns=%d Nb=%d", Ns, Nb);
    PTN p = {};
    p.nprob = count;
    int wide = 1;
    for (int i = 0; i < count; ++i) {
        AAS_CHECK(h_A[i] && h_B[i] && h_C0[i], "aas_gemm_planes_tn: null operand %d", i);
        AAS_CHECK(h_M[i] >= 1 && h_N[i] >= 1 && h_K[i] >= 0 && h_msplit[i] >= 0 && (h_msplit[i] >= h_M[i] || h_C1[i]),
                  "aas_gemm_planes_tn: bad sizes of problem %d (M=%d N=%d K=%d msplit=%d)", i, h_M[i], h_N[i], h_K[i], h_msplit[i]);
        AAS_CHECK(h_acol0[i] >= 0 && h_acol0[i] % 8 == 0 && h_lda[i] % 128 == 0 && h_ldb[i] % 128 == 0 && h_acols[i] % 32 == 0 &&
                      h_bcols[i] % 32 == 0 && (int64_t)h_acols[i] * 4 <= h_lda[i] && (int64_t)h_bcols[i] * 4 <= h_ldb[i],
                  "aas_gemm_planes_tn: problem %d: column base must be a multiple of 8, row pitches multiples of 128 bytes", i);
        AAS_CHECK(((reinterpret_cast<uintptr_t>(h_A[i]) | reinterpret_cast<uintptr_t>(h_B[i])) & 15) == 0, "aas_gemm_planes_tn: planes must be 16-byte aligned");
        AAS_CHECK(h_n0[i] >= 0 && h_n0[i] + Ns <= Nb && h_ta[i] >= 0 && h_tb[i] >= 0, "aas_gemm_planes_tn: bad row map of problem %d", i);
        p.A[i] = (const char*)h_A[i]; p.B[i] = (const char*)h_B[i]; p.C0[i] = h_C0[i]; p.C1[i] = h_C1[i]; p.alpha[i] = h_alpha[i];
        p.M[i] = h_M[i]; p.N[i] = h_N[i]; p.K[i] = h_K[i]; p.msplit[i] = h_msplit[i]; p.acol0[i] = h_acol0[i];
        p.n0[i] = h_n0[i]; p.ta[i] = h_ta[i]; p.tb[i] = h_tb[i];
        p.lda[i] = h_lda[i]; p.acols[i] = h_acols[i]; p.ldb[i] = h_ldb[i]; p.bcols[i] = h_bcols[i]; p.ldc[i] = h_ldc[i];
        if (h_N[i] < 192 || h_M[i] < 512) wide = 0;          // narrow results keep 128 x 128 tiles
    }
    p.Ns = Ns; p.Nb = Nb;
    p.zero = (const char*)zero512;
    p.accumulate = accumulate;
    p.flags = aas_debug_flags_value();
    // debug bits: 65536 = 128 x 128 tiles with four waves (one per SIMD); 8388608 = 256 x 256 tiles for wide results: 45 % less
    // CU time per product (64 KB staged per k-step for four times the flops) but a quarter of the workgroups, each living
    // 0.5 ms - config 2 with a frozen A 16.33 vs 16.40 ms, with a trainable A 19.6 vs 17.7, AM step 7.28 vs 7.15: off.
    // 33554432 = 256 x 128 tiles (a layer = 128 workgroups): frozen A 15.65 vs 15.73, trainable A 18.0 vs 17.5, AM 7.4 vs 7.0: off
    int rc;
    if (p.flags & 65536) rc = launch_tn<128, 128, 2, 2>(p, h_M, h_N, count, (hipStream_t)stream);
    else if (wide && (p.flags & 8388608)) rc = launch_tn<256, 256, 4, 2>(p, h_M, h_N, count, (hipStream_t)stream);
    else if (wide && (p.flags & 33554432)) rc = launch_tn<256, 128, 4, 2>(p, h_M, h_N, count, (hipStream_t)stream);
    else rc = launch_tn<128, 128, 4, 2>(p, h_M, h_N, count, (hipStream_t)stream);
    AAS_CHECK(rc == 0, "aas_gemm_planes_tn: could not raise the dynamic LDS limit");
    AAS_LAUNCH_CHECK("aas_gemm_planes_tn");
    return 0;
}
