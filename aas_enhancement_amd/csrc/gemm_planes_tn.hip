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
};

__device__ __forceinline__ bf16x8 tr_frag(const char* a) {
    // rows 8g .. 8g+3 then 8g+4 .. 8g+7 (2048 B = 4 LDS rows further on): lane i of the group gets column i
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 2048));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ int fsw(int k) { return (2 * (k & 3)) ^ (8 * ((k >> 3) & 1)); }

// 128 x 128 x 32 tile, NW waves as (NW/2) x 2: wave tile (256/NW) x 64 rows x columns, i.e. MI x 4 MFMA tiles with MI = 16/NW;
// two LDS slots of (A, B) x 32 rows x 512 B.  NW = 8 (two waves per SIMD from ONE workgroup) is the default: a layer's products
// are only ~256 tiles, one per CU, and with a single wave per SIMD nothing covers that wave's LDS / barrier waits.
template <int NW>
__global__ __launch_bounds__(64 * NW) void gemm_planes_tn_kernel(PTN p) {
    constexpr int TILE_B = 32 * 512, STAGE_B = 2 * TILE_B;
    constexpr int MI = 16 / NW;          // 16-row m tiles per wave
    constexpr int JS = 16 / NW;          // staging loads per operand, wave and k-step (2 LDS rows each)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD x (workgroups with blockIdx % 8 == x) takes the contiguous run of tiles [x*per, (x+1)*per); with a capped grid a
    // workgroup walks its XCD's run in strides of gridDim/8
    for (int slot = blockIdx.x >> 3; slot < p.per; slot += gridDim.x >> 3) {
    const int qt = (blockIdx.x & 7) * p.per + slot;
    if (qt >= p.tiles) break;
    __syncthreads();             // (the previous tile's LDS reads are done before this tile's first writes)
    int pr = 0;
    while (pr + 1 < p.nprob && qt >= p.tile0[pr + 1]) ++pr;
    const int qq = qt - p.tile0[pr];
    const int gx = p.gx[pr];
    const int m0 = (qq / gx) * 128, n0t = (qq % gx) * 128;
    const int K = p.K[pr], Ns = p.Ns, Nb = p.Nb;
    const char* Ab = p.A[pr];
    const char* Bb = p.B[pr];
    const int64_t lda = p.lda[pr], ldb = p.ldb[pr];
    const int arow0 = p.ta[pr] * Nb + p.n0[pr], brow0 = p.tb[pr] * Nb + p.n0[pr];

    // ---- staging roles: load j of this wave fills LDS rows 2*(JS*wave + j) + (lane >> 5), chunk slot lane & 31.  Each lane walks
    // its rows down the operands with running 64-bit pointers: per k-step a row index advances by 32 positions of the class,
    // i.e. by 32 + (time-step wraps) * (Nb - Ns) plane rows - one multiply-add per pointer and step.
    const int pc = lane & 31;
    const char* zsrc = p.zero + pc * 16;
    const char *pa[JS], *pb[JS];
    unsigned sa[JS], sb_[JS];     // row pitch in bytes, 0 for a column chunk outside the operand (pointer parked on the zero block)
    int nn[JS];
    const int gap = Nb - Ns;
#pragma unroll
    for (int j = 0; j < JS; ++j) {
        const int kr = 2 * (JS * wave + j) + (lane >> 5);
        const int c = pc ^ fsw(kr);                        // logical chunk: piece (16 columns), hi / lo, 8-column half
        const int piece = c >> 2, hl = (c >> 1) & 1, half = c & 1;
        const int ca = p.acol0[pr] + m0 + 16 * piece + 8 * half;
        const int cb = n0t + 16 * piece + 8 * half;
        const bool va = (m0 + 16 * piece + 8 * half < p.M[pr]) && (ca + 8 <= p.acols[pr]);
        const bool vb = (cb < p.N[pr]) && (cb + 8 <= p.bcols[pr]);
        const int t0 = kr / Ns;
        nn[j] = kr - t0 * Ns;
        const int64_t r0 = (int64_t)t0 * Nb + nn[j];
        pa[j] = va ? Ab + (r0 + arow0) * lda + ((ca >> 5) * 128 + hl * 64 + (ca & 31) * 2) : zsrc;
        pb[j] = vb ? Bb + (r0 + brow0) * ldb + ((cb >> 5) * 128 + hl * 64 + (cb & 31) * 2) : zsrc;
        sa[j] = va ? (unsigned)lda : 0u;
        sb_[j] = vb ? (unsigned)ldb : 0u;
    }
    // Register staging (global_load_dwordx4 -> ds_write_b128), one k-step ahead: the tile of step i+1 is written into the
    // other LDS slot right after the barrier of step i, and the loads of step i+2 are issued before the MFMAs of step i, so a
    // load has a whole k-step to land.  (LDS-DMA as in gemm_planes.hip was measured equal; behind a pending global_load_lds
    // the compiler also puts vmcnt(0) in front of the first transposing LDS read.)
    u32x4 ra_[JS], rb_[JS];
    auto load = [&](int k0) {
#pragma unroll
        for (int j = 0; j < JS; ++j) {
            const int kr = 2 * (JS * wave + j) + (lane >> 5);
            const bool rok = (k0 + kr) < K && !(p.flags & 64);
            ra_[j] = *reinterpret_cast<const u32x4*>(rok ? pa[j] : zsrc);
            rb_[j] = *reinterpret_cast<const u32x4*>(rok ? pb[j] : zsrc);
            nn[j] += 32;                                   // the next k-step of this lane's row
            unsigned adv = 32;
            while (nn[j] >= Ns) { nn[j] -= Ns; adv += gap; }
            pa[j] += (uint64_t)adv * sa[j];
            pb[j] += (uint64_t)adv * sb_[j];
        }
    };
    auto put = [&](int st) {
        char* base = smem + st * STAGE_B + wave * (JS * 1024) + lane * 16;
#pragma unroll
        for (int j = 0; j < JS; ++j) {
            *reinterpret_cast<u32x4*>(base + j * 1024) = ra_[j];
            *reinterpret_cast<u32x4*>(base + TILE_B + j * 1024) = rb_[j];
        }
    };

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- fragment addresses: group g = lane >> 4 reads LDS rows 8g .. 8g+7; lane 4q'+p' supplies row 8g + q', bytes 8p' ----
    const int i16 = lane & 15, g = lane >> 4, qp = i16 >> 2, pp = i16 & 3;
    const int fq = (2 * qp) ^ (8 * (g & 1));
    const int rbase = (8 * g + qp) * 512 + 8 * (pp & 1);
    int fa[MI][2], fb[4][2];
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i][hl] = rbase + ((((wm * MI + i) * 4 + 2 * hl) ^ fq) + (pp >> 1)) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j][hl] = TILE_B + rbase + ((((wn * 4 + j) * 4 + 2 * hl) ^ fq) + (pp >> 1)) * 16;
    }

    const int nk = (K + 31) / 32;
    if (nk > 0) { load(0); put(0); }
    if (nk > 1) load(32);
    for (int i = 0; i < nk; ++i) {
        __syncthreads();   // slot i&1 is complete; everyone is done reading slot (i+1)&1
        if (i + 1 < nk && !(p.flags & 32)) put((i + 1) & 1);
        if (i + 2 < nk && !(p.flags & 128)) load((i + 2) * 32);
        if (!(p.flags & 16)) {
            const char* sb = smem + (i & 1) * STAGE_B;
            bf16x8 ah[MI], al[MI];
#pragma unroll
            for (int ii = 0; ii < MI; ++ii) {
                ah[ii] = tr_frag(sb + fa[ii][0]);
                al[ii] = tr_frag(sb + fa[ii][1]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8 bh = tr_frag(sb + fb[j][0]);
                const bf16x8 bl = tr_frag(sb + fb[j][1]);
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
        for (int j = 0; j < 4; ++j) {
            const int n = n0t + (wn * 4 + j) * 16 + g * 4;
            if (n >= N) continue;
            const f32x4 v = acc[i][j] * alpha;
            float* cp = crow + n;
            if (n + 3 < N && vec) {
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
    int tiles = 0;
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
        p.gx[i] = cdiv(h_N[i], 128);
        p.tile0[i] = tiles;
        tiles += p.gx[i] * cdiv(h_M[i], 128);
    }
    p.tile0[count] = tiles;
    p.Ns = Ns; p.Nb = Nb;
    p.zero = (const char*)zero512;
    p.accumulate = accumulate;
    p.flags = aas_debug_flags_value();
    p.tiles = tiles;
    p.per = (tiles + 7) / 8;
    constexpr int LDS = 2 * 2 * 32 * 512;
    static bool attr_done = false;
    if (!attr_done) {
        AAS_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_planes_tn_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess &&
                      hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_planes_tn_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess,
                  "aas_gemm_planes_tn: could not raise the dynamic LDS limit");
        attr_done = true;
    }
    int grid = 8 * p.per;
    const int cap = aas_wgrad_wg_cap();     // > 0: at most this many workgroups (a multiple of 8), each walking several tiles
    if (cap > 0 && grid > cap) grid = cap < 8 ? 8 : cap / 8 * 8;
    if (p.flags & 65536) hipLaunchKernelGGL(gemm_planes_tn_kernel<4>, dim3(grid), dim3(256), LDS, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gemm_planes_tn_kernel<8>, dim3(grid), dim3(512), LDS, (hipStream_t)stream, p);
    AAS_LAUNCH_CHECK("aas_gemm_planes_tn");
    return 0;
}
