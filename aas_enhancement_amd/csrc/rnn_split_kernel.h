// Split-bf16 variant of the persistent recurrent kernels (see rnn_kernel.h for the structure).
//
// The exchanged vector (h_t forward, d(gates)_t in BPTT) is published by its PRODUCER already split
// into bf16 hi = rne(x) and bf16 lo = rne(x - hi), two values per 32-bit word, in two arrays with
// a row pitch of Hp = P*U units (pad units are written as zeros).  Consumers load MFMA A-fragments
// of 8 consecutive k straight from those arrays (16-byte sc1 buffer loads, no VALU in the loop) and
// issue hi*hi + lo*hi + hi*lo on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: fp32-class accuracy
// (~1e-5 relative, dropped term lo*lo) at 3/16 of the fp32-MFMA issue time.  The W_hh slice is
// split once before the time loop.  fp32 copies of h / d(gates) are still written (plain stores)
// for the layer's GEMMs.  Selected by aas_set_precision(1) when the caller passes an exchange buffer.
//
// Hand-off without a counter: the exchange arrays are filled with the poison word 0xFFFFFFFF (two bf16
// NaNs with all mantissa bits set - never produced by the gate math) before the launch; a consumer
// simply re-issues its sc1 loads until no word of the fragment is poison.  Every word is written by
// exactly one 4-byte write-through store and every address is written once per launch, so a non-poison
// word IS the data (the "data is the flag" granule form of MI355X_MICROARCH.md, with 4-byte granules):
// no drain, no arrival atomic, no poll of a separate flag - one store->load trip per step instead of
// three dependent round trips.  Spins are bounded (0.5 s) and set the sticky error word.
#pragma once
#include <type_traits>

#include "rnn_kernel.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_bf16(float x, unsigned& hi16, unsigned& lo16) {
    // round-to-nearest on both halves: |x - hi - lo| <= 2^-18 |x|
    const __bf16 hb = (__bf16)x;
    hi16 = (unsigned)__builtin_bit_cast(unsigned short, hb);
    const __bf16 l = (__bf16)(x - __uint_as_float(hi16 << 16));
    lo16 = (unsigned)__builtin_bit_cast(unsigned short, l);
}

// three bf16 terms, x = h + m + l EXACTLY (24 significant bits; the same split as gemm_planes.hip: split3)
__device__ __forceinline__ void split3_bf16(float x, unsigned& h16, unsigned& m16, unsigned& l16) {
    const __bf16 hb = (__bf16)x;
    h16 = (unsigned)__builtin_bit_cast(unsigned short, hb);
    const float r1 = x - __uint_as_float(h16 << 16);
    const __bf16 mb = (__bf16)r1;
    m16 = (unsigned)__builtin_bit_cast(unsigned short, mb);
    const float r2 = r1 - __uint_as_float(m16 << 16);
    const __bf16 lb = (__bf16)r2;
    l16 = (unsigned)__builtin_bit_cast(unsigned short, lb);
}

__device__ __forceinline__ void st_sc1_u32(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// sc0: written through the CU's L1 at once, kept in the XCD's L2 (no write-through to the fabric)
__device__ __forceinline__ void st_sc0_u32(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Managed exchange buffer (common.h: AasXchgPlan): every thread of the grid poisons its share of what the PREVIOUS launch on this
// buffer left un-poisoned in the other half (write-through 16-byte stores; that launch is complete, nobody reads the region now,
// and the launch after this one works there).  Replaces the poison memset launch in front of every persistent launch.
__device__ __forceinline__ void xchg_clean_other_half(const RnnP& p) {
    if (p.clean_words == 0) return;
    const unsigned nb = gridDim.x * gridDim.y * gridDim.z;
    const unsigned bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned n4 = p.clean_words >> 2;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.clean_ptr, 0, (int)(p.clean_words * 4u), 0x00020000);
    using B128 = decltype(__builtin_amdgcn_raw_buffer_load_b128(rs, 0, 0, 0));
    const u32x4 poison = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (unsigned i = bid * blockDim.x + threadIdx.x; i < n4; i += nb * blockDim.x)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(B128, poison), rs, (int)(i * 16u), 0, 16);
}

// ---- operand fragments of one 32-wide k chunk, in the two arithmetic modes of the persistent kernels -----------------------
// EX = false (aas_set_precision(1)): bf16 hi / lo halves, 8 consecutive k per lane, three v_mfma_f32_16x16x32_bf16.
// EX = true  (aas_set_precision(0)): the fp32 values themselves, 8 consecutive k per lane (k = 8q + j feeds the j-th of eight
//            v_mfma_f32_16x16x4_f32; A and B are permuted identically, so the sum runs over all 32 k): exact fp32 products with
//            fp32 accumulation - the reference's arithmetic.  Either way a fragment is two 16-byte words per lane.
template <bool EX>
__device__ __forceinline__ void frag_make(const float (&v)[8], u32x4& w0, u32x4& w1) {
    if constexpr (EX) {
        w0 = (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        w1 = (u32x4){__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])};
    } else {
        unsigned h[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split_bf16(v[e], h[e], l[e]);
        w0 = (u32x4){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
        w1 = (u32x4){l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
    }
}

// acc += X * Y over one 32-wide k chunk; X (x0, x1) is the MFMA's first operand (16 rows), Y (y0, y1) its second (16 columns)
template <bool EX>
__device__ __forceinline__ f32x4 mma_chunk(f32x4 acc, const u32x4& x0, const u32x4& x1, const u32x4& y0, const u32x4& y1) {
    if constexpr (EX) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(x0[j]), __uint_as_float(y0[j]), acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(x1[j]), __uint_as_float(y1[j]), acc, 0, 0, 0);
    } else {
        const bf16x8 xh = __builtin_bit_cast(bf16x8, x0), xl = __builtin_bit_cast(bf16x8, x1);
        const bf16x8 yh = __builtin_bit_cast(bf16x8, y0), yl = __builtin_bit_cast(bf16x8, y1);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, yh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, yl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, yh, acc, 0, 0, 0);
    }
    return acc;
}

// Exchange rows are KC chunks of 128 bytes per 32 elements in both modes: split = 64 B of bf16 hi | 64 B of bf16 lo, exact = 32 fp32.
// -> byte offsets inside a chunk of the two 16-byte words of lane quarter q, and the 32-bit word that carries element k (split: the
//    hi word of k's pair) - the word a consumer probes / a producer writes
template <bool EX> __device__ __forceinline__ constexpr unsigned frag_off0(int q) { return EX ? (unsigned)q * 32u : (unsigned)q * 16u; }
template <bool EX> __device__ __forceinline__ constexpr unsigned frag_off1() { return EX ? 16u : 64u; }
template <bool EX> __device__ __forceinline__ constexpr int elem_word(int k) { return (k / 32) * 32 + (EX ? (k % 32) : (k % 32) / 2); }

// ---- XCD co-location of an exchange set -----------------------------------------------------------------------------------
// The P workgroups of one (direction, row group) set are the only ones that exchange data.  With the XCD-aware grid (8 sets,
// workgroup b -> set b % 8, slice b / 8) they normally land on ONE XCD, and then the publish stores need not write through to
// the fabric: a plain store is acknowledged by that XCD's L2, which also serves the consumers' L1-bypassing (sc1) loads -
// LSTM BPTT 3.09 -> 2.65 us / step at N=30 and 4.74 -> 3.49 at N=60, bit-identical results.  Placement is not a contract, so
// every workgroup publishes its XCC id (write-through) in a table behind the exchange data, reads its set's P entries back and
// takes the plain-store path only if all agree; otherwise - or if the table read times out - the launch runs exactly as before.
constexpr int XCD_TAB_BYTES = 32 * 64 * 4;   // up to 32 sets x 64 slices

// -> bit c set: slice c of this workgroup's set runs on the same XCD as this workgroup (0: unknown - table read timed out)
// Placement statistics (diagnostics: tools/xcd_stats.py): per launch class four words behind the phase stamps of the sync buffer -
// workgroups that looked, workgroups whose whole set shares their XCD, co-located peers, peers.
constexpr int XSTAT_WORD = 1060;   // + 4 * class (0 LSTM forward, 1 LSTM BPTT, 2 GRU forward, 3 GRU BPTT)

__device__ __forceinline__ unsigned long long xcd_peer_mask(unsigned* tab, int set, int pslice, int P, unsigned* err, int tag,
                                                            unsigned long long* lds_word, unsigned* stat = nullptr) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xFu;
        unsigned* row = tab + set * 64;
        if (tid == 0) __hip_atomic_store(row + pslice, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned v = xcc, spins = 0;
        bool ok = true;
        while (true) {
            v = (tid < P) ? __hip_atomic_load(row + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : xcc;
            if (!__any(v == 0xFFFFFFFFu)) break;
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 255u) == 0 && (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                                          __builtin_amdgcn_s_memrealtime() - t0 > 50000000ull)) {
                if (tid == 0) __hip_atomic_store(err, (unsigned)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        const unsigned long long m = __ballot(tid < P && v == xcc);
        if (tid == 0) *lds_word = ok ? m : 0ull;
        if (tid == 0 && stat) {
            atomicAdd(stat + 0, 1u);
            atomicAdd(stat + 1, (ok && m == (P >= 64 ? ~0ull : ((1ull << P) - 1ull))) ? 1u : 0u);
            atomicAdd(stat + 2, ok ? (unsigned)__popcll(m) : 0u);
            atomicAdd(stat + 3, (unsigned)P);
        }
    }
    __syncthreads();
    const unsigned long long r = *lds_word;
    __syncthreads();
    return r;
}

__device__ __forceinline__ bool xcd_set_colocated(unsigned* tab, int set, int pslice, int P, unsigned* err, int tag, unsigned long long* lds_word,
                                                  unsigned* stat = nullptr) {
    const unsigned long long all = P >= 64 ? ~0ull : ((1ull << P) - 1ull);
    return xcd_peer_mask(tab, set, pslice, P, err, tag, lds_word, stat) == all;
}

// MODE, MT = 16-row batch tiles per workgroup, KS = 32-wide k chunks per wave
//
// NRB > 0 (exact forward modes, row groups of <= 4 NRB <= 8 rows): the recurrent product on v_mfma_f32_4x4x1_16B_f32 - sixteen 4 x 4 x 1
// blocks per instruction, D[lane 4b+j][reg i] = A[lane 4b+i] * B[lane 4b+j], and with the A broadcast (cbsz = 4, abid = a) every block
// takes block a's A lanes (tools/probe/mfma4x4.hip).  Here B = the W value of gate column `lane`, A = h_{t-1}: lane 4b+i of a 16-byte
// exchange load holds row i, k = 4b .. 4b+3, so ONE load per 64 k feeds 64 instructions (abid = b picks the k quad) with no LDS and no
// lane shuffles.  A 16 x 16 x 4 tile spends 32 cycles on 16 rows whether 8 or 16 exist; this form spends 8 cycles per k on each 4 rows
// that do, at the same peak rate: E's forward (N=30 over the chip: 8 rows per group) issues half the MFMA cycles.
//
// XF (R4 LSTM forward, one wave per SIMD: the lane's register budget is 512): the layer's INPUT PROJECTION inside the launch.  The wave
// also holds its k range of W_ih for gate column `lane` (KT more registers); x_{t+1} W_ih^T is formed right after h_t is published -
// MFMAs that need nothing from the exchange, issued while the published words travel - and kept as the next step's starting sum.
// The [T N, 2 G H] pre-activation tensor and the projection GEMM in front of the launch are gone (E's forward: 0.215 of 0.785 ms per layer).
template <int MODE, int MT, int KS, bool EX = false, int NRB = 0, bool XF = false>
__global__ __launch_bounds__(256, 1) void rnn_split_kernel(RnnP p) {
    using C = Cfg<MODE>;
    constexpr int G = C::G, U = C::U, NT = C::NT;
    constexpr bool FWD = (MODE == LSTM_FWD || MODE == GRU_FWD);
    constexpr bool LSTM = (MODE == LSTM_FWD || MODE == LSTM_BWD);
    constexpr int LDR = red_ld(NT, U);
    constexpr int ROWS = MT * 16;
    constexpr bool R4 = NRB > 0;
    static_assert(!R4 || (EX && (MODE == LSTM_FWD || MODE == GRU_FWD) && MT == 1 && KS % 2 == 0 && G * U <= 64 && NRB <= 4), "R4: exact forward, one 64-column group");
    static_assert(!XF || (R4 && MODE == LSTM_FWD), "XF: the 4 x 4 x 1 LSTM forward form only");
    constexpr int KT = KS * 32, KG = KS / 2;             // R4: k extent per wave, 64-wide k groups
    constexpr int EPT = (ROWS * U + 255) / 256;         // (row, unit) slots per thread; lanes l, l^1 hold a unit pair
    // double-buffered by step parity (one barrier per step) when it fits the 64 KB static LDS limit
    constexpr bool DB = (2 * 4 * ROWS * LDR * 4 <= 65536);
    __shared__ float red2[DB ? 2 : 1][4][ROWS][LDR];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware launch (p.xcd; forward modes, 8 m sets): workgroup b -> XCD class b % 8, which hosts the m sets
    // class + 8 j (j < m) of P workgroups each: set = (row group, direction), slice = (b / 8) % P
    const int xidx = (int)(blockIdx.x >> 3);
    const int xset = (int)(blockIdx.x & 7) + 8 * (xidx / p.P);
    const int pslice = p.xcd ? xidx % p.P : (int)blockIdx.x;
    const int qg = p.xcd ? (xset >> 1) : (int)blockIdx.y;
    const int d = p.xcd ? (xset & 1) : (int)blockIdx.z;
    const int T = p.T, N = p.N, H = p.H, GH = G * H;
    const int Hp = p.P * U;                             // padded unit pitch of the exchange arrays
    const int u0 = pslice * U;
    const int q0 = p.n0 + qg * p.rpg;
    const int NB = min(p.n1, q0 + p.rpg);
    const int Kxp = FWD ? Hp : G * Hp;                  // exchanged (padded) vector length per row
    const int kb = wave * KS * 32;
    unsigned* err = p.sync + ERR_WORD;

    // ---- B fragments (hi / lo) of this workgroup's W_hh slice ---------------------------------
    const float* W = d == 0 ? p.w_hh : p.w_hh_r;
    u32x4 b0[R4 ? 1 : KS][R4 ? 1 : NT], b1[R4 ? 1 : KS][R4 ? 1 : NT];
    float wr[R4 ? KT : 1];                               // R4: W[gate column `lane`][k of this wave]
    // XF: W_ih[gate column `lane`][this wave's k of the INPUT].  The waves split x's k unevenly: waves 0 and 1 also do the step's gate
    // math (8 rows x 16 units = 128 slots, ~0.45 us) while 2 and 3 wait at the next barrier, so they take 104 of every 256 k and
    // waves 2 and 3 take 152 - between two barriers every wave then has the same work (2.97 -> 2.75 us per step at N = 30, H = 500).
    constexpr int XK0 = !XF ? 1 : (KS == 4 ? 104 : KT), XK1 = !XF ? 1 : (KS == 4 ? 152 : KT);
    constexpr int XG = (XK1 + 63) / 64;                  // 64-k groups of the longer share
    const int xk = wave < 2 ? XK0 : XK1;
    const int xb = wave < 2 ? wave * XK0 : 2 * XK0 + (wave - 2) * XK1;
    float wx[XF ? XK1 : 1];
    if constexpr (R4) {
        // a lane's KT weights are contiguous in its row of W: 16-byte loads where the rows allow (a quarter of the load instructions,
        // each of them 64 different cache lines: the launch reaches its first step ~ 15 us sooner)
        const int gate = lane / U, unit = u0 + lane % U;
        const bool wrow_ok = lane < G * U && unit < H;
        auto load_slice = [&](const float* Wm, int K, float (&dst)[KT]) {
            const float* wrow = Wm + (int64_t)(gate * H + unit) * K;
            if ((K & 3) == 0) {
#pragma unroll
                for (int kk = 0; kk < KT; kk += 4) {
                    const int k = kb + kk;
                    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (wrow_ok && k < K) v = *reinterpret_cast<const f32x4*>(wrow + k);
                    dst[kk] = v[0]; dst[kk + 1] = v[1]; dst[kk + 2] = v[2]; dst[kk + 3] = v[3];
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < KT; ++kk) {
                    const int k = kb + kk;
                    dst[kk] = (wrow_ok && k < K) ? wrow[k] : 0.f;
                }
            }
        };
        load_slice(W, H, wr);
        if constexpr (XF) {
            const float* wrow = (d == 0 ? p.w_ih : p.w_ih_r) + (int64_t)(gate * H + unit) * p.I;      // (I % 4 == 0: split_xf_covers)
#pragma unroll
            for (int kk = 0; kk < XK1; kk += 4) {
                const int k = xb + kk;
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (wrow_ok && kk < xk && k < p.I) v = *reinterpret_cast<const f32x4*>(wrow + k);
                wx[kk] = v[0]; wx[kk + 1] = v[1]; wx[kk + 2] = v[2]; wx[kk + 3] = v[3];
            }
        }
    } else {
        const int n = lane & 15, q = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c = nt * 16 + n;
                // (two 16-byte loads instead of these eight - the lane's k are contiguous - reach the first step 10 us sooner and cost the
                //  32-unit LSTM forward 0.55 us on EVERY step, N = 60 on 128 CUs: 6.11 -> 6.65; the register assignment that follows
                //  from the loads decides the steady state.  Only the 4 x 4 x 1 forms load their slices with 16-byte loads.)
                float wv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = kb + ks * 32 + q * 8 + e;
                    float v = 0.f;
                    if (FWD) {
                        const int gate = c / U, unit = u0 + c % U;
                        if (c < G * U && unit < H && k < H) v = W[(int64_t)(gate * H + unit) * H + k];
                    } else {
                        const int unit = u0 + c, gate = k / Hp, ku = k - gate * Hp;
                        if (c < U && unit < H && gate < G && ku < H) v = W[(int64_t)(gate * H + ku) * H + unit];
                    }
                    wv[e] = v;
                }
                frag_make<EX>(wv, b0[ks][nt], b1[ks][nt]);
            }
    }

    // exchange arrays: hi then lo, each rows x (Kxp/2) 32-bit words
    // fwd: [2][T][N] rows - or, ring form of a managed buffer (exact forward kernels only), [2][4 time slots][N]: the row of time
    // index t is slot t & 3, and its producers poison their words again two steps after they published them (below);  bwd: [T][N][2]
    const bool ring = FWD && EX && p.ring;
    const int64_t xrows = ring ? (int64_t)2 * 4 * N : (int64_t)2 * T * N;
    auto xrow_f = [&](int dd, int tt) -> int64_t { return ring ? (int64_t)(dd * 4 + (tt & 3)) : (int64_t)dd * T + tt; };
    xchg_clean_other_half(p);
    // row = KC chunks of 128 bytes; chunk c holds elements [32c, 32c+32): 64 B of bf16 hi then 64 B of bf16 lo, so the
    // hi and lo fragments of a chunk share one 128-byte line (half as many distinct lines per step as two arrays)
    const int KC = (Kxp + 31) / 32;
    unsigned* xq = p.xchg;
    auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xq, 0, (int)(xrows * KC * 128), 0x00020000);

    float carry[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) carry[i] = 0.f;

    unsigned long long ph[5] = {0, 0, 0, 0, 0};
    unsigned long long retry_n = 0, retry_steps = 0;     // (stamp mode) reloads of the exchanged rows / steps whose first attempt found poison
    const bool stamp = (p.flags & 64) && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    __shared__ unsigned long long xcd_flag;
    bool plain = false;   // plain (L2-resident) publish stores once the set is verified to share an XCD (xcd_set_colocated)
    if (FWD && p.xcd) {
        unsigned* tab = p.xchg + xrows * KC * 32;          // behind the exchange rows, poisoned by the same memset
        plain = xcd_set_colocated(tab, xset, pslice, p.P, p.sync + ERR_WORD, p.tag, &xcd_flag, p.sync + XSTAT_WORD + 4 * (LSTM ? 0 : 2)) && !(p.flags & 524288);
    }
    // XF: lane 4b+i of a 16-byte load holds row q0 + 4 rb + i, k = xb + 64 g + 4b .. +3 of the layer input at one time index
    u32x4 xf[XF ? NRB : 1][XF ? XG : 1];
    f32x4 r4x[XF ? NRB : 1];
    [[maybe_unused]] auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(XF ? p.xin : nullptr), 0, XF ? (int)((int64_t)T * N * p.I * 4) : 0, 0x00020000);
    // per-lane offsets of those loads, formed once: the time index adds a wave-uniform term under the lane's validity mask (no
    // branches in the step: a divergent select around a load costs an s_waitcnt vmcnt(0) - i.e. the publish stores' round trip)
    unsigned xo[XF ? NRB : 1][XF ? XG : 1], xm[XF ? NRB : 1][XF ? XG : 1], ho[XF ? NRB : 1][XF ? KG : 1], hm[XF ? NRB : 1][XF ? KG : 1];
    if constexpr (XF) {
        const int i4 = lane & 3, bq = lane >> 2;
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const int gr = q0 + rb * 4 + i4;
#pragma unroll
            for (int g = 0; g < XG; ++g) {
                const int kk = g * 64 + bq * 4, k = xb + kk;
                const bool vx = gr < NB && kk < xk && k < p.I;
                xm[rb][g] = vx ? 0xFFFFFFFFu : 0u;
                xo[rb][g] = vx ? (unsigned)((gr * p.I + k) * 4) : 0x80000000u;
            }
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                const int k = kb + kg * 64 + bq * 4;
                const bool vh = gr < NB && k < Kxp;
                hm[rb][kg] = vh ? 0xFFFFFFFFu : 0u;
                ho[rb][kg] = vh ? (unsigned)(gr * KC * 128 + k * 4) : 0x80000000u;
            }
        }
    }
    u32x4 xn[XF ? NRB : 1][XF ? XG : 1];                 // the rows of the step after next, in flight (two-deep: see the step loop)
    [[maybe_unused]] auto x_load = [&](int tx, u32x4 (&dst)[XF ? NRB : 1][XF ? XG : 1]) __attribute__((always_inline)) {
        if constexpr (XF) {
            const unsigned trow = (unsigned)tx * (unsigned)(N * p.I * 4);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int g = 0; g < XG; ++g)
                    dst[rb][g] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(xo[rb][g] + (trow & xm[rb][g])), 0, 0));
        }
    };
    // the projection in two halves of the wave's k steps (4 k each: one abid value of a 64-k group): the step's h loads are issued
    // between them (below).  XK = the wave's share (XK0 / XK1), compile-time per role.
    f32x4 cx[XF ? NRB : 1][XF ? 4 : 1];
    [[maybe_unused]] auto x_mma_role = [&](auto XKC, auto PART) __attribute__((always_inline)) {
        if constexpr (XF) {
            constexpr int part = decltype(PART)::value, J = decltype(XKC)::value / 4;
            constexpr int JB = part == 0 ? 0 : J / 2, JE = part == 0 ? J / 2 : J;
            if constexpr (part == 0) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int v = 0; v < 4; ++v) cx[rb][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#define AAS_XF_STEP(G_, B_)                                                                                                          \
    if constexpr ((G_) < XG && (G_) * 16 + (B_) >= JB && (G_) * 16 + (B_) < JE) {                                                     \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) _Pragma("unroll") for (int rb = 0; rb < NRB; ++rb)                              \
            cx[rb][v] = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(xf[rb][(G_) < XG ? (G_) : 0][v]), wx[(((G_) < XG ? (G_) : 0) * 16 + (B_)) * 4 + v], cx[rb][v], 4, (B_), 0); \
    }
#define AAS_XF_GROUP(G_)                                                                                                              \
    AAS_XF_STEP(G_, 0) AAS_XF_STEP(G_, 1) AAS_XF_STEP(G_, 2) AAS_XF_STEP(G_, 3) AAS_XF_STEP(G_, 4) AAS_XF_STEP(G_, 5) AAS_XF_STEP(G_, 6) AAS_XF_STEP(G_, 7) \
    AAS_XF_STEP(G_, 8) AAS_XF_STEP(G_, 9) AAS_XF_STEP(G_, 10) AAS_XF_STEP(G_, 11) AAS_XF_STEP(G_, 12) AAS_XF_STEP(G_, 13) AAS_XF_STEP(G_, 14) AAS_XF_STEP(G_, 15)
            AAS_XF_GROUP(0) AAS_XF_GROUP(1) AAS_XF_GROUP(2)
#undef AAS_XF_GROUP
#undef AAS_XF_STEP
            if constexpr (part == 1) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) r4x[rb] = (cx[rb][0] + cx[rb][1]) + (cx[rb][2] + cx[rb][3]);
            }
        }
    };
    [[maybe_unused]] auto x_mma = [&](auto PART) __attribute__((always_inline)) {   // (not inlined: its state goes through scratch)
        if constexpr (XF) {
            if constexpr (XK0 == XK1) x_mma_role(std::integral_constant<int, XK0>{}, PART);
            else if (wave < 2) x_mma_role(std::integral_constant<int, XK0>{}, PART);
            else x_mma_role(std::integral_constant<int, XK1>{}, PART);
        }
    };
    // XF: the exchange loads of a step are issued at the END of the step before (between the projection halves, ~0.5 us behind this
    // workgroup's own publish, when the other producers' words of the same step are about to be visible): no pre-poll round trip
    u32x4 hfx[XF ? NRB : 1][XF ? KG : 1];
    [[maybe_unused]] auto h_issue = [&](int tprev) __attribute__((always_inline)) {
        if constexpr (XF) {
            const unsigned trow = (unsigned)(xrow_f(d, tprev) * N) * (unsigned)(KC * 128);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int kg = 0; kg < KG; ++kg)
                    hfx[rb][kg] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(ho[rb][kg] + (trow & hm[rb][kg])), 0, 16));
        }
    };
    if constexpr (XF) {
        x_load(d == 0 ? 0 : T - 1, xf);
        x_mma(std::integral_constant<int, 0>{});
        x_mma(std::integral_constant<int, 1>{});
        if (T > 1) x_load(d == 0 ? 1 : T - 2, xf);          // step 1's rows: used at the end of step 0
    }
    for (int s = 0; s < T; ++s) {
        unsigned long long st0 = stamp ? __builtin_amdgcn_s_memrealtime() : 0ull, st1 = st0, st2 = st0, st3 = st0;
        const int fwd_order = (d == 0) ? s : T - 1 - s;
        const int t = FWD ? fwd_order : (T - 1 - fwd_order);

        const int tp = FWD ? (d == 0 ? t - 1 : t + 1) : (d == 0 ? t + 1 : t - 1);

        // ---- prefetch the step's private inputs ------------------------------------------------
        float pin[EPT][4];
        float sav[EPT][6];
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / U, u = idx % U;
            const int gr = q0 + row, unit = u0 + u;
            const bool ok = (idx < ROWS * U) && gr < NB && unit < H;
#pragma unroll
            for (int g = 0; g < 4; ++g) pin[i][g] = 0.f;
#pragma unroll
            for (int g = 0; g < 6; ++g) sav[i][g] = 0.f;
            if (ok) {
                const int64_t tn = (int64_t)t * N + gr;
                if (FWD) {
                    if constexpr (!XF) {
                        const float* pp = p.pre + (tn * 2 + d) * GH + unit;
#pragma unroll
                        for (int g = 0; g < G; ++g) pin[i][g] = pp[g * H];
                    }
                } else {
                    pin[i][0] = p.dy[tn * H + unit];
                    const f32x4 ga4 = *reinterpret_cast<const f32x4*>(p.gact + (((int64_t)d * T * N + tn) * H + unit) * 4);
                    sav[i][0] = ga4.x; sav[i][1] = ga4.y; sav[i][2] = ga4.z; sav[i][3] = ga4.w;
                    const int tq = (d == 0) ? t - 1 : t + 1;
                    const bool hasq = (tq >= 0 && tq < T);
                    const int64_t qn = ((int64_t)d * T * N + (int64_t)tq * N + gr) * H + unit;
                    if (LSTM) {
                        sav[i][4] = p.cst[((int64_t)d * T * N + tn) * H + unit];
                        sav[i][5] = hasq ? p.cst[qn] : 0.f;
                    } else {
                        sav[i][5] = hasq ? p.hout[qn] : 0.f;
                    }
                }
            }
        }

        // ---- recurrent product ------------------------------------------------------------------
        f32x4 acc[MT][NT];
        f32x4 r4[R4 ? NRB : 1];                          // R4: lane = gate column, register i = row 4 rb + i
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            const int m = lane & 15, q = lane >> 4;
            // byte offset (within the hi or lo array) of this lane's first 8 elements
            constexpr unsigned OOB = 0x80000000u;
            constexpr unsigned POISON = 0xFFFFFFFFu;
            unsigned roff[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int gr = q0 + mt * 16 + m;
                const int64_t xr = FWD ? xrow_f(d, tp) * N + gr : ((int64_t)tp * N + gr) * 2 + d;
                roff[mt] = (gr < NB && !(p.flags & 1)) ? (unsigned)((xr * KC + wave * KS) * 128) + frag_off0<EX>(q) : OOB;
            }
            const int klane = kb + q * 8;
            constexpr int CH = KS >= 2 ? 2 : 1;
            constexpr int NCH = KS / CH;
            constexpr int DEPTH = NCH >= 3 ? 2 : (NCH >= 2 ? 1 : 0);
            u32x4 ahb[DEPTH + 1][CH][MT], alb[DEPTH + 1][CH][MT];
            // Every exchange load is an sc1 load (MI355X_MICROARCH.md "Valid forms": all loads of the handed-off bytes sc1).
            // A cacheable (plain) first attempt - so that the workgroups of one XCD share the rows through its L2 - measured no
            // faster, and its safety rests on "a stale L2 line can only show POISON", which does not cover lines that
            // survive from the previous launch on this buffer (the previous layer's h at the same positions).  Debug flag 32
            // (also a GEMM ablation bit) re-enables the plain attempt for experiments.
            auto issue_aux = [&](int c, u32x4 (&dh)[CH][MT], u32x4 (&dl)[CH][MT], auto AUX) {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int ks = c * CH + j;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const unsigned off = (klane + ks * 32 < Kxp) ? roff[mt] + (unsigned)(ks * 128) : OOB;
                        dh[j][mt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, decltype(AUX)::value));
                        dl[j][mt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(off + frag_off1<EX>()), 0, decltype(AUX)::value));
                    }
                }
            };
            auto issue = [&](int c, u32x4 (&dh)[CH][MT], u32x4 (&dl)[CH][MT]) {
                if (p.flags & 32) issue_aux(c, dh, dl, std::integral_constant<int, 0>{});   // experiment only
                else issue_aux(c, dh, dl, std::integral_constant<int, 16>{});
            };
            auto issue_fresh = [&](int c, u32x4 (&dh)[CH][MT], u32x4 (&dl)[CH][MT]) {
                issue_aux(c, dh, dl, std::integral_constant<int, 16>{});
            };
            // a fragment is complete when none of its words is the poison word
            auto poisoned = [&](const u32x4 (&dh)[CH][MT], const u32x4 (&dl)[CH][MT]) -> bool {
                unsigned mx = 0u;
#pragma unroll
                for (int j = 0; j < CH; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const u32x4 a = dh[j][mt], b = dl[j][mt];
                        mx = max(mx, max(max(a.x, a.y), max(a.z, a.w)));
                        mx = max(mx, max(max(b.x, b.y), max(b.z, b.w)));
                    }
                return __any(mx == POISON) != 0;
            };
            // cheap pre-poll: before streaming the fragments, each wave watches ONE word per producer slice of its
            // k-range (first row of the group, hi array) with a single 4-byte sc1 load per lane; the full loads
            // below are still validated word by word, this only keeps 250 workgroups from hammering the fabric
            // with 16-byte re-loads while the step's data is in flight.
            if (!XF && !(p.flags & 4)) {
                const int nprod = (KS * 32) / U;               // producer slices inside this wave's k-range
                const int64_t xr0 = FWD ? xrow_f(d, tp) * N + q0 : ((int64_t)tp * N + q0) * 2 + d;
                const int kprobe = kb + lane * U;
                const bool probe = lane < nprod && kprobe < Kxp && !(p.flags & 1);
                const unsigned* wp = xq + xr0 * KC * 32 + elem_word<EX>(kprobe);
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                unsigned spins = 0;
                while (true) {
                    const unsigned w = probe ? __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    if (!__any(w == POISON)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 0) {
                        if (ld_cnt(err) != 0) break;
                        if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) {
                            if (lane == 0) __hip_atomic_store(err, (unsigned)p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
            }
            if (stamp) st1 = __builtin_amdgcn_s_memrealtime();
            if constexpr (R4) {
                // lane 4b+i: row q0 + 4 rb + i, k = kb + 64 kg + 4b .. +3 (16 bytes of the fp32 exchange row)
                const int i4 = lane & 3, bq = lane >> 2;
                u32x4 hf[NRB][KG];
                unsigned hoff[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const int gr = q0 + rb * 4 + i4;
                    const int64_t xr = xrow_f(d, tp) * N + gr;
                    hoff[rb] = (gr < NB && !(p.flags & 1)) ? (unsigned)(xr * KC * 128 + (kb + bq * 4) * 4) : OOB;
                }
                auto load_all = [&]() {
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                        for (int kg = 0; kg < KG; ++kg) {
                            const unsigned off = (kb + kg * 64 + bq * 4 < Kxp) ? hoff[rb] + (unsigned)(kg * 256) : OOB;
                            hf[rb][kg] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 16));
                        }
                };
                auto any_poison = [&]() -> bool {
                    unsigned mx = 0u;
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                        for (int kg = 0; kg < KG; ++kg) mx = max(mx, max(max(hf[rb][kg].x, hf[rb][kg].y), max(hf[rb][kg].z, hf[rb][kg].w)));
                    return __any(mx == POISON) != 0;
                };
                if constexpr (XF) {
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                        for (int kg = 0; kg < KG; ++kg) hf[rb][kg] = hfx[rb][kg];
                } else {
                    load_all();
                }
                if (!(p.flags & 4) && any_poison()) {
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    unsigned spins = 0;
                    do {
                        load_all();
                        if ((++spins & 63u) == 0) {
                            if (ld_cnt(err) != 0) break;
                            if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) {
                                if (lane == 0) __hip_atomic_store(err, (unsigned)p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                        }
                    } while (any_poison());
                    if (stamp) { retry_n += spins; retry_steps += 1; }
                }
                // NRB x 4 independent accumulation chains (row block, k mod 4)
                f32x4 cc[NRB][4];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int v = 0; v < 4; ++v) cc[rb][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!(p.flags & 2)) {
#pragma unroll
                    for (int kg = 0; kg < KG; ++kg) {
#define AAS_R4_STEP(B_)                                                                                                              \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) _Pragma("unroll") for (int rb = 0; rb < NRB; ++rb)                                  \
        cc[rb][v] = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(hf[rb][kg][v]), wr[kg * 64 + (B_) * 4 + v], cc[rb][v], 4, (B_), 0);
                        AAS_R4_STEP(0) AAS_R4_STEP(1) AAS_R4_STEP(2) AAS_R4_STEP(3) AAS_R4_STEP(4) AAS_R4_STEP(5) AAS_R4_STEP(6) AAS_R4_STEP(7)
                        AAS_R4_STEP(8) AAS_R4_STEP(9) AAS_R4_STEP(10) AAS_R4_STEP(11) AAS_R4_STEP(12) AAS_R4_STEP(13) AAS_R4_STEP(14) AAS_R4_STEP(15)
#undef AAS_R4_STEP
                    }
                }
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) r4[rb] = (cc[rb][0] + cc[rb][1]) + (cc[rb][2] + cc[rb][3]);
            } else {
#pragma unroll
            for (int c = 0; c < DEPTH && c < NCH; ++c) issue(c, ahb[c % (DEPTH + 1)], alb[c % (DEPTH + 1)]);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c + DEPTH < NCH) issue(c + DEPTH, ahb[(c + DEPTH) % (DEPTH + 1)], alb[(c + DEPTH) % (DEPTH + 1)]);
                if (!(p.flags & 4)) {
                    if (poisoned(ahb[c % (DEPTH + 1)], alb[c % (DEPTH + 1)])) {
                        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                        unsigned spins = 0;
                        do {
                            issue_fresh(c, ahb[c % (DEPTH + 1)], alb[c % (DEPTH + 1)]);
                            if ((++spins & 63u) == 0) {
                                if (ld_cnt(err) != 0) break;
                                if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) {
                                    if (lane == 0) __hip_atomic_store(err, (unsigned)p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    break;
                                }
                            }
                        } while (poisoned(ahb[c % (DEPTH + 1)], alb[c % (DEPTH + 1)]));
                    }
                }
                if (!(p.flags & 2)) {
#pragma unroll
                    for (int j = 0; j < CH; ++j)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = mma_chunk<EX>(acc[mt][nt], ahb[c % (DEPTH + 1)][j][mt], alb[c % (DEPTH + 1)][j][mt], b0[c * CH + j][nt], b1[c * CH + j][nt]);
                        }
                }
            }
            }
        }
        if (stamp) st2 = __builtin_amdgcn_s_memrealtime();
        if constexpr (XF) {
            // the rows the projection at the end of this step works on: loaded a step ago (below), every load has returned by now
            if (s > 0) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int g = 0; g < XG; ++g) xf[rb][g] = xn[rb][g];
            }
        }
        float (*red)[ROWS][LDR] = red2[DB ? (s & 1) : 0];
        // ---- cross-wave reduction through LDS ---------------------------------------------------
        if constexpr (R4) {
            if (lane < NT * 16) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[wave][rb * 4 + r][lane] = ((s > 0) ? r4[rb][r] : 0.f) + (XF ? r4x[XF ? rb : 0][r] : 0.f);
            }
        } else {
            const int col = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[wave][mt * 16 + rq + r][nt * 16 + col] = acc[mt][nt][r];
        }
        __syncthreads();

        if (stamp) st3 = __builtin_amdgcn_s_memrealtime();
        // ---- gate math: one (row, unit) per thread slot; lanes l and l^1 (adjacent units) pair up to publish
        //      one hi word and one lo word ---------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / U, u = idx % U;
            const int gr = q0 + row, unit = u0 + u;
            const bool rowok = (idx < ROWS * U) && gr < NB;
            const bool ok = rowok && unit < H;
            const int64_t tn = (int64_t)t * N + gr;
            float xv[4] = {0.f, 0.f, 0.f, 0.f};  // published values: fwd [0] = h; bwd [g] = exchanged gate gradients
            float kp[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // values only this layer's later GEMMs / BPTT need (stored after the publish)
            if (ok) {
                float rs[G];
                if (FWD) {
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        rs[g] = red[0][row][g * U + u] + red[1][row][g * U + u] + red[2][row][g * U + u] + red[3][row][g * U + u];
                } else {
                    rs[0] = red[0][row][u] + red[1][row][u] + red[2][row][u] + red[3][row][u];
                }
                if (MODE == LSTM_FWD) {
                    const float ig = sigmoidf_(pin[i][0] + rs[0]);
                    const float fg = sigmoidf_(pin[i][1] + rs[1]);
                    const float gg = tanhf_(pin[i][2] + rs[2]);
                    const float og = sigmoidf_(pin[i][3] + rs[3]);
                    const float c = fg * carry[i] + ig * gg;
                    carry[i] = c;
                    const float h = og * tanhf_(c);
                    xv[0] = h;
                    kp[0] = ig; kp[1] = fg; kp[2] = gg; kp[3] = og; kp[4] = c;
                    if (t >= (gr < p.cls_n ? p.cls_t0 : p.cls_t1)) {   // dead (t, row) of a shorter row class
                        carry[i] = 0.f; xv[0] = 0.f;
                        kp[0] = kp[1] = kp[2] = kp[3] = kp[4] = 0.f;
                    }
                } else if (MODE == GRU_FWD) {
                    const float rg = sigmoidf_(pin[i][0] + rs[0]);
                    const float zg = sigmoidf_(pin[i][1] + rs[1]);
                    const float hn = rs[2];
                    const float ng = tanhf_(pin[i][2] + rg * hn);
                    const float h = (1.f - zg) * ng + zg * carry[i];
                    carry[i] = h;
                    xv[0] = h;
                    kp[0] = rg; kp[1] = zg; kp[2] = ng; kp[3] = hn;
                    if (t >= (gr < p.cls_n ? p.cls_t0 : p.cls_t1)) {   // z = 0, n = 1: BPTT forms no gate gradient and carries nothing on
                        carry[i] = 0.f; xv[0] = 0.f;
                        kp[0] = 0.f; kp[1] = 0.f; kp[2] = 1.f; kp[3] = 0.f;
                    }
                } else if (MODE == LSTM_BWD) {
                    const float dh = pin[i][0] + rs[0];
                    const float ig = sav[i][0], fg = sav[i][1], gg = sav[i][2], og = sav[i][3];
                    const float c = sav[i][4], cp = sav[i][5];
                    const float tc = tanhf_(c);
                    const float dc = dh * og * (1.f - tc * tc) + carry[i];
                    carry[i] = dc * fg;
                    xv[0] = dc * gg * ig * (1.f - ig);
                    xv[1] = dc * cp * fg * (1.f - fg);
                    xv[2] = dc * ig * (1.f - gg * gg);
                    xv[3] = dh * tc * og * (1.f - og);
                } else {  // GRU_BWD
                    const float dh = pin[i][0] + rs[0] + carry[i];
                    const float rg = sav[i][0], zg = sav[i][1], ng = sav[i][2], hn = sav[i][3];
                    const float hp = sav[i][5];
                    carry[i] = dh * zg;
                    const float dnp = dh * (1.f - zg) * (1.f - ng * ng);
                    const float dzp = dh * (hp - ng) * zg * (1.f - zg);
                    const float drp = dnp * hn * rg * (1.f - rg);
                    xv[0] = drp; xv[1] = dzp; xv[2] = dnp * rg;
                    kp[0] = dnp;
                }
            }
            // publish: even-unit lanes store {own, partner} packed hi and lo words (pad units publish zeros)
            constexpr int GX = FWD ? 1 : G;
            const int64_t xr = FWD ? xrow_f(d, t) * N + gr : ((int64_t)t * N + gr) * 2 + d;
            const int64_t rbase_w = xr * KC * 32;  // row start in 32-bit words
#pragma unroll
            for (int g = 0; g < GX; ++g) {
                const int k = g * Hp + unit;  // element index within the exchanged row (split: of the even unit of the pair)
                if constexpr (EX) {           // every (row, unit) lane publishes its own fp32 word (pad units: zeros)
                    if (rowok && s + 1 < T && !(p.flags & 8)) {
                        unsigned* wq = xq + rbase_w + k;
                        if (plain) st_sc0_u32(wq, __float_as_uint(xv[g]));
                        else st_sc1_u32(wq, __float_as_uint(xv[g]));
                    }
                    if constexpr (FWD) {
                        // ring: the word this lane published two steps ago is poison again.  Every workgroup of the set has read
                        // that row: this workgroup is past the reduction barrier of step s, so all its waves have seen every
                        // producer's h of step s-1, and a producer publishes that only after its own reads of step s-2's row.
                        // The slot is written next at step s+2 - a whole step (with its vmcnt(0) waits) after this store.
                        if (ring && rowok && s >= 2) {
                            unsigned* cq = xq + (xrow_f(d, d == 0 ? t - 2 : t + 2) * N + gr) * KC * 32 + k;
                            if (plain) st_sc0_u32(cq, 0xFFFFFFFFu);
                            else st_sc1_u32(cq, 0xFFFFFFFFu);
                        }
                    }
                } else {
                    unsigned h0, l0;
                    split_bf16(xv[g], h0, l0);
                    const unsigned mine = h0 | (l0 << 16);
                    const unsigned other = __shfl_xor(mine, 1, 64);
                    if (rowok && !(u & 1) && s + 1 < T && !(p.flags & 8)) {  // the last step's output is not exchanged
                        unsigned* wq = xq + rbase_w + (k / 32) * 32 + (k % 32) / 2;
                        if (plain) {
                            st_sc0_u32(wq, (mine & 0xFFFFu) | (other << 16));
                            st_sc0_u32(wq + 16, (mine >> 16) | (other & 0xFFFF0000u));
                        } else {
                            st_sc1_u32(wq, (mine & 0xFFFFu) | (other << 16));
                            st_sc1_u32(wq + 16, (mine >> 16) | (other & 0xFFFF0000u));
                        }
                    }
                }
            }
            // fp32 results for the rest of the layer (plain stores, off the exchange critical path)
            if (ok) {
                if (FWD) {
                    p.hout[((int64_t)d * T * N + tn) * H + unit] = xv[0];
                    *reinterpret_cast<f32x4*>(p.gact + (((int64_t)d * T * N + tn) * H + unit) * 4) = (f32x4){kp[0], kp[1], kp[2], kp[3]};
                    if (LSTM) p.cst[((int64_t)d * T * N + tn) * H + unit] = kp[4];
                } else {
                    float* dg = p.dg1 + (tn * 2 + d) * GH + unit;
#pragma unroll
                    for (int g = 0; g < G; ++g) dg[g * H] = xv[g];
                    if (!LSTM) {
                        float* dx_ = p.dg2 + (tn * 2 + d) * GH + unit;
                        dx_[0] = xv[0]; dx_[H] = xv[1]; dx_[2 * H] = kp[0];
                    }
                }
            }
        }
        if (!DB) __syncthreads();  // single reduction buffer: readers must finish before the next step's writes
        if constexpr (XF) {
            // behind the publish stores, in front of the next step's poll: the projection of the next step's input
            __builtin_amdgcn_sched_barrier(0);
            // Input rows of the step AFTER NEXT, two-deep: where a load is issued decides who waits for it.  Loads return in order and the
            // compiler places s_waitcnt vmcnt(0) in front of the barrier and of the first register it cannot prove free (the gate
            // math's), so the only stretch of the step that no wait cuts is this one: behind the publish, in front of the exchange
            // loads - whose wait, a projection later, then covers these too.
            if (s + 2 < T) x_load(d == 0 ? t + 2 : t - 2, xn);
            if (s + 1 < T) {
                __builtin_amdgcn_sched_barrier(0);
                x_mma(std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
                // the row this step published = the next step's h_{t-1}.  (Issued right behind the publish instead: 3 of 200 first
                // attempts find poison, same step time; behind the whole projection: + 0.3 us per step.)
                h_issue(t);
                __builtin_amdgcn_sched_barrier(0);
                x_mma(std::integral_constant<int, 1>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (stamp) {
            const unsigned long long st4 = __builtin_amdgcn_s_memrealtime();
            ph[0] += st1 - st0; ph[1] += st2 - st1; ph[2] += st3 - st2; ph[3] += st4 - st3; ph[4] += st4 - st0;
        }
    }
    if (stamp && tid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.sync + STAMP_WORD);
        for (int i = 0; i < 5; ++i) o[i] = ph[i];
        o[5] = retry_n; o[6] = retry_steps;
    }
}

template <int MODE, int MT, int KS, bool EX>
int launch_sk(const RnnP& p, hipStream_t s) {
    dim3 grid(p.P, p.Q, 2);
    if (p.xcd) grid = dim3(p.P * p.Q * 2);
    if constexpr (EX && (MODE == LSTM_FWD || MODE == GRU_FWD) && MT == 1 && KS % 2 == 0 && KS <= 4) {
        // <= 8 rows per group: the 4 x 4 x 1 block form (debug bit 268435456: the 16 x 16 x 4 tiles).  (Tried for the 1000-unit GRU
        // over the whole chip too - KS = 8, four row blocks, 63 slices x 2 groups x 2 directions instead of the 32-unit kernel's
        // 8-row tiles: 8.26 vs 8.23 us / step, its sixteen exchange loads are not overlapped with the MFMAs - not kept.)
        const int rows = (p.n1 - p.n0) < p.rpg ? (p.n1 - p.n0) : p.rpg;
        if (rows <= 8 && !(p.flags & 268435456)) {
            if constexpr (MODE == LSTM_FWD) {
                if (p.xin) {     // input projection inside the launch (split_xf_covers() admitted the shape)
                    if (rows <= 4) hipLaunchKernelGGL((rnn_split_kernel<MODE, MT, KS, EX, 1, true>), grid, dim3(256), 0, s, p);
                    else hipLaunchKernelGGL((rnn_split_kernel<MODE, MT, KS, EX, 2, true>), grid, dim3(256), 0, s, p);
                    return 0;
                }
            }
            if (rows <= 4) hipLaunchKernelGGL((rnn_split_kernel<MODE, MT, KS, EX, 1>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((rnn_split_kernel<MODE, MT, KS, EX, 2>), grid, dim3(256), 0, s, p);
            return 0;
        }
    }
    if (p.xin) return -1;   // (no `pre` exists for any other kernel of the family)
    hipLaunchKernelGGL((rnn_split_kernel<MODE, MT, KS, EX>), grid, dim3(256), 0, s, p);
    return 0;
}

template <int MODE, int MT, bool EX>
int launch_split_mt(const RnnP& p, int ks_need, hipStream_t s) {
    constexpr bool FWD = (MODE == LSTM_FWD || MODE == GRU_FWD);
    if (ks_need <= 1) return launch_sk<MODE, MT, 1, EX>(p, s);
    if (ks_need <= 2) return launch_sk<MODE, MT, 2, EX>(p, s);
    if (ks_need <= 4) return launch_sk<MODE, MT, 4, EX>(p, s);
    if constexpr (MODE == LSTM_FWD) return -1;  // H > 512: fall back to the fp32 kernel
    else {
        if (ks_need <= 8) return launch_sk<MODE, MT, 8, EX>(p, s);
        if constexpr (FWD || EX) return -1;
        else {
            if (ks_need <= 16) return launch_sk<MODE, MT, 16, EX>(p, s);
            if constexpr (MT == 1) {     // (two row tiles x 24 k-steps spilled registers: those shapes take the counter-based kernel)
                if (ks_need <= 24) return launch_sk<MODE, MT, 24, EX>(p, s);
            }
            return -1;
        }
    }
}

// Which (MODE, EX, row tiles, k slices) launch_split_mt has a kernel for - asked BEFORE anything is planned or poisoned for the launch.
template <int MODE, bool EX>
constexpr bool split_covers(int mt, int ks_need) {
    constexpr bool FWD = (MODE == LSTM_FWD || MODE == GRU_FWD);
    if (ks_need <= 4) return true;
    if (MODE == LSTM_FWD) return false;
    if (ks_need <= 8) return true;
    if (FWD || EX) return false;
    if (ks_need <= 16) return true;
    return mt == 1 && ks_need <= 24;
}

// Does the LSTM forward launch with the input projection inside (XF) cover this shape, in the present settings?  It is the exact
// 4 x 4 x 1 form or nothing: <= 8 rows per workgroup on the CU budget, the whole batch in one launch, H and I <= 512 (the two weight
// slices of a wave are 2 x 128 registers), 16-byte rows of x.  Debug bit 1073741824 switches it off (A/B runs).
inline bool split_xf_covers(int T, int N, int H, int I, const void* xchg) {
    if (!xchg || aas_precision_value() != 0 || T < 1 || N < 1 || H < 1) return false;
    const int flags = aas_debug_flags_value();
    if (flags & (134217728 | 268435456 | 1073741824)) return false;
    const int cus = aas_rnn_cus();
    if (cus <= 0) return false;
    if (!(H < 256 || (flags & 512) || cdiv(H, 16) * cdiv(N, 8) * 2 <= cus)) return false;    // run_fwd32 takes the launch
    const int P = cdiv(H, 16);
    if (P * 2 > cus || P * 16 > 512 || I > 512 || I < 4 || (I & 3)) return false;
    int mt, rpg;
    pick_groups(P, N, cus, mt, rpg);
    if (mt != 1 || rpg > 8) return false;
    const int qmax = cus / (P * 2) < 1 ? 1 : cus / (P * 2);
    if (N > qmax * rpg) return false;
    if ((int64_t)T * N * I * 4 >= 0x7fffffffLL) return false;
    if ((int64_t)2 * T * N * ((P * 16 + 31) / 32) * 128 >= 0x7fffffffLL) return false;
    return true;
}

// Same chunking of the batch as run() in rnn_kernel.h; falls back to the counter-based fp32 kernel (returns -1)
// when the shape is outside the instantiated kernels.  EX: exact-fp32 products (forward modes only).
template <int MODE, bool EX = false>
int run_split(const char* name, RnnP p, hipStream_t s) {
    using C = Cfg<MODE>;
    constexpr bool FWD = (MODE == LSTM_FWD || MODE == GRU_FWD);
    if constexpr (EX && !FWD) return -1;
    else {
    AAS_CHECK(p.T >= 1 && p.N >= 1 && p.H >= 1, "%s: bad sizes T=%d N=%d H=%d", name, p.T, p.N, p.H);
    const int cus = aas_rnn_cus();
    AAS_CHECK(cus > 0, "%s: no HIP device", name);
    p.flags = aas_debug_flags_value();
    p.tag = aas_rnn_launch_tag_value();
    p.P = cdiv(p.H, C::U);
    AAS_CHECK(p.P * 2 <= cus, "%s: H=%d needs %d resident workgroups, device has %d CUs", name, p.H, p.P * 2, cus);
    const int Hp = p.P * C::U;
    const int kxp = FWD ? Hp : C::G * Hp;
    int ks_need = cdiv(kxp, 128);
    if (p.xin) {     // fused input projection: the waves split max(H, I) the same way, on the 4 x 4 x 1 form (k slices of 64 or 128 per wave)
        const int kin = cdiv(p.I > kxp ? p.I : kxp, 128);
        ks_need = kin < 2 ? 2 : kin;
    }
    const int64_t xbytes = (int64_t)2 * p.T * p.N * ((kxp + 31) / 32) * 128;  // rows x chunks x 128 B (64 B hi + 64 B lo, or 32 fp32)
    if (xbytes >= 0x7fffffffLL) return -1;
    int mt, rpg;
    pick_groups(p.P, p.N, cus, mt, rpg);
    // no kernel of this family for the shape: say so before the exchange buffer's bookkeeping is committed to a launch (aas_xchg_plan)
    // or the buffer is poisoned for one - the caller runs another kernel on the whole buffer
    if (!split_covers<MODE, EX>(mt, ks_need)) return -1;
    p.rpg = rpg;
    const int qmax = cus / (p.P * 2) < 1 ? 1 : cus / (p.P * 2);
    if (FWD) aas_note_fwd_h_planes((!EX && p.N <= qmax * rpg) ? ((kxp + 31) / 32) * 128 : 0);   // (chunked launches re-poison the buffer)
    for (int n0 = 0; n0 < p.N; n0 += qmax * rpg) {
        p.n0 = n0;
        const int rows = (p.N - n0) < qmax * rpg ? (p.N - n0) : qmax * rpg;
        p.n1 = n0 + rows;
        p.Q = cdiv(rows, rpg);
        // poison the exchange arrays: a word is valid data once it is no longer 0xFFFFFFFF (+ the XCC table) - or, on a managed buffer
        // (exact forward kernels, one launch for the batch), the ring of four time slots in this launch's half: no memset launch
        AasXchgPlan plan = {};
        unsigned* const xchg0 = p.xchg;
        if (FWD && EX && p.N <= qmax * rpg && p.T >= 4) aas_xchg_plan(p.xchg, (size_t)2 * 4 * p.N * ((kxp + 31) / 32) * 128 + XCD_TAB_BYTES, s, &plan);
        p.clean_ptr = nullptr; p.clean_words = 0; p.ring = 0;
        if (plan.managed) {
            p.xchg = plan.base; p.clean_ptr = plan.clean_ptr; p.clean_words = plan.clean_words; p.ring = 1;
        } else {
            if (aas_xchg_legacy_fill(p.xchg, (size_t)xbytes + XCD_TAB_BYTES, s)) return 2;
        }
        p.xcd = (FWD && (p.Q * 2) % 8 == 0 && p.P * (p.Q * 2 / 8) <= 32 && rpg <= 8 && !(p.flags & 262144)) ? 1 : 0;   // (see rnn_fwd32_kernel.h)
        int rc = (mt == 1) ? launch_split_mt<MODE, 1, EX>(p, ks_need, s) : launch_split_mt<MODE, 2, EX>(p, ks_need, s);
        p.xchg = xchg0;
        if (rc != 0) {
            // (not reached: split_covers() refused the shape above.  Should a kernel table and split_covers ever disagree, the
            //  buffer's management is dropped so that the next launch poisons it afresh instead of trusting stale bookkeeping.)
            if (plan.managed) aas_rnn_xchg_forget(xchg0);
            return -1;
        }
        AAS_LAUNCH_CHECK(name);
    }
    return 0;
    }
}

// The data-is-the-flag kernels in the arithmetic the library is set to (aas_set_precision): split-bf16, or exact fp32 (forward
// modes; shapes they do not cover, and the all-gather BPTT, run on the counter-based fp32 kernel of rnn_kernel.h).
// Debug bit 134217728: always the counter-based kernel in exact mode (the previous round's fp32 path, for A/B runs).
template <int MODE>
int run_any(const char* name, RnnP p, hipStream_t s) {
    if (p.xchg && aas_precision_value() == 1) {
        const int rc = run_split<MODE, false>(name, p, s);
        if (rc >= 0) return rc;
    } else if (p.xchg && !(aas_debug_flags_value() & 134217728)) {
        const int rc = run_split<MODE, true>(name, p, s);
        if (rc >= 0) return rc;
    }
    aas_note_fwd_h_planes(0);
    return run<MODE>(name, p, s);
}

}  // namespace
