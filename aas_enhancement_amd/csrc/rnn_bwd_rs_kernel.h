// BPTT of the persistent recurrent kernels as a per-step REDUCE-SCATTER (split-bf16 products).
//
// rnn_split_kernel.h runs BPTT as an all-gather: every workgroup reads the full d(gates)_{t+1} row (G*H values) of
// each of its batch rows and multiplies it with its column slice of W_hh - P-fold redundant reads of the widest
// vector of the layer (2000 values per row for the 500-unit LSTM, 3000 for the 1000-unit GRU) bound the step.
// Here the product is split along K instead.  The workgroup that owns units [u0, u0+16) already holds its own
// d(gates) slice (G*16 values per row, just computed by its gate math), so it multiplies that slice with the
// matching G*16 ROWS of W_hh (resident in VGPRs as MFMA B-fragments) and gets a partial dh for ALL H units;
// the partial destined for units [16c, 16c+16) is written to consumer c's block of the exchange ring, and each
// consumer sums the P partials of its own 16 units.  Exchanged per batch row and step: H fp32 written + H fp32
// read per workgroup-slice, instead of G*H read - 4x (LSTM) / 3x (GRU) fewer bytes through the fabric, and
// every block has exactly one reader.
//
// Hand-off: "data is the flag" with a 2-bit step tag.  A partial is published with the two low mantissa bits
// replaced by (step mod 3); the ring has two time slots (step parity) and is filled with 0xFFFFFFFF (tag 3) before
// the launch, so a word carries the tag of step s only once step s's value has been written (slot reuse: the word
// written two steps earlier carries tag (s-2) mod 3 != s mod 3).  Reuse is safe with two slots because a producer
// can only reach step s after it has consumed every other workgroup's step s-1 partials, and those were produced
// after their owners finished reading step s-2.  Consumers mask the tag off before summing (|error| <= 3 ulp of
// the partial, far below the 2^-18 of the split product).  Loads are sc1 (each block has one reader, nothing to
// share in L2), every spin is bounded and sets the sticky error word.
#pragma once
#include "rnn_split_kernel.h"

#ifndef AAS_GRU_BWD_LKS
#define AAS_GRU_BWD_LKS 2
#endif

namespace {

// MODE = LSTM_BWD or GRU_BWD; U = units per workgroup (16: 256 threads, 32: 512 threads - half as many slices, so
// half the exchanged bytes chip-wide); NTW = 16-column result tiles per wave (P <= 4*NTW)
//
// R4 (exact mode, row groups of <= 8 rows): the partial product on v_mfma_f32_4x4x1_16B_f32 - sixteen independent 4 x 4 x 1 blocks
// per instruction, here 64 columns x 4 batch rows x one k (block b, D[lane 4b+j][reg i] = A[lane 4b+i] * B[lane 4b+j]: A = the W
// value of column 4b+i, B = d(gates) of row j).  A 16 x 16 x 4 tile spends its 32 cycles on 16 rows whether 4 or 16 of them
// exist; this form spends 8 cycles per k on every 4 rows that do, at the same peak rate - E's BPTT (N=30 over the whole chip:
// 4 rows per group; over half of it: 8) issues a quarter / half of the MFMA cycles.
//
// X6 (aas_set_precision(2), LSTM): the partial product as SIX bf16 products of three-term operands - d(gates) = h + m + l and
// W = h' + m' + l' exactly, hh' + hm' + mh' + mm' + hl' + lh' (dropped <= 2^-25 relative: the fp32-equivalent form of
// gemm_planes.hip) - on v_mfma_f32_16x16x32_bf16: 6 x 16 cycles per 32 k instead of the exact mode's 8 x 32.  W's h' and m'
// fragments stay in registers, the l' fragments (used once per tile and k-step) live in LDS (128 KB); the partials are tagged with
// round-to-nearest as in the exact mode.
template <int MODE, int U, int NTW, int LKSP = -1, bool EX = false, bool R4 = false, bool X6 = false>
__global__ __launch_bounds__(16 * U, 1) void rnn_bwd_rs_kernel(RnnP p) {
    using C = Cfg<MODE>;
    constexpr int G = C::G;
    constexpr bool LSTM = (MODE == LSTM_BWD);
    static_assert(U == 16 || U == 32, "16 rows x U units = one (row, unit) per thread");
    constexpr int THREADS = 16 * U;
    constexpr int KSTEPS = (G * U + 31) / 32;             // 32-wide k steps over my d(gates) slice (k' = gate*U + unit)
    constexpr int RS_LDA = KSTEPS * 32 + 8;               // bf16 elements per A-tile row (+8 pad: conflict-free 16-byte reads)
    constexpr int KPP = NTW;                              // producers per lane in the consumer-side sum (4 lanes share a row)
    constexpr int UQ = U / 4;                             // 16-byte unit quads per block
    constexpr int TPC = U / 16;                           // result tiles per consumer slice
    // my d(gates) slice as the MFMA operand tile, double-buffered by step parity: split - bf16 hi and lo tiles; exact - one fp32
    // tile (EX_LDA floats per row: +4 keeps the two 16-byte reads of a lane apart from its neighbours' banks)
    constexpr int EX_LDA = KSTEPS * 32 + 4;
    __shared__ __attribute__((aligned(16))) unsigned short a_hi[EX ? 1 : 2][EX ? 1 : 16][EX ? 8 : RS_LDA];
    __shared__ __attribute__((aligned(16))) unsigned short a_lo[EX ? 1 : 2][EX ? 1 : 16][EX ? 8 : RS_LDA];
    __shared__ __attribute__((aligned(16))) float a_f[EX ? 2 : 1][EX ? 16 : 1][EX ? EX_LDA : 4];
    __shared__ __attribute__((aligned(16))) unsigned short a_mi[X6 ? 2 : 1][X6 ? 16 : 1][X6 ? RS_LDA : 8];
    static_assert(!X6 || (!EX && !R4 && LKSP < 0), "X6 is an arithmetic of its own");
    // the 1000-unit GRU at U = 32 needs 192 VGPRs for its W fragments alone (of 256 at two waves per SIMD): the lo
    // fragments of its last k-step live in LDS instead (64 KB, re-read once per step) so that nothing spills
    constexpr int LKS = X6 ? KSTEPS : (LKSP >= 0 ? LKSP : ((!LSTM && U == 32 && NTW == 8) ? AAS_GRU_BWD_LKS : 0));
    __shared__ u32x4 bl_lds[LKS ? LKS * NTW : 1][LKS ? THREADS : 1];
    static_assert(!R4 || (EX && LKS == 0 && NTW % 4 == 0), "R4: exact mode, all of W in registers, 64-column groups per wave");
    constexpr int CG = R4 ? NTW / 4 : 1;                  // 64-column groups per wave
    constexpr int KT = KSTEPS * 32;                       // k extent of my d(gates) slice (padded)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware launch (p.xcd != 0; 8 (direction, row group) sets): 1-D grid, workgroup b -> set b % 8, slice b / 8.  Workgroups
    // are dealt round-robin over the 8 XCDs, so the P workgroups of a set - the only ones that exchange data - share an XCD
    // and its L2.
    // p.xcd == 2: 4 sets of (up to 64) workgroups, each spread over the XCD classes s and s + 4 (the 1000-unit GRU: a set is
    // 32 workgroups, a launch beside another chain gets 16 CUs per XCD): set b % 4, slice 2*(b / 8) + (b % 8) / 4.  A partial
    // block then stays in L2 exactly when its ONE consumer sits on the producer's XCD (per-consumer mask from the XCC table).
    const int xidx = (int)(blockIdx.x >> 3);        // p.xcd == 1: XCD class b % 8 hosts the sets class + 8 j, P workgroups each
    const int xset = p.xcd == 2 ? (int)(blockIdx.x & 3) : (int)(blockIdx.x & 7) + 8 * (xidx / p.P);
    const int pslice = p.xcd == 2 ? (int)(2 * (blockIdx.x >> 3) + ((blockIdx.x & 7) >> 2)) : p.xcd ? xidx % p.P : (int)blockIdx.x;
    const int qg = p.xcd ? (xset >> 1) : (int)blockIdx.y;
    const int d = p.xcd ? (xset & 1) : (int)blockIdx.z;
    const int T = p.T, N = p.N, H = p.H, GH = G * H, P = p.P;
    const int u0 = pslice * U;
    const int q0 = p.n0 + qg * p.rpg;
    const int NB = min(p.n1, q0 + p.rpg);
    const int nrows = NB - q0;                            // valid batch rows of this group (<= 16)
    unsigned* err = p.sync + ERR_WORD;

    // ---- B fragments: rows {g*H + u0 + u} of W_hh (k' = g*16 + u, padded to 64) x all Hp columns --------------
    const float* W = d == 0 ? p.w_hh : p.w_hh_r;
    u32x4 b0[R4 ? 1 : KSTEPS][R4 ? 1 : NTW], b1[R4 ? 1 : KSTEPS][R4 ? 1 : NTW];     // (rnn_split_kernel.h: frag_make)
    float wr[R4 ? CG : 1][R4 ? KT : 1];         // R4: W[k' row][my column] - lane l of column group cg owns column (wave*NTW*16 + cg*64 + l)
    if constexpr (R4) {
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            const int col = wave * NTW * 16 + cg * 64 + lane;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                const int gate = kk / U, unit = u0 + kk % U;
                wr[cg][kk] = (gate < G && unit < H && col < H) ? W[(int64_t)(gate * H + unit) * H + col] : 0.f;
            }
        }
    } else {
        const int n = lane & 15, q = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int col = (wave * NTW + nt) * 16 + n;       // unit whose dh this column feeds
                float wv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int kk = ks * 32 + q * 8 + e;
                    const int gate = kk / U, unit = u0 + kk % U;
                    wv[e] = (gate < G && unit < H && col < H) ? W[(int64_t)(gate * H + unit) * H + col] : 0.f;
                }
                u32x4 w0, w1;
                if constexpr (X6) {   // h' and m' in registers, l' in LDS (every k-step)
                    unsigned h[8], m[8], l[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) split3_bf16(wv[e], h[e], m[e], l[e]);
                    b0[ks][nt] = (u32x4){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
                    b1[ks][nt] = (u32x4){m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16)};
                    bl_lds[ks * NTW + nt][tid] = (u32x4){l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
                } else {
                frag_make<EX>(wv, w0, w1);
                b0[ks][nt] = w0;
                if (ks >= KSTEPS - LKS) bl_lds[LKS ? (ks - (KSTEPS - LKS)) * NTW + nt : 0][LKS ? tid : 0] = w1;
                else b1[ks][nt] = w1;
                }
            }
    }
    xchg_clean_other_half(p);           // managed exchange buffer: rnn_split_kernel.h
    // plain (L2-resident) publish stores when the whole set shares an XCD (rnn_split_kernel.h: xcd_set_colocated)
    __shared__ unsigned long long xcd_flag;
    unsigned long long local = 0ull;    // bit c: consumer slice c shares this workgroup's XCD
    if (p.xcd) {
        unsigned* tab = p.xchg + (int64_t)4 * p.N * p.P * p.P * U;   // behind the ring, poisoned by the same memset
        local = xcd_peer_mask(tab, xset, pslice, p.P, p.sync + ERR_WORD, p.tag, &xcd_flag, p.sync + XSTAT_WORD + 4 * (MODE == LSTM_BWD ? 1 : 3));
        if (p.flags & 524288) local = 0ull;
    }
    // zero both A tiles once: pad rows / pad gate columns stay zero for the whole launch
    if constexpr (EX) {
        for (int i = tid; i < 2 * 16 * EX_LDA; i += THREADS) (&a_f[0][0][0])[i] = 0.f;
    } else {
        for (int i = tid; i < 2 * 16 * RS_LDA; i += THREADS) {
            (&a_hi[0][0][0])[i] = 0;
            (&a_lo[0][0][0])[i] = 0;
            if constexpr (X6) (&a_mi[0][0][0])[i] = 0;
        }
    }
    __syncthreads();

    // exchange ring: [slot 2][dir 2][N rows][consumer P][producer P][U units] tagged fp32
    unsigned* ring = p.xchg;
    const int64_t ring_words = (int64_t)4 * N * P * P * U;
    auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)ring, 0, (int)(ring_words * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // consumer-side thread role: 4 lanes (ppg) share one (row, unit quad) and split the producers
    const int ppg = lane & 3, uq = (lane >> 2) & (UQ - 1), rl = lane / (4 * UQ);
    const int row = wave * (64 / (4 * UQ)) + rl;          // batch row within the group
    const int unit = uq * 4 + ppg;                        // after the 4-lane reduction each lane keeps one unit
    const int gr = q0 + row, gunit = u0 + unit;
    const bool rowok = row < nrows;
    const bool ok = rowok && gunit < H;

    float carry = 0.f;
    unsigned long long ph[5] = {0, 0, 0, 0, 0};
    const bool stamp = (p.flags & 64) && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    for (int s = 0; s < T; ++s) {
        unsigned long long st0 = stamp ? __builtin_amdgcn_s_memrealtime() : 0ull, st1 = st0, st2 = st0, st3 = st0;
        const int t = (d == 0) ? T - 1 - s : s;           // BPTT runs against the direction's forward order
        const int64_t tn = (int64_t)t * N + gr;

        // ---- private inputs of the step ---------------------------------------------------------
        float dyv = 0.f, sav[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ok) {
            dyv = p.dy[tn * H + gunit];
            const f32x4 ga4 = *reinterpret_cast<const f32x4*>(p.gact + (((int64_t)d * T * N + tn) * H + gunit) * 4);
            sav[0] = ga4.x; sav[1] = ga4.y; sav[2] = ga4.z; sav[3] = ga4.w;
            const int tq = (d == 0) ? t - 1 : t + 1;
            const bool hasq = (tq >= 0 && tq < T);
            const int64_t qn = ((int64_t)d * T * N + (int64_t)tq * N + gr) * H + gunit;
            if (LSTM) {
                sav[4] = p.cst[((int64_t)d * T * N + tn) * H + gunit];
                sav[5] = hasq ? p.cst[qn] : 0.f;
            } else {
                sav[5] = hasq ? p.hout[qn] : 0.f;
            }
        }

        // ---- recurrent term: sum over producers of the partial dh of my 16 units -----------------
        float rec = 0.f;
        if (s > 0) {
            const unsigned tag = (unsigned)((s - 1) % 3);
            const int slot = (s - 1) & 1;
            const int64_t blk0 = (((int64_t)(slot * 2 + d) * N + q0) * P + pslice) * P;   // (row q0, consumer me, producer 0), in blocks of U words
            auto poll = [&]() {
                // poll one word per producer (last valid row of the group) before streaming the block
                const bool probe = lane < P && !(p.flags & 1);
                const unsigned* wp = ring + (blk0 + (int64_t)(nrows - 1) * P * P + lane) * U + (U - 1);
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                unsigned spins = 0;
                while (true) {
                    const unsigned w = probe ? __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
                    if (!__any((w & 3u) != tag)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 0) {
                        if (ld_cnt(err) != 0) break;
                        if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) {
                            if (lane == 0) __hip_atomic_store(err, (unsigned)p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
            };
            // Stream the block first: the workgroup that arrives last - the one on the critical path - finds the step's
            // partials complete and saves the poll's round trip (LSTM 3.17 -> 2.83 us / step); only when a tag is missing
            // does the wave fall back to the cheap poll and then re-read.  Debug flag 8192: poll first.
            bool polled = (p.flags & 8192) != 0;
            if (polled && !(p.flags & 4)) poll();
            if (stamp) st1 = __builtin_amdgcn_s_memrealtime();
            // lane (rl, uq, ppg) loads units [4uq, 4uq+4) of producers pp = 4k + ppg: 256 contiguous bytes per row and k
            const unsigned rbase = (rowok && !(p.flags & 1)) ? (unsigned)(((blk0 + (int64_t)row * P * P) * U + uq * 4) * 4) : OOB;
            u32x4 v[KPP];
            unsigned spins = 0;
            unsigned long long t0 = 0;
            while (true) {
                unsigned bad = 0u;
#pragma unroll
                for (int k = 0; k < KPP; ++k) {
                    const int pp = k * 4 + ppg;
                    const unsigned off = (pp < P) ? rbase + (unsigned)(pp * U * 4) : OOB;
                    v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 16));
                }
#pragma unroll
                for (int k = 0; k < KPP; ++k) {
                    const int pp = k * 4 + ppg;
                    const unsigned m = ((v[k].x ^ tag) | (v[k].y ^ tag) | (v[k].z ^ tag) | (v[k].w ^ tag)) & 3u;
                    bad |= (pp < P && rbase != OOB) ? m : 0u;
                }
                if ((p.flags & 4) || !__any(bad != 0u)) break;
                if (!polled) { poll(); polled = true; continue; }
                if (spins == 0) t0 = __builtin_amdgcn_s_memrealtime();
                if ((++spins & 63u) == 0) {
                    if (ld_cnt(err) != 0) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) {
                        if (lane == 0) __hip_atomic_store(err, (unsigned)p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KPP; ++k) {
                s4[0] += __uint_as_float(v[k].x & ~3u);
                s4[1] += __uint_as_float(v[k].y & ~3u);
                s4[2] += __uint_as_float(v[k].z & ~3u);
                s4[3] += __uint_as_float(v[k].w & ~3u);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s4[j] += __shfl_xor(s4[j], 1, 64);
                s4[j] += __shfl_xor(s4[j], 2, 64);
            }
            rec = ppg == 0 ? s4[0] : (ppg == 1 ? s4[1] : (ppg == 2 ? s4[2] : s4[3]));
        }
        if (stamp) st2 = __builtin_amdgcn_s_memrealtime();

        // ---- gate math -----------------------------------------------------------------------------
        float xv[4] = {0.f, 0.f, 0.f, 0.f};
        float dn_keep = 0.f;
        if (ok) {
            if (LSTM) {
                const float dh = dyv + rec;
                const float ig = sav[0], fg = sav[1], gg = sav[2], og = sav[3];
                const float c = sav[4], cp = sav[5];
                const float tc = tanhf_(c);
                const float dc = dh * og * (1.f - tc * tc) + carry;
                carry = dc * fg;
                xv[0] = dc * gg * ig * (1.f - ig);
                xv[1] = dc * cp * fg * (1.f - fg);
                xv[2] = dc * ig * (1.f - gg * gg);
                xv[3] = dh * tc * og * (1.f - og);
            } else {
                const float dh = dyv + rec + carry;
                const float rg = sav[0], zg = sav[1], ng = sav[2], hn = sav[3];
                const float hp = sav[5];
                carry = dh * zg;
                const float dnp = dh * (1.f - zg) * (1.f - ng * ng);
                const float dzp = dh * (hp - ng) * zg * (1.f - zg);
                const float drp = dnp * hn * rg * (1.f - rg);
                xv[0] = drp; xv[1] = dzp; xv[2] = dnp * rg;
                dn_keep = dnp;
            }
        }
        const int par = s & 1;
        unsigned h16[4] = {0, 0, 0, 0}, l16[4] = {0, 0, 0, 0}, m16[4] = {0, 0, 0, 0};
        if constexpr (X6) {
#pragma unroll
            for (int g = 0; g < G; ++g) split3_bf16(xv[g], h16[g], m16[g], l16[g]);
        } else if constexpr (!EX) {
#pragma unroll
            for (int g = 0; g < G; ++g) split_bf16(xv[g], h16[g], l16[g]);
        }
        if (rowok && s + 1 < T) {  // my d(gates) slice as the operand tile of the partial product
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if constexpr (EX) a_f[par][row][g * U + unit] = xv[g];
                else {
                    a_hi[par][row][g * U + unit] = (unsigned short)h16[g];
                    a_lo[par][row][g * U + unit] = (unsigned short)l16[g];
                    if constexpr (X6) a_mi[par][row][g * U + unit] = (unsigned short)m16[g];
                }
            }
        }
        if (X6 && p.dgp1) {
            // d(gates) straight as the two three-term plane sets of the six-product GEMMs (aas_split_planes3's form: row (t, n) =
            // Q1 | Q2, each dgKp columns of 128-byte blocks; Q1 block = 64 B of m | 64 B of h, Q2 block = 64 B of h | 64 B of l)
            if (ok) {
                char* r1 = reinterpret_cast<char*>(p.dgp1) + tn * (int64_t)p.dgKp * 8;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int k = d * GH + g * H + gunit;
                    char* o = r1 + (k >> 5) * 128 + (k & 31) * 2;
                    *reinterpret_cast<unsigned short*>(o) = (unsigned short)m16[g];
                    *reinterpret_cast<unsigned short*>(o + 64) = (unsigned short)h16[g];
                    *reinterpret_cast<unsigned short*>(o + (int64_t)p.dgKp * 4) = (unsigned short)h16[g];
                    *reinterpret_cast<unsigned short*>(o + (int64_t)p.dgKp * 4 + 64) = (unsigned short)l16[g];
                }
            }
            if (rowok && d == 1 && pslice == P - 1) {   // zero pad columns [2*G*H, Kp) of my rows in both sets
                for (int k = 2 * GH + unit; k < p.dgKp; k += U) {
                    char* o = reinterpret_cast<char*>(p.dgp1) + tn * (int64_t)p.dgKp * 8 + (k >> 5) * 128 + (k & 31) * 2;
                    *reinterpret_cast<unsigned short*>(o) = 0;
                    *reinterpret_cast<unsigned short*>(o + 64) = 0;
                    *reinterpret_cast<unsigned short*>(o + (int64_t)p.dgKp * 4) = 0;
                    *reinterpret_cast<unsigned short*>(o + (int64_t)p.dgKp * 4 + 64) = 0;
                }
            }
        } else if (p.dgp1) {
            // d(gates) for the layer's GEMMs straight in their operand form: row (t, n) of interleaved bf16 hi | lo planes
            // (per 32-wide k block 64 B of hi then 64 B of lo), k = d*G*H + g*H + unit - no fp32 copy, no split pass
            if (ok) {
                char* r1 = reinterpret_cast<char*>(p.dgp1) + tn * (int64_t)p.dgKp * 4;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int k = d * GH + g * H + gunit;
                    char* o = r1 + (k >> 5) * 128 + (k & 31) * 2;
                    *reinterpret_cast<unsigned short*>(o) = (unsigned short)h16[g];
                    *reinterpret_cast<unsigned short*>(o + 64) = (unsigned short)l16[g];
                }
                if (!LSTM) {
                    char* r2 = reinterpret_cast<char*>(p.dgp2) + tn * (int64_t)p.dgKp * 4;
                    unsigned hn, ln;
                    split_bf16(dn_keep, hn, ln);
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const int k = d * GH + g * H + gunit;
                        char* o = r2 + (k >> 5) * 128 + (k & 31) * 2;
                        *reinterpret_cast<unsigned short*>(o) = (unsigned short)(g == 2 ? hn : h16[g]);
                        *reinterpret_cast<unsigned short*>(o + 64) = (unsigned short)(g == 2 ? ln : l16[g]);
                    }
                }
            }
            // zero pad columns [2*G*H, Kp) of my rows (the last slice of the reverse direction owns them)
            if (rowok && d == 1 && pslice == P - 1) {
                for (int k = 2 * GH + unit; k < p.dgKp; k += U) {
                    char* o1 = reinterpret_cast<char*>(p.dgp1) + tn * (int64_t)p.dgKp * 4 + (k >> 5) * 128 + (k & 31) * 2;
                    *reinterpret_cast<unsigned short*>(o1) = 0;
                    *reinterpret_cast<unsigned short*>(o1 + 64) = 0;
                    if (!LSTM) {
                        char* o2 = reinterpret_cast<char*>(p.dgp2) + tn * (int64_t)p.dgKp * 4 + (k >> 5) * 128 + (k & 31) * 2;
                        *reinterpret_cast<unsigned short*>(o2) = 0;
                        *reinterpret_cast<unsigned short*>(o2 + 64) = 0;
                    }
                }
            }
        } else if (ok) {  // fp32 d(gates) (plain stores)
            float* dg = p.dg1 + (tn * 2 + d) * GH + gunit;
#pragma unroll
            for (int g = 0; g < G; ++g) dg[g * H] = xv[g];
            if (!LSTM) {
                float* dx_ = p.dg2 + (tn * 2 + d) * GH + gunit;
                dx_[0] = xv[0]; dx_[H] = xv[1]; dx_[2 * H] = dn_keep;
            }
        }
        __syncthreads();
        if (stamp) st3 = __builtin_amdgcn_s_memrealtime();

        // ---- partial dh for all units: [16 rows x 64] x [64 x Hp], published to the consumers' blocks ---
        if constexpr (R4) {
            if (s + 1 < T) {
                const unsigned tag = (unsigned)(s % 3);
                const int slot = s & 1;
                const int j = lane & 3, b4 = (lane >> 2) * 4;        // batch row within the 4-row block; first of my 4 result columns
                const int nrb = (nrows + 3) >> 2;
                for (int rb = 0; rb < nrb; ++rb) {
                    const int prow = rb * 4 + j;
                    const float* arow = &a_f[par][prow][0];
#pragma unroll
                    for (int cg = 0; cg < CG; ++cg) {
                        // four independent accumulation chains (k mod 4): back-to-back issue without waiting on the previous result
                        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
                        if (!(p.flags & 2)) {
                            // (the tile reads run AHEAD of the MFMAs that use them: eight 16-byte reads in flight)
                            constexpr int AH = 8;
                            f32x4 dq[AH];
#pragma unroll
                            for (int i = 0; i < AH; ++i) dq[i] = *reinterpret_cast<const f32x4*>(arow + i * 4);
#pragma unroll
                            for (int k4 = 0; k4 < KT / 4; ++k4) {
                                const f32x4 dv = dq[k4 % AH];
                                if (k4 + AH < KT / 4) dq[k4 % AH] = *reinterpret_cast<const f32x4*>(arow + (k4 + AH) * 4);
                                __builtin_amdgcn_sched_barrier(0);   // (keep the read ahead: the scheduler would sink it to its use)
                                c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[cg][k4 * 4 + 0], dv[0], c0, 0, 0, 0);
                                c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[cg][k4 * 4 + 1], dv[1], c1, 0, 0, 0);
                                c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[cg][k4 * 4 + 2], dv[2], c2, 0, 0, 0);
                                c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[cg][k4 * 4 + 3], dv[3], c3, 0, 0, 0);
                            }
                        }
                        const f32x4 acc = (c0 + c1) + (c2 + c3);
                        const int col = wave * NTW * 16 + cg * 64 + b4;      // lane holds columns col .. col+3 of batch row prow
                        const int c = col / U;
                        if (c < P && prow < nrows && !(p.flags & 8)) {
                            u32x4 o;    // (tag in the two low mantissa bits, rounded to nearest: see below)
                            o.x = ((__float_as_uint(acc[0]) + 2u) & ~3u) | tag;
                            o.y = ((__float_as_uint(acc[1]) + 2u) & ~3u) | tag;
                            o.z = ((__float_as_uint(acc[2]) + 2u) & ~3u) | tag;
                            o.w = ((__float_as_uint(acc[3]) + 2u) & ~3u) | tag;
                            const int64_t rblk = ((int64_t)(slot * 2 + d) * N + q0 + prow) * P;
                            const unsigned boff = (unsigned)((((rblk + c) * P + pslice) * U + col % U) * 4);
                            if ((local >> c) & 1ull) __builtin_amdgcn_raw_buffer_store_b128(o, rs_x, (int)boff, 0, 0);
                            else __builtin_amdgcn_raw_buffer_store_b128(o, rs_x, (int)boff, 0, 16);
                        }
                    }
                }
            }
        } else
        if (s + 1 < T) {
            const int m = lane & 15, q = lane >> 4;
            u32x4 ah[KSTEPS], al[KSTEPS], am[X6 ? KSTEPS : 1];
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                if constexpr (X6) am[ks] = *reinterpret_cast<const u32x4*>(&a_mi[par][m][ks * 32 + q * 8]);
                if constexpr (EX) {
                    ah[ks] = *reinterpret_cast<const u32x4*>(&a_f[par][m][ks * 32 + q * 8]);
                    al[ks] = *reinterpret_cast<const u32x4*>(&a_f[par][m][ks * 32 + q * 8 + 4]);
                } else {
                    ah[ks] = *reinterpret_cast<const u32x4*>(&a_hi[par][m][ks * 32 + q * 8]);
                    al[ks] = *reinterpret_cast<const u32x4*>(&a_lo[par][m][ks * 32 + q * 8]);
                }
            }
            const unsigned tag = (unsigned)(s % 3);
            const int slot = s & 1;
            // Transposed product (W fragment as the A operand, my d(gates) tile as B): the 16x16 result tile is
            // [unit][batch row], so a lane holds 4 CONSECUTIVE units of one row - one 16-byte write-through store
            // per tile and lane, 64 contiguous bytes per row, instead of four 4-byte stores.
            const int prow = lane & 15, u4 = (lane >> 4) * 4;
            const int64_t rblk = ((int64_t)(slot * 2 + d) * N + q0 + prow) * P;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (!(p.flags & 2)) {
#pragma unroll
                    for (int ks = 0; ks < KSTEPS; ++ks) {
                        u32x4 w1;
                        if (ks >= KSTEPS - LKS) w1 = bl_lds[LKS ? (ks - (KSTEPS - LKS)) * NTW + nt : 0][LKS ? tid : 0];
                        else w1 = b1[ks][nt];
                        if constexpr (X6) {   // the small terms first; W is the MFMA's first operand (transposed product, see above)
                            const bf16x8 wh = __builtin_bit_cast(bf16x8, b0[ks][nt]), wm = __builtin_bit_cast(bf16x8, b1[ks][nt]);
                            const bf16x8 wl = __builtin_bit_cast(bf16x8, w1);
                            const bf16x8 dh = __builtin_bit_cast(bf16x8, ah[ks]), dm = __builtin_bit_cast(bf16x8, am[ks]), dl = __builtin_bit_cast(bf16x8, al[ks]);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dl, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, dh, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, dm, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dm, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, dh, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dh, acc, 0, 0, 0);
                        } else if constexpr (EX) acc = mma_chunk<true>(acc, b0[ks][nt], w1, ah[ks], al[ks]);
                        else {   // (the order of the three products is part of the bit-exact contract between the kernel variants)
                            const bf16x8 whi = __builtin_bit_cast(bf16x8, b0[ks][nt]), wlo = __builtin_bit_cast(bf16x8, w1);
                            const bf16x8 dhi = __builtin_bit_cast(bf16x8, ah[ks]), dlo = __builtin_bit_cast(bf16x8, al[ks]);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, dlo, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, dhi, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, dhi, acc, 0, 0, 0);
                        }
                    }
                }
                const int tile = wave * NTW + nt;
                const int c = tile / TPC, half = tile % TPC;   // consumer slice, 16-unit half of its block
                if (c < P && prow < nrows && !(p.flags & 8)) {
                    // the tag replaces the two low mantissa bits: exact mode rounds to nearest there (an unbiased 22-bit partial,
                    // |error| <= 2 ulp, below the rounding noise of any fp32 summation order); split mode truncates as before
                    constexpr unsigned RND = (EX || X6) ? 2u : 0u;
                    u32x4 o;
                    o.x = ((__float_as_uint(acc[0]) + RND) & ~3u) | tag;
                    o.y = ((__float_as_uint(acc[1]) + RND) & ~3u) | tag;
                    o.z = ((__float_as_uint(acc[2]) + RND) & ~3u) | tag;
                    o.w = ((__float_as_uint(acc[3]) + RND) & ~3u) | tag;
                    const unsigned boff = (unsigned)((((rblk + c) * P + pslice) * U + half * 16 + u4) * 4);
                    if ((local >> c) & 1ull) __builtin_amdgcn_raw_buffer_store_b128(o, rs_x, (int)boff, 0, 0);   // the line stays in this XCD's L2
                    else __builtin_amdgcn_raw_buffer_store_b128(o, rs_x, (int)boff, 0, 16);            // sc1: agent-scope write-through
                }
            }
        }
        if (stamp) {
            const unsigned long long st4 = __builtin_amdgcn_s_memrealtime();
            ph[0] += st1 - st0; ph[1] += st2 - st1; ph[2] += st3 - st2; ph[3] += st4 - st3; ph[4] += st4 - st0;
        }
    }
    if (stamp && tid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.sync + STAMP_WORD);
        for (int i = 0; i < 5; ++i) o[i] = ph[i];
    }
}

inline size_t rs_ring_bytes(int N, int H, int U) {
    const size_t P = (size_t)cdiv(H, U);
    return (size_t)4 * N * P * P * U * 4;
}

template <int MODE, int U, bool EX, bool X6 = false>
int launch_rs(const RnnP& p, hipStream_t s) {
    dim3 grid(p.P, p.Q, 2), block(16 * U);
    if (p.xcd) grid = dim3(p.P * p.Q * 2);
    if constexpr (X6) {
        if constexpr (MODE == LSTM_BWD && U == 32) {
            if (p.P > 8 && p.P <= 16) {
                hipLaunchKernelGGL((rnn_bwd_rs_kernel<MODE, U, 4, -1, false, false, true>), grid, block, 0, s, p);
                return 0;
            }
        }
        return -1;
    } else {
    if constexpr (EX && MODE == LSTM_BWD && U == 32) {
        // <= 8 rows per group: the 4 x 4 x 1 block form (debug bit 268435456: the 16 x 16 x 4 tiles)
        if (p.rpg <= 8 && p.P > 8 && p.P <= 16 && !(p.flags & 268435456)) {
            hipLaunchKernelGGL((rnn_bwd_rs_kernel<MODE, U, 4, -1, true, true>), grid, block, 0, s, p);
            return 0;
        }
    }
    if (p.P <= 8) hipLaunchKernelGGL((rnn_bwd_rs_kernel<MODE, U, 2, -1, EX>), grid, block, 0, s, p);
    else if (p.P <= 16) hipLaunchKernelGGL((rnn_bwd_rs_kernel<MODE, U, 4, -1, EX>), grid, block, 0, s, p);
    else if (p.P <= 32) {
        // (LSTM layers wider than 512 units never get here with 32-unit slices: 128 gate rows x 1024 k of W^T do not fit the
        //  register file of a 512-thread workgroup - that instance spilled 190-256 VGPRs - so run_bwd_rs gives them 16-unit slices)
        if constexpr (MODE == LSTM_BWD && U == 32) return -1;
        else hipLaunchKernelGGL((rnn_bwd_rs_kernel<MODE, U, 8, -1, EX>), grid, block, 0, s, p);
    }
    else if (U == 16 && p.P <= 64) hipLaunchKernelGGL((rnn_bwd_rs_kernel<MODE, 16, 16, -1, EX>), grid, block, 0, s, p);
    else return -1;
    return 0;
    }
}

template <int MODE, bool EX = false, bool X6 = false>
int run_bwd_rs(const char* name, RnnP p, hipStream_t s) {
    AAS_CHECK(p.T >= 1 && p.N >= 1 && p.H >= 1, "%s: bad sizes T=%d N=%d H=%d", name, p.T, p.N, p.H);
    const int cus = aas_rnn_cus();
    AAS_CHECK(cus > 0, "%s: no HIP device", name);
    p.flags = aas_debug_flags_value();
    p.tag = aas_rnn_launch_tag_value();
    // 32-unit slices (512-thread workgroups) halve the number of slices and with it the bytes every step moves
    // through the fabric; small layers keep 16-unit slices so that enough workgroups share the work
    // (LSTM layers of 512 < H <= 1024 units - legal: AM_training/train.py:46,203-204 - take 16-unit slices: 455 VGPRs in 256-thread
    //  workgroups, no scratch)
    const int U = (p.H >= 256 && !(p.flags & 512) && !(MODE == LSTM_BWD && p.H > 512)) ? 32 : 16;
    p.P = cdiv(p.H, U);
    if (X6 && !(MODE == LSTM_BWD && U == 32 && p.P > 8 && p.P <= 16)) return -1;   // (before anything is queued)
    if (p.P > (U == 16 ? 64 : 32)) return -1;              // one poll lane per producer, <= 8/16 tiles per wave
    AAS_CHECK(p.P * 2 <= cus, "%s: H=%d needs %d resident workgroups, device has %d CUs", name, p.H, p.P * 2, cus);
    const size_t rbytes = rs_ring_bytes(p.N, p.H, U);
    if (rbytes >= 0x7fffffffULL) return -1;
    // rows per group: the smallest of 4 / 8 / 16 whose grid is still resident (fewest bytes per workgroup and step);
    // larger batches run as consecutive launches over row chunks
    int rpg = 16;
    for (int cand = 4; cand < 16; cand *= 2)
        if (p.P * cdiv(p.N, cand) * 2 <= cus) { rpg = cand; break; }
    p.rpg = rpg;
    const int qmax = cus / (p.P * 2) < 1 ? 1 : cus / (p.P * 2);
    // tag 3 = "no step's value yet" (+ the XCC table): the poison memset - or, on a managed buffer (one launch for the batch), this
    // launch's half, poisoned by its predecessor from inside its kernel
    AasXchgPlan plan = {};
    if (p.N <= qmax * rpg) aas_xchg_plan(p.xchg, rbytes + XCD_TAB_BYTES, s, &plan);
    if (plan.managed) {
        p.xchg = plan.base; p.clean_ptr = plan.clean_ptr; p.clean_words = plan.clean_words;
    } else {
        if (aas_xchg_legacy_fill(p.xchg, rbytes + XCD_TAB_BYTES, s)) return 2;
    }
    for (int n0 = 0; n0 < p.N; n0 += qmax * rpg) {
        p.n0 = n0;
        const int rows = (p.N - n0) < qmax * rpg ? (p.N - n0) : qmax * rpg;
        p.n1 = n0 + rows;
        p.Q = cdiv(rows, rpg);
        // 8 (direction, row group) sets of at most 32 workgroups: XCD-aware grid (debug bit 262144: the plain 3-D grid,
        // 524288: XCD-aware grid but write-through publish stores)
        p.xcd = (p.flags & 262144) ? 0 : ((p.Q * 2) % 8 == 0 && p.P * (p.Q * 2 / 8) <= 32) ? 1 : (p.Q * 2 == 4 && p.P <= 64 && p.P % 2 == 0) ? 2 : 0;
        const int rc = (U == 32) ? launch_rs<MODE, 32, EX, X6>(p, s) : launch_rs<MODE, 16, EX, X6>(p, s);
        if (rc != 0) return -1;
        AAS_LAUNCH_CHECK(name);
    }
    return 0;
}

// BPTT entry: the reduce-scatter kernel in the library's arithmetic mode (split-bf16, or exact fp32 when the caller takes fp32
// d(gates)); the all-gather split kernel under debug flag 256; the counter-based fp32 kernel otherwise (and under debug bit 134217728)
template <int MODE>
int run_bwd_any(const char* name, RnnP p, hipStream_t s) {
    if (p.xchg && aas_precision_value() == 1 && !(aas_debug_flags_value() & 256)) {
        const int rc = run_bwd_rs<MODE, false>(name, p, s);
        if (rc >= 0) return rc;
    }
    if (p.dgp1 && p.dgsets == 3) {   // three-term plane sets: only the six-product kernel of the fp32-equivalent mode writes them
        if (p.xchg && aas_precision_value() == 2 && !(aas_debug_flags_value() & (256 | 134217728 | 536870912))) {
            const int rc = run_bwd_rs<MODE, false, true>(name, p, s);
            if (rc >= 0) return rc;
        }
        aas_set_error("%s: three-term plane output needs the six-product BPTT kernel (precision 2, exchange buffer, 256 < H <= 512)", name);
        return 3;
    }
    if (p.dgp1) {   // only the split-bf16 reduce-scatter kernel writes operand planes: the caller falls back to fp32 d(gates) + a split pass
        aas_set_error("%s: plane output needs the split-bf16 reduce-scatter BPTT kernel (precision 1, exchange buffer, supported H)", name);
        return 3;
    }
    if (p.xchg && aas_precision_value() == 2 && !(aas_debug_flags_value() & (256 | 134217728 | 536870912))) {
        // fp32-equivalent mode: six-product bf16 form where it is instantiated (500-unit LSTM); debug bit 536870912: exact kernels
        const int rc = run_bwd_rs<MODE, false, true>(name, p, s);
        if (rc >= 0) return rc;
    }
    if (p.xchg && aas_precision_value() != 1 && !(aas_debug_flags_value() & (256 | 134217728))) {
        const int rc = run_bwd_rs<MODE, true>(name, p, s);
        if (rc >= 0) return rc;
    }
    return run_any<MODE>(name, p, s);
}

}  // namespace
