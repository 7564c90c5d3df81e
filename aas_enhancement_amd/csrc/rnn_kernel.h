// Persistent recurrent kernels for gfx950: bidirectional bias-free LSTM / GRU, forward and BPTT.
// Replaces the cuDNN RNN kernels under nn.LSTM / nn.GRU in the reference
// (Speech_enhancement_by_AAS/model.py:73-74,83-85,94-95,102-104).
//
// One launch runs all T time steps of both directions.  Grid = (P unit slices, Q batch groups,
// 2 directions), 256 threads (4 waves); every workgroup must be resident (P*Q*2 <= #CUs).
//   * The workgroup's slice of W_hh lives in VGPRs as MFMA B-fragments for the whole launch
//     (forward: rows {g*H + unit} x all H columns; BPTT: all G*H rows x the slice's columns).
//   * Per step each wave loads its K-range of the exchanged vector (h_{t-1}, or d(gates)_{t+1})
//     as 16-byte sc1 loads straight into MFMA A-fragments (k is permuted identically on A and B,
//     so a lane's float4 feeds 4 consecutive v_mfma_f32_16x16x4_f32), the 4 waves' partial sums
//     are reduced through LDS, gate math runs one (row, unit) per thread with the cell / carry
//     state held in registers across steps.
//   * Exchange between workgroups: write-through (sc1) stores of the slice, s_waitcnt vmcnt(0),
//     workgroup barrier, ONE agent-scope atomic add on the (direction, batch-group) arrival
//     counter; consumers poll that counter with sc1 loads and read with sc1 loads only
//     (MI355X_MICROARCH.md "Valid forms", table row 1).  Every spin is bounded; a timeout sets
//     sync[ERR] and releases every waiter.
#pragma once
#include "common.h"

namespace {

constexpr int SYNC_WORDS = 1024;  // 4 KiB of arrival counters, zeroed per launch
constexpr int ERR_WORD = 1024;    // sticky timeout flag, after the counters (caller zero-initialises once)
constexpr int SYNC_BYTES = (SYNC_WORDS + 64) * 4;
constexpr int STAMP_WORD = 1040;   // 8 x u64 phase totals of workgroup (0,0,0) wave 0 when debug flag 64 is set
constexpr int CNT_STRIDE = 16;  // one 64-B line per counter

// RNN_* = the vanilla tanh recurrence h_t = tanh(W_ih x_t + W_hh h_{t-1}) (nn.RNN, the `rnn` entry of supported_rnns, model.py:12-17):
// one "gate"; it runs on the counter-based kernel below in both precisions (no BASELINE configuration uses it)
enum Mode { LSTM_FWD = 0, LSTM_BWD = 1, GRU_FWD = 2, GRU_BWD = 3, RNN_FWD = 4, RNN_BWD = 5 };

struct RnnP {
    int T, N, H;
    const float* pre;    // fwd: [T,N,2,G*H]
    const float* w_hh;   // [G*H,H] forward direction
    const float* w_hh_r; // [G*H,H] reverse direction
    float* hout;         // [2,T,N,H]
    float* gact;         // [2,T,N,4H]
    float* cst;          // [2,T,N,H]   (LSTM)
    const float* dy;     // bwd: [T,N,H]
    float* dg1;          // bwd: LSTM dgates / GRU dgh  [T,N,2,G*H]  (exchanged)
    float* dg2;          // bwd: GRU dgx               [T,N,2,G*H]
    unsigned short* dgp1; // bwd, optional: dg1 / dg2 as interleaved bf16 hi|lo PLANES [T*N rows][dgKp] (k = d*G*H + g*H + unit)
    unsigned short* dgp2; //   instead of fp32 - the operand form of the layer's input-gradient and weight-gradient GEMMs
    int dgKp;
    int dgsets;          // 3: dgp1 is a three-term plane-set pair (Q1 | Q2 side by side, row pitch 8 dgKp bytes: aas_split_planes3's form)
    unsigned* sync;
    unsigned* xchg;      // split-bf16 exchange arrays (hi | lo), NULL -> exact fp32 kernels
    int P, Q;
    int n0, n1;          // batch rows [n0, n1) handled by this launch
    int rpg;             // batch rows per group (<= 16*MT): smaller groups = fewer exchanged bytes per workgroup
    int tag;             // written to the sticky error word when a bounded spin times out (aas_set_rnn_launch_tag)
    int flags;           // debug/ablation bits (aas_set_debug_flags): 1 no exchange loads, 2 no MFMA, 4 no wait, 8 no publish
    int xcd;             // bit 0: XCD-aware 1-D grid (8 sets of P workgroups, a set per XCD class); bit 1: plain publish stores
    // managed exchange buffer (aas_rnn_xchg_prepare; common.h: AasXchgPlan): the region of the OTHER half this launch poisons itself
    unsigned* clean_ptr;
    unsigned clean_words;
    // row classes (aas_set_rnn_row_classes; forward launches): rows < cls_n are live for t < cls_t0, the others for t < cls_t1; a dead
    // (t, row) gets h = c = 0 and stored gate values under which BPTT forms zero gate gradients and carries nothing on, whatever dh is
    // (LSTM: all zeros; GRU: z = 0, n = 1)
    int cls_n, cls_t0, cls_t1;
    int ring;            // exact forward kernels: h_t exchanged through 4 time slots ([2][4][N] rows), producers re-poison behind themselves
    // fused input projection (rnn_split_kernel.h, XF): `pre` is not read; the launch forms x_t W_ih^T itself from the layer input
    const float* xin;    // [T,N,I]
    const float* w_ih;   // [G*H,I] forward direction
    const float* w_ih_r; // [G*H,I] reverse direction
    int I;
};

__device__ __forceinline__ unsigned ld_cnt(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wait until *cnt >= target (thread 0 only).  Bounded: ~0.5 s of wall clock, then flags ERR.
__device__ __forceinline__ void wait_counter(unsigned* cnt, unsigned target, unsigned* err, unsigned tag) {
    if (ld_cnt(cnt) >= target) return;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    unsigned spins = 0;
    while (ld_cnt(cnt) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0) {
            if (ld_cnt(err) != 0) return;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) {
                __hip_atomic_store(err, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
    }
}

__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE> struct Cfg;
template <> struct Cfg<LSTM_FWD> { static constexpr int G = 4, U = 16, NT = 4; };
template <> struct Cfg<GRU_FWD>  { static constexpr int G = 3, U = 16, NT = 3; };
template <> struct Cfg<LSTM_BWD> { static constexpr int G = 4, U = 16, NT = 1; };
template <> struct Cfg<GRU_BWD>  { static constexpr int G = 3, U = 16, NT = 1; };
template <> struct Cfg<RNN_FWD>  { static constexpr int G = 1, U = 16, NT = 1; };
template <> struct Cfg<RNN_BWD>  { static constexpr int G = 1, U = 16, NT = 1; };

constexpr int red_ld(int NT, int U) {
    // row stride of the LDS reduction buffer: (ld mod 32) == 8 (U=8) or 16 (U=16) keeps the
    // (row, unit)-per-thread reads conflict-free
    int want = (U == 8) ? 8 : 16;
    int base = NT * 16;
    int pad = ((want - base) % 32 + 32) % 32;
    return base + pad;
}

// MODE, MT = 16-row batch tiles per workgroup, KS = 16-wide k super-steps per wave,
// VEC = (H % 4 == 0): 16-byte exchanged-vector loads
template <int MODE, int MT, int KS, bool VEC>
__global__ __launch_bounds__(256, 1) void rnn_kernel(RnnP p) {
    using C = Cfg<MODE>;
    constexpr int G = C::G, U = C::U, NT = C::NT;
    constexpr bool FWD = (MODE == LSTM_FWD || MODE == GRU_FWD || MODE == RNN_FWD);
    constexpr bool LSTM = (MODE == LSTM_FWD || MODE == LSTM_BWD);
    constexpr bool GRU = (MODE == GRU_FWD || MODE == GRU_BWD);
    constexpr int LDR = red_ld(NT, U);
    constexpr int ROWS = MT * 16;
    constexpr int EPT = (ROWS * U + 255) / 256;  // (row, unit) elements per thread
    __shared__ float red[4][ROWS][LDR];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pslice = blockIdx.x, qg = blockIdx.y, d = blockIdx.z;
    const int T = p.T, N = p.N, H = p.H, GH = G * H;
    const int u0 = pslice * U;   // first hidden unit of this workgroup
    const int q0 = p.n0 + qg * p.rpg;  // first batch row of this group
    const int NB = min(p.n1, q0 + p.rpg);  // rows >= NB belong to another group / a later launch
    const int Kx = FWD ? H : GH; // length of the exchanged vector per row
    const int kw = KS * 16;      // k-range per wave
    const int kb = wave * kw;
    unsigned* cnt = p.sync + (d * p.Q + qg) * CNT_STRIDE;
    unsigned* err = p.sync + ERR_WORD;

    // ---- B fragments: this workgroup's slice of W_hh, resident for the whole launch ----------
    const float* W = d == 0 ? p.w_hh : p.w_hh_r;
    float bf[KS][4][NT];
    {
        const int n = lane & 15, q = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c = nt * 16 + n;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = kb + ks * 16 + q * 4 + e;
                    float v = 0.f;
                    if (FWD) {
                        // out column c -> gate c/U, unit u0 + c%U ; B[k][c] = W[gate*H+unit][k]
                        const int gate = c / U, unit = u0 + c % U;
                        if (c < G * U && unit < H && k < H) v = W[(int64_t)(gate * H + unit) * H + k];
                    } else {
                        // out column c -> unit u0 + c ; B[k][c] = W[k][unit], k over G*H gate rows
                        const int unit = u0 + c;
                        if (c < U && unit < H && k < GH) v = W[(int64_t)k * H + unit];
                    }
                    bf[ks][e][nt] = v;
                }
            }
    }

    // exchanged buffer (read with sc1 buffer loads)
    const float* xbase = FWD ? (const float*)p.hout : (const float*)p.dg1;
    const int64_t xtot = FWD ? (int64_t)2 * T * N * H : (int64_t)T * N * 2 * GH;
    auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, (int)(xtot * 4 > 0x7fffffffLL ? 0x7fffffffLL : xtot * 4), 0x00020000);

    // per-thread carried state
    float carry[EPT];   // LSTM fwd: c_{t-1}; GRU fwd: h_{t-1}; LSTM bwd: dc*f carry; GRU bwd: dh*z carry
#pragma unroll
    for (int i = 0; i < EPT; ++i) carry[i] = 0.f;

    for (int s = 0; s < T; ++s) {
        // forward direction d=0 walks t = 0..T-1; d=1 walks T-1..0.  BPTT walks the opposite way.
        const int fwd_order = (d == 0) ? s : T - 1 - s;
        const int t = FWD ? fwd_order : (T - 1 - fwd_order);
        const int tp = FWD ? (d == 0 ? t - 1 : t + 1) : (d == 0 ? t + 1 : t - 1);  // step processed before this one

        // ---- prefetch the step's private inputs (independent of the exchange) ----------------
        float pin[EPT][4];   // fwd: input projections of the G gates; bwd: [0] = dy
        float sav[EPT][6];   // bwd: saved gate activations [0..3], cell / previous-state values [4..5]
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
                    const float* pp = p.pre + (tn * 2 + d) * GH + unit;
#pragma unroll
                    for (int g = 0; g < G; ++g) pin[i][g] = pp[g * H];
                } else {
                    pin[i][0] = p.dy[tn * H + unit];
                    const f32x4 ga4 = *reinterpret_cast<const f32x4*>(p.gact + (((int64_t)d * T * N + tn) * H + unit) * 4);
                    sav[i][0] = ga4.x; sav[i][1] = ga4.y; sav[i][2] = ga4.z; sav[i][3] = ga4.w;
                    // state at the previous time step IN FORWARD ORDER: t-1 for d=0, t+1 for d=1 (zero at the start)
                    const int tq = (d == 0) ? t - 1 : t + 1;
                    const bool hasq = (tq >= 0 && tq < T);
                    const int64_t qn = ((int64_t)d * T * N + (int64_t)tq * N + gr) * H + unit;
                    if (LSTM) {
                        sav[i][4] = p.cst[((int64_t)d * T * N + tn) * H + unit];
                        sav[i][5] = hasq ? p.cst[qn] : 0.f;
                    } else if (GRU) {
                        sav[i][5] = hasq ? p.hout[qn] : 0.f;
                    }
                }
            }
        }

        // ---- recurrent product over the exchanged vector of the previous step ----------------
        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            if (!(p.flags & 4)) {
                if (tid == 0) wait_counter(cnt, (unsigned)p.P * (unsigned)s, err, (unsigned)p.tag);
                __syncthreads();
            }
            const int m = lane & 15, q = lane >> 4;
            // byte offset of this lane's first float4 of the previous step's vector, per 16-row tile; rows
            // beyond the batch get an out-of-range offset: the buffer descriptor's bounds check returns 0
            // for them, so the loads below are branch-free and can be issued back to back.
            const int64_t rbase = FWD ? ((int64_t)d * T + tp) * N * H : ((int64_t)tp * N * 2 + d) * GH;
            const int64_t rstride = FWD ? H : 2 * GH;
            constexpr unsigned OOB = 0x80000000u;
            unsigned roff[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int gr = q0 + mt * 16 + m;
                roff[mt] = (gr < NB && !(p.flags & 1)) ? (unsigned)((rbase + (int64_t)gr * rstride + kb + q * 4) * 4) : OOB;
            }
            const int klane = kb + q * 4;  // first k of this lane within super-step 0
            if (VEC) {
                // software pipeline over chunks of CH super-steps: loads of chunk c+DEPTH are in flight while the
                // MFMAs of chunk c issue (the compiler emits counted vmcnt waits for these builtin loads)
                constexpr int CH = KS >= 4 ? 4 : KS;
                constexpr int NCH = KS / CH;
                constexpr int DEPTH = NCH >= 3 ? 2 : (NCH >= 2 ? 1 : 0);
                f32x4 abuf[DEPTH + 1][CH][MT];
                auto issue = [&](int c, f32x4 (&dst)[CH][MT]) {
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const int ks = c * CH + j;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const unsigned off = (klane + ks * 16 < Kx) ? roff[mt] + (unsigned)(ks * 64) : OOB;
                            dst[j][mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)off, 0, 16));
                        }
                    }
                };
#pragma unroll
                for (int c = 0; c < DEPTH && c < NCH; ++c) issue(c, abuf[c % (DEPTH + 1)]);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c + DEPTH < NCH) issue(c + DEPTH, abuf[(c + DEPTH) % (DEPTH + 1)]);
                    if (!(p.flags & 2)) {
#pragma unroll
                        for (int j = 0; j < CH; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                    for (int nt = 0; nt < NT; ++nt)
                                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[c % (DEPTH + 1)][j][mt][e], bf[c * CH + j][e][nt], acc[mt][nt], 0, 0, 0);
                    }
                }
            } else {
                // H % 4 != 0 (tiny test shapes only, KS == 1): scalar sc1 loads
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (roff[mt] != OOB && klane < Kx) {
                        const float* xp = xbase + roff[mt] / 4;
                        v.x = __hip_atomic_load(xp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (klane + 1 < Kx) v.y = __hip_atomic_load(xp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (klane + 2 < Kx) v.z = __hip_atomic_load(xp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (klane + 3 < Kx) v.w = __hip_atomic_load(xp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[e], bf[0][e][nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
        // ---- cross-wave reduction through LDS: C/D map col = lane&15, row = (lane>>4)*4 + r ----
        {
            const int col = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[wave][mt * 16 + rq + r][nt * 16 + col] = acc[mt][nt][r];
        }
        __syncthreads();

        // ---- gate math: one (row, unit) per thread slot ---------------------------------------
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / U, u = idx % U;
            const int gr = q0 + row, unit = u0 + u;
            const bool ok = (idx < ROWS * U) && gr < NB && unit < H;
            if (!ok) continue;
            float rs[G];
            if (FWD) {
#pragma unroll
                for (int g = 0; g < G; ++g)
                    rs[g] = red[0][row][g * U + u] + red[1][row][g * U + u] + red[2][row][g * U + u] + red[3][row][g * U + u];
            } else {
                rs[0] = red[0][row][u] + red[1][row][u] + red[2][row][u] + red[3][row][u];
            }
            const int64_t tn = (int64_t)t * N + gr;
            const bool live = !FWD || t < (gr < p.cls_n ? p.cls_t0 : p.cls_t1);
            if (MODE == LSTM_FWD) {
                float ig = sigmoidf_(pin[i][0] + rs[0]);
                float fg = sigmoidf_(pin[i][1] + rs[1]);
                float gg = tanhf_(pin[i][2] + rs[2]);
                float og = sigmoidf_(pin[i][3] + rs[3]);
                float c = fg * carry[i] + ig * gg;
                float h = og * tanhf_(c);
                if (!live) { ig = fg = gg = og = c = h = 0.f; }
                carry[i] = c;
                st_sc1(p.hout + ((int64_t)d * T * N + tn) * H + unit, h);
                                *reinterpret_cast<f32x4*>(p.gact + (((int64_t)d * T * N + tn) * H + unit) * 4) = (f32x4){ig, fg, gg, og};
                p.cst[((int64_t)d * T * N + tn) * H + unit] = c;
            } else if (MODE == GRU_FWD) {
                float rg = sigmoidf_(pin[i][0] + rs[0]);
                float zg = sigmoidf_(pin[i][1] + rs[1]);
                float hn = rs[2];
                float ng = tanhf_(pin[i][2] + rg * hn);
                float h = (1.f - zg) * ng + zg * carry[i];
                if (!live) { rg = zg = hn = h = 0.f; ng = 1.f; }
                carry[i] = h;
                st_sc1(p.hout + ((int64_t)d * T * N + tn) * H + unit, h);
                                *reinterpret_cast<f32x4*>(p.gact + (((int64_t)d * T * N + tn) * H + unit) * 4) = (f32x4){rg, zg, ng, hn};
            } else if (MODE == RNN_FWD) {
                const float h = tanhf_(pin[i][0] + rs[0]);
                st_sc1(p.hout + ((int64_t)d * T * N + tn) * H + unit, h);
                *reinterpret_cast<f32x4*>(p.gact + (((int64_t)d * T * N + tn) * H + unit) * 4) = (f32x4){h, 0.f, 0.f, 0.f};
            } else if (MODE == RNN_BWD) {
                const float dh = pin[i][0] + rs[0];
                const float h = sav[i][0];
                st_sc1(p.dg1 + (tn * 2 + d) * GH + unit, dh * (1.f - h * h));
            } else if (MODE == LSTM_BWD) {
                const float dh = pin[i][0] + rs[0];
                const float ig = sav[i][0], fg = sav[i][1], gg = sav[i][2], og = sav[i][3];
                const float c = sav[i][4], cp = sav[i][5];
                const float tc = tanhf_(c);
                const float dc = dh * og * (1.f - tc * tc) + carry[i];
                carry[i] = dc * fg;
                float* dg = p.dg1 + (tn * 2 + d) * GH + unit;
                st_sc1(dg, dc * gg * ig * (1.f - ig));
                st_sc1(dg + H, dc * cp * fg * (1.f - fg));
                st_sc1(dg + 2 * H, dc * ig * (1.f - gg * gg));
                st_sc1(dg + 3 * H, dh * tc * og * (1.f - og));
            } else {  // GRU_BWD
                const float dh = pin[i][0] + rs[0] + carry[i];
                const float rg = sav[i][0], zg = sav[i][1], ng = sav[i][2], hn = sav[i][3];
                const float hp = sav[i][5];
                carry[i] = dh * zg;
                const float dnp = dh * (1.f - zg) * (1.f - ng * ng);
                const float dzp = dh * (hp - ng) * zg * (1.f - zg);
                const float drp = dnp * hn * rg * (1.f - rg);
                float* dh_ = p.dg1 + (tn * 2 + d) * GH + unit;   // d/d(W_hh h): exchanged
                st_sc1(dh_, drp);
                st_sc1(dh_ + H, dzp);
                st_sc1(dh_ + 2 * H, dnp * rg);
                float* dx_ = p.dg2 + (tn * 2 + d) * GH + unit;   // d/d(pre)
                dx_[0] = drp; dx_[H] = dzp; dx_[2 * H] = dnp;
            }
        }
        // ---- publish: drain this wave's stores, workgroup barrier, one arrival per workgroup ---
        if (!(p.flags & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0 && !(p.flags & 8)) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int MODE, int MT, int KS, bool VEC>
int launch_k(const RnnP& p, hipStream_t s) {
    dim3 grid(p.P, p.Q, 2);
    hipLaunchKernelGGL((rnn_kernel<MODE, MT, KS, VEC>), grid, dim3(256), 0, s, p);
    return 0;
}

template <int MODE, int MT>
int launch_mt(const RnnP& p, int ks_need, bool vec, hipStream_t s) {
    constexpr bool FWD = (MODE == LSTM_FWD || MODE == GRU_FWD || MODE == RNN_FWD);
    if (!vec) return ks_need <= 1 ? launch_k<MODE, MT, 1, false>(p, s) : -1;
    if (ks_need <= 1) return launch_k<MODE, MT, 1, true>(p, s);
    if (ks_need <= 2) return launch_k<MODE, MT, 2, true>(p, s);
    if (ks_need <= 4) return launch_k<MODE, MT, 4, true>(p, s);
    if (ks_need <= 8) return launch_k<MODE, MT, 8, true>(p, s);
    {
        if (ks_need <= 16) return launch_k<MODE, MT, 16, true>(p, s);      // (forward kernels: H <= 1024)
        if constexpr (FWD || MODE == RNN_BWD) return -1;
        else {
            if (ks_need <= 32) return launch_k<MODE, MT, 32, true>(p, s);
            if (ks_need <= 48) return launch_k<MODE, MT, 48, true>(p, s);
            return -1;
        }
    }
}

// rows per group: the smallest of {8, 16, 32} whose grid P x ceil(N/rpg) x 2 fits the CUs (else 32 + chunking)
inline void pick_groups(int P, int N, int cus, int& mt, int& rpg) {
    const int cand[3] = {8, 16, 32};
    rpg = 32;
    for (int i = 0; i < 3; ++i)
        if (P * cdiv(N, cand[i]) * 2 <= cus) { rpg = cand[i]; break; }
    if (N <= 8 && rpg > 8) rpg = 8;
    mt = rpg > 16 ? 2 : 1;
}

template <int MODE>
int run(const char* name, RnnP p, hipStream_t s) {
    using C = Cfg<MODE>;
    constexpr bool FWD = (MODE == LSTM_FWD || MODE == GRU_FWD || MODE == RNN_FWD);
    AAS_CHECK(p.T >= 1 && p.N >= 1 && p.H >= 1, "%s: bad sizes T=%d N=%d H=%d", name, p.T, p.N, p.H);
    const int cus = aas_rnn_cus();
    AAS_CHECK(cus > 0, "%s: no HIP device", name);
    p.flags = aas_debug_flags_value();
    p.tag = aas_rnn_launch_tag_value();
    p.P = cdiv(p.H, C::U);
    AAS_CHECK(p.P * 2 <= cus, "%s: H=%d needs %d resident workgroups, device has %d CUs", name, p.H, p.P * 2, cus);
    const int kx = FWD ? p.H : C::G * p.H;
    const int ks_need = cdiv(kx, 64);
    const bool vec = (p.H % 4 == 0);
    const int64_t xbytes = (FWD ? (int64_t)2 * p.T * p.N * p.H : (int64_t)p.T * p.N * 2 * C::G * p.H) * 4;
    AAS_CHECK(xbytes < 0x7fffffffLL, "%s: exchanged buffer of %lld bytes exceeds the 2 GiB buffer-descriptor range", name, (long long)xbytes);
    // batch rows are independent: they are split into groups of `rpg` rows, one workgroup column per group, so
    // that the persistent grid P x Q x 2 stays resident; smaller groups mean fewer exchanged bytes per workgroup
    // and step (the per-CU L2 fetch rate is what bounds a step), so take the smallest that still fits
    int mt, rpg;
    pick_groups(p.P, p.N, cus, mt, rpg);
    p.rpg = rpg;
    const int qmax = cus / (p.P * 2) < 1 ? 1 : cus / (p.P * 2);
    for (int n0 = 0; n0 < p.N; n0 += qmax * rpg) {
        p.n0 = n0;
        const int rows = (p.N - n0) < qmax * rpg ? (p.N - n0) : qmax * rpg;
        p.n1 = n0 + rows;
        p.Q = cdiv(rows, rpg);
        AAS_CHECK((p.Q * 2) * CNT_STRIDE <= SYNC_WORDS, "%s: too many batch groups", name);
        AAS_HIP(hipMemsetAsync(p.sync, 0, SYNC_WORDS * sizeof(unsigned), s));  // the ERR word after it stays sticky
        int rc = (mt == 1) ? launch_mt<MODE, 1>(p, ks_need, vec, s) : launch_mt<MODE, 2>(p, ks_need, vec, s);
        AAS_CHECK(rc == 0, "%s: hidden size H=%d not supported by the instantiated kernels", name, p.H);
        AAS_LAUNCH_CHECK(name);
    }
    return 0;
}

}  // namespace
