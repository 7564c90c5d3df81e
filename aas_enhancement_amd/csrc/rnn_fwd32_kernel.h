// Forward persistent recurrent kernel with 32-unit slices (512-thread workgroups), split-bf16 products.
//
// Same algorithm and exchange format as the forward modes of rnn_split_kernel.h (all-gather of h_t through poison-tagged
// 128-byte hi|lo lines), with twice the units per workgroup: half as many slices P, so half as many producers to poll
// and half the bytes that the chip moves per step (every workgroup still reads rows x H, but there are half as many
// workgroups per batch group - or, at a fixed CU budget, half as many rows per workgroup).  8 waves split K eight ways
// (a quarter of the fragment loads per wave of the 4-wave kernel), the W_hh slice [H x G*32] stays in VGPRs as bf16 hi/lo
// B-fragments; for the 1000-unit GRU (192 of 256 VGPRs) the lo fragments of the last LKS k-steps live in LDS.
#pragma once
#include "rnn_split_kernel.h"

namespace {

// MODE = LSTM_FWD or GRU_FWD; KS = 32-wide k chunks per wave (Hp = 256*KS); LKS = k-steps whose lo fragments are in LDS
// NRB > 0 (exact LSTM, KS = 2, row groups of <= 4 NRB <= 8 rows): the product on 4 x 4 x 1 MFMA blocks with the A broadcast, as in
// rnn_split_kernel.h - B = W[gate column 64 cg + lane][k], A = h_{t-1} (lane 4b+i of ONE 16-byte exchange load per row block:
// row i, k = 4b .. 4b+3; abid = b picks the quad): half the MFMA cycles of the 16-row tiles at 8 rows per group.
template <int MODE, int KS, int LKS, bool EX = false, int NRB = 0>
__global__ __launch_bounds__(512, 1) void rnn_fwd32_kernel(RnnP p) {
    using C = Cfg<MODE>;
    constexpr int G = C::G, U = 32, NW = 8;
    constexpr bool LSTM = (MODE == LSTM_FWD);
    constexpr int NT = G * U / 16;                       // 16-column tiles of the workgroup's G*U gate columns
    constexpr int LDR = NT * 16 + 16;
    constexpr bool DB = LSTM;                            // parity-double-buffered reduction (one barrier per step)
    constexpr bool R4 = NRB > 0;
    static_assert(!R4 || (EX && LSTM && KS == 2 && LKS == 0 && NRB <= 2), "R4: exact LSTM, 64 k per wave, all of W in registers");
    constexpr int CG = NT / 4;                           // R4: 64-column groups
    __shared__ __attribute__((aligned(16))) float red2[DB ? 2 : 1][NW][16][LDR];
    __shared__ u32x4 bl_lds[LKS ? LKS * NT : 1][LKS ? 512 : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xidx = (int)(blockIdx.x >> 3);                                     // XCD-aware launch: see rnn_split_kernel.h
    const int xset = (int)(blockIdx.x & 7) + 8 * (xidx / p.P);
    const int pslice = p.xcd ? xidx % p.P : (int)blockIdx.x;
    const int qg = p.xcd ? (xset >> 1) : (int)blockIdx.y;
    const int d = p.xcd ? (xset & 1) : (int)blockIdx.z;
    const int T = p.T, N = p.N, H = p.H, GH = G * H;
    const int Hp = p.P * U;                              // padded unit pitch of the exchange rows
    const int u0 = pslice * U;
    const int q0 = p.n0 + qg * p.rpg;
    const int NB = min(p.n1, q0 + p.rpg);
    const int kb = wave * KS * 32;
    unsigned* err = p.sync + ERR_WORD;

    // ---- B fragments (hi / lo) of this workgroup's W_hh slice: rows {gate*H + unit}, this wave's k range ------------
    const float* W = d == 0 ? p.w_hh : p.w_hh_r;
    u32x4 b0[R4 ? 1 : KS][R4 ? 1 : NT], b1[R4 ? 1 : KS][R4 ? 1 : NT];     // (rnn_split_kernel.h: frag_make - bf16 hi / lo words, or the 8 fp32 values of the exact mode)
    float wr[R4 ? CG : 1][R4 ? 64 : 1];                  // R4: W[gate column 64 cg + lane][k of this wave]
    if constexpr (R4) {
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            const int c = cg * 64 + lane;
            const int gate = c % G, unit = u0 + c / G;
            const float* wrow = W + (int64_t)(gate * H + unit) * H;
            if ((H & 3) == 0) {    // 16-byte loads of the lane's contiguous k range (rnn_split_kernel.h)
#pragma unroll
                for (int kk = 0; kk < 64; kk += 4) {
                    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (unit < H && kb + kk < H) v = *reinterpret_cast<const f32x4*>(wrow + kb + kk);
                    wr[cg][kk] = v[0]; wr[cg][kk + 1] = v[1]; wr[cg][kk + 2] = v[2]; wr[cg][kk + 3] = v[3];
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 64; ++kk) wr[cg][kk] = (unit < H && kb + kk < H) ? wrow[kb + kk] : 0.f;
            }
        }
    } else {
        const int n = lane & 15, q = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c = nt * 16 + n;                    // gate column c = u*G + gate: a unit's G gates are adjacent
                const int gate = c % G, unit = u0 + c / G;
                float wv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = kb + ks * 32 + q * 8 + e;
                    wv[e] = (unit < H && k < H) ? W[(int64_t)(gate * H + unit) * H + k] : 0.f;
                }
                u32x4 w0, w1;
                frag_make<EX>(wv, w0, w1);
                b0[ks][nt] = w0;
                if (ks >= KS - LKS) bl_lds[LKS ? (ks - (KS - LKS)) * NT + nt : 0][LKS ? tid : 0] = w1;
                else b1[ks][nt] = w1;
            }
    }

    // exchange rows: [2][T][N] rows of KC 128-byte chunks (32 units: 64 B bf16 hi | 64 B bf16 lo)
    const int KC = Hp / 32;
    unsigned* xq = p.xchg;
    // ring form of a managed buffer (exact kernels only): [2][4 time slots][N] rows, see rnn_split_kernel.h
    const bool ring = EX && p.ring;
    const int TR = ring ? 4 : T;
    auto xrow_f = [&](int dd, int tt) -> int64_t { return ring ? (int64_t)(dd * 4 + (tt & 3)) : (int64_t)dd * T + tt; };
    auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xq, 0, (int)((int64_t)2 * TR * N * KC * 128), 0x00020000);
    xchg_clean_other_half(p);
    constexpr unsigned OOB = 0x80000000u;
    constexpr unsigned POISON = 0xFFFFFFFFu;
    __shared__ unsigned long long xcd_flag;
    bool plain = false;   // plain (L2-resident) publish stores once the set is verified to share an XCD
    if (p.xcd) {
        unsigned* tab = xq + (int64_t)2 * TR * N * KC * 32;
        plain = xcd_set_colocated(tab, xset, pslice, p.P, err, p.tag, &xcd_flag, p.sync + XSTAT_WORD + 4 * (MODE == LSTM_FWD ? 0 : 2)) && !(p.flags & 524288) && p.xcd == 1;
    }

    // gate-math role: one (row, unit) per thread
    const int row = tid >> 5, u = tid & 31;
    const int gr = q0 + row, unit = u0 + u;
    const bool rowok = gr < NB;
    const bool ok = rowok && unit < H;

    float carry = 0.f;
    unsigned long long ph[5] = {0, 0, 0, 0, 0};
    const bool stamp = (p.flags & 64) && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    for (int s = 0; s < T; ++s) {
        unsigned long long st0 = stamp ? __builtin_amdgcn_s_memrealtime() : 0ull, st1 = st0, st2 = st0, st3 = st0;
        const int t = (d == 0) ? s : T - 1 - s;
        const int tp = (d == 0) ? t - 1 : t + 1;
        const int64_t tn = (int64_t)t * N + gr;

        // ---- private inputs: pre-activations of my (row, unit) --------------------------------------
        float pin[G];
#pragma unroll
        for (int g = 0; g < G; ++g) pin[g] = 0.f;
        if (ok) {
            const float* pp = p.pre + (tn * 2 + d) * GH + unit;
#pragma unroll
            for (int g = 0; g < G; ++g) pin[g] = pp[g * H];
        }

        // ---- recurrent product: this wave's k range of h_{t-1} x W slice ------------------------------
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            const int m = lane & 15, q = lane >> 4;
            const int64_t xr0 = xrow_f(d, tp) * N + q0;
            if (!(p.flags & 4)) {
                // poll one word per producer slice of my k range (first row of the group)
                const int nprod = (KS * 32) / U;
                const int kprobe = kb + lane * U;
                const bool probe = lane < nprod && kprobe < Hp && !(p.flags & 1);
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
                const int i4 = lane & 3, bq = lane >> 2;
                u32x4 hf[NRB];
                unsigned hoff[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const int grr = q0 + rb * 4 + i4;
                    hoff[rb] = (grr < NB && kb + bq * 4 < Hp && !(p.flags & 1)) ? (unsigned)(((xr0 + rb * 4 + i4) * KC) * 128 + (kb + bq * 4) * 4) : OOB;
                }
                unsigned spins = 0;
                unsigned long long t0 = 0;
                while (true) {
                    unsigned mx = 0u;
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) {
                        hf[rb] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)hoff[rb], 0, 16));
                        mx = max(mx, max(max(hf[rb].x, hf[rb].y), max(hf[rb].z, hf[rb].w)));
                    }
                    if ((p.flags & 4) || !__any(mx == POISON)) break;
                    if (spins == 0) t0 = __builtin_amdgcn_s_memrealtime();
                    if ((++spins & 63u) == 0) {
                        if (ld_cnt(err) != 0) break;
                        if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) {
                            if (lane == 0) __hip_atomic_store(err, (unsigned)p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
                // NRB x CG x 2 independent accumulation chains (row block, column group, k parity)
                f32x4 cc[NRB][CG][2];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int cg = 0; cg < CG; ++cg) cc[rb][cg][0] = cc[rb][cg][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!(p.flags & 2)) {
#define AAS_R4_STEP(B_)                                                                                                              \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) _Pragma("unroll") for (int rb = 0; rb < NRB; ++rb) _Pragma("unroll") for (int cg = 0; cg < CG; ++cg) \
        cc[rb][cg][v & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(hf[rb][v]), wr[cg][(B_) * 4 + v], cc[rb][cg][v & 1], 4, (B_), 0);
                    AAS_R4_STEP(0) AAS_R4_STEP(1) AAS_R4_STEP(2) AAS_R4_STEP(3) AAS_R4_STEP(4) AAS_R4_STEP(5) AAS_R4_STEP(6) AAS_R4_STEP(7)
                    AAS_R4_STEP(8) AAS_R4_STEP(9) AAS_R4_STEP(10) AAS_R4_STEP(11) AAS_R4_STEP(12) AAS_R4_STEP(13) AAS_R4_STEP(14) AAS_R4_STEP(15)
#undef AAS_R4_STEP
                }
                // lane = gate column within the group, register i = row 4 rb + i: parked in acc[] for the reduction write below
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int cg = 0; cg < CG; ++cg) acc[rb * CG + cg] = cc[rb][cg][0] + cc[rb][cg][1];
            } else {
            const int grm = q0 + m;
            const unsigned roff = (grm < NB && !(p.flags & 1)) ? (unsigned)(((xr0 + m) * KC + wave * KS) * 128) + frag_off0<EX>(q) : OOB;
            u32x4 ah[KS], al[KS];
            unsigned spins = 0;
            unsigned long long t0 = 0;
            bool fresh = !(p.flags & 32);   // sc1 from the first attempt (see rnn_split_kernel.h); debug flag 32: plain first
            while (true) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const unsigned off = (kb + ks * 32 + q * 8 < Hp) ? roff + (unsigned)(ks * 128) : OOB;
                    if (fresh) {
                        ah[ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 16));
                        al[ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(off + frag_off1<EX>()), 0, 16));
                    } else {
                        ah[ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 0));
                        al[ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(off + frag_off1<EX>()), 0, 0));
                    }
                }
                unsigned mx = 0u;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    mx = max(mx, max(max(ah[ks].x, ah[ks].y), max(ah[ks].z, ah[ks].w)));
                    mx = max(mx, max(max(al[ks].x, al[ks].y), max(al[ks].z, al[ks].w)));
                }
                if ((p.flags & 4) || !__any(mx == POISON)) break;
                fresh = true;
                if (spins == 0) t0 = __builtin_amdgcn_s_memrealtime();
                if ((++spins & 63u) == 0) {
                    if (ld_cnt(err) != 0) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) {
                        if (lane == 0) __hip_atomic_store(err, (unsigned)p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            if (!(p.flags & 2)) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        u32x4 w1;
                        if (ks >= KS - LKS) w1 = bl_lds[LKS ? (ks - (KS - LKS)) * NT + nt : 0][LKS ? tid : 0];
                        else w1 = b1[ks][nt];
                        acc[nt] = mma_chunk<EX>(acc[nt], ah[ks], al[ks], b0[ks][nt], w1);
                    }
                }
            }
            }
        }
        if (stamp) st2 = __builtin_amdgcn_s_memrealtime();
        float (*red)[16][LDR] = red2[DB ? (s & 1) : 0];
        if constexpr (R4) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[wave][rb * 4 + r][cg * 64 + lane] = acc[rb * CG + cg][r];
        } else {
            const int col = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][rq + r][nt * 16 + col] = acc[nt][r];
        }
        __syncthreads();
        if (stamp) st3 = __builtin_amdgcn_s_memrealtime();

        // ---- gate math ----------------------------------------------------------------------------------
        float hval = 0.f, kp[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (ok) {
            float rs[G];
#pragma unroll
            for (int g = 0; g < G; ++g) rs[g] = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                if (G == 4) {  // one 16-byte LDS read per wave partial
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&red[w][row][u * 4]);
                    rs[0] += v[0]; rs[1] += v[1]; rs[2] += v[2]; rs[G - 1] += v[3];
                } else {
#pragma unroll
                    for (int g = 0; g < G; ++g) rs[g] += red[w][row][u * G + g];
                }
            }
            if (LSTM) {
                const float ig = sigmoidf_(pin[0] + rs[0]);
                const float fg = sigmoidf_(pin[1] + rs[1]);
                const float gg = tanhf_(pin[2] + rs[2]);
                const float og = sigmoidf_(pin[G - 1] + rs[G - 1]);
                const float c = fg * carry + ig * gg;
                carry = c;
                hval = og * tanhf_(c);
                kp[0] = ig; kp[1] = fg; kp[2] = gg; kp[3] = og; kp[4] = c;
                if (t >= (gr < p.cls_n ? p.cls_t0 : p.cls_t1)) {   // dead (t, row) of a shorter row class (rnn_kernel.h: RnnP)
                    carry = 0.f; hval = 0.f;
                    kp[0] = kp[1] = kp[2] = kp[3] = kp[4] = 0.f;
                }
            } else {
                const float rg = sigmoidf_(pin[0] + rs[0]);
                const float zg = sigmoidf_(pin[1] + rs[1]);
                const float hn = rs[2];
                const float ng = tanhf_(pin[2] + rg * hn);
                hval = (1.f - zg) * ng + zg * carry;
                carry = hval;
                kp[0] = rg; kp[1] = zg; kp[2] = ng; kp[3] = hn;
                if (t >= (gr < p.cls_n ? p.cls_t0 : p.cls_t1)) {
                    carry = 0.f; hval = 0.f;
                    kp[0] = 0.f; kp[1] = 0.f; kp[2] = 1.f; kp[3] = 0.f;
                }
            }
        }
        // publish (pad units publish zeros): split - even-unit lanes store {own, partner} packed hi and lo words; exact - every lane
        // stores its own fp32 word
        if constexpr (EX) {
            if (rowok && s + 1 < T && !(p.flags & 8)) {
                const int64_t xr = xrow_f(d, t) * N + gr;
                unsigned* wq = xq + xr * KC * 32 + unit;
                if (plain) st_sc0_u32(wq, __float_as_uint(hval));
                else st_sc1_u32(wq, __float_as_uint(hval));
            }
            if (ring && rowok && s >= 2) {     // the word published two steps ago is poison again (rnn_split_kernel.h: why this is safe)
                unsigned* cq = xq + (xrow_f(d, d == 0 ? t - 2 : t + 2) * N + gr) * KC * 32 + unit;
                if (plain) st_sc0_u32(cq, 0xFFFFFFFFu);
                else st_sc1_u32(cq, 0xFFFFFFFFu);
            }
        } else {
            unsigned h0, l0;
            split_bf16(hval, h0, l0);
            const unsigned mine = h0 | (l0 << 16);
            const unsigned other = __shfl_xor(mine, 1, 64);
            if (rowok && !(u & 1) && s + 1 < T && !(p.flags & 8)) {
                const int64_t xr = ((int64_t)d * T + t) * N + gr;
                unsigned* wq = xq + xr * KC * 32 + (unit / 32) * 32 + (unit % 32) / 2;
                if (plain) {
                    st_sc0_u32(wq, (mine & 0xFFFFu) | (other << 16));
                    st_sc0_u32(wq + 16, (mine >> 16) | (other & 0xFFFF0000u));
                } else {
                    st_sc1_u32(wq, (mine & 0xFFFFu) | (other << 16));
                    st_sc1_u32(wq + 16, (mine >> 16) | (other & 0xFFFF0000u));
                }
            }
        }
        if (ok) {
            p.hout[((int64_t)d * T * N + tn) * H + unit] = hval;
            *reinterpret_cast<f32x4*>(p.gact + (((int64_t)d * T * N + tn) * H + unit) * 4) = (f32x4){kp[0], kp[1], kp[2], kp[3]};
            if (LSTM) p.cst[((int64_t)d * T * N + tn) * H + unit] = kp[4];
        }
        if (!DB) __syncthreads();
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

// 32-unit forward launches: H >= 256 (smaller layers keep the 16-unit kernel so that enough workgroups share the work),
// Hp = 32*P a multiple of... any; KS = ceil(Hp / 256) <= 4.  Returns -1 when the shape is not covered.
template <int MODE, bool EX = false>
int run_fwd32(const char* name, RnnP p, hipStream_t s) {
    constexpr bool LSTM = (MODE == LSTM_FWD);
    p.flags = aas_debug_flags_value();
    p.tag = aas_rnn_launch_tag_value();
    if (p.H < 256 || (p.flags & 512)) return -1;
    const int cus = aas_rnn_cus();
    AAS_CHECK(cus > 0, "%s: no HIP device", name);
    // measured: while 16-unit slices still get <= 8 rows per workgroup on this CU budget they are as fast or faster
    // (500-unit LSTM, N=30, whole chip: 2.87 vs 3.01 us / step); beyond that the 32-unit kernel wins clearly (N=60 on
    // 128 CUs: 3.6 vs 5.1 us / step; 1000-unit GRU on 128 CUs: 4.5 vs 7.1)
    if (cdiv(p.H, 16) * cdiv(p.N, 8) * 2 <= cus) return -1;
    p.P = cdiv(p.H, 32);
    if (p.P * 2 > cus) return -1;
    const int Hp = p.P * 32;
    const int ks = cdiv(Hp, 256);
    if (ks > 4) return -1;
    // (coverage is decided BEFORE the exchange buffer is planned: a plan commits the buffer's bookkeeping for a launch that follows)
    if (LSTM && ks > 2) return -1;      // 128 gate columns x more than 512 k do not fit the register file
    const int64_t xbytes = (int64_t)2 * p.T * p.N * (Hp / 32) * 128;
    if (xbytes >= 0x7fffffffLL) return -1;
    int rpg = 16;
    for (int cand = 4; cand < 16; cand *= 2)
        if (p.P * cdiv(p.N, cand) * 2 <= cus) { rpg = cand; break; }
    p.rpg = rpg;
    const int qmax = cus / (p.P * 2) < 1 ? 1 : cus / (p.P * 2);
    // managed buffer (exact kernels, one launch for the batch): ring of four time slots in this launch's half, no memset launch
    AasXchgPlan plan = {};
    if (EX && p.N <= qmax * rpg && p.T >= 4) aas_xchg_plan(p.xchg, (size_t)2 * 4 * p.N * (Hp / 32) * 128 + XCD_TAB_BYTES, s, &plan);
    if (plan.managed) {
        p.xchg = plan.base; p.clean_ptr = plan.clean_ptr; p.clean_words = plan.clean_words; p.ring = 1;
    } else {
        if (aas_xchg_legacy_fill(p.xchg, (size_t)xbytes + XCD_TAB_BYTES, s)) return 2;
    }
    // one launch covers the batch: afterwards the buffer holds h_t of every step but each direction's last as operand planes
    aas_note_fwd_h_planes((!EX && p.N <= qmax * rpg) ? (Hp / 32) * 128 : 0);
    for (int n0 = 0; n0 < p.N; n0 += qmax * rpg) {
        p.n0 = n0;
        const int rows = (p.N - n0) < qmax * rpg ? (p.N - n0) : qmax * rpg;
        p.n1 = n0 + rows;
        p.Q = cdiv(rows, rpg);
        // XCD-aware grid + L2-resident publish stores: only with <= 8 rows per group - in the all-gather every workgroup of a set
        // reads the WHOLE exchanged block, and 16 readers on one L2 lose against 16 readers spread over eight at 16 rows per
        // group (N=60: 3.65 -> 4.13 us / step; N=30 at 8 rows per group: 3.06 -> 2.74)
        // (more rows per group: the XCD-aware grid alone, write-through stores - same speed as the plain grid, but a set's block
        //  crosses the fabric once instead of once per XCD; debug bit 67108864: plain grid there)
        p.xcd = ((p.Q * 2) % 8 == 0 && p.P * (p.Q * 2 / 8) <= 32 && !(p.flags & 262144)) ? (rpg <= 8 ? 1 : ((p.flags & 67108864) ? 0 : 3)) : 0;
        dim3 grid(p.P, p.Q, 2), block(512);
        if (p.xcd) grid = dim3(p.P * p.Q * 2);
        if constexpr (LSTM) {
            if (ks == 1) hipLaunchKernelGGL((rnn_fwd32_kernel<MODE, 1, 0, EX>), grid, block, 0, s, p);
            else if (ks == 2) {
                // <= 8 rows per group: the 4 x 4 x 1 block form (debug bit 268435456: the 16 x 16 x 4 tiles)
                // (this kernel is only chosen when the 16-unit kernel would get more than 8 rows per workgroup, so groups of <= 4 rows do not
                //  occur in practice: two row blocks always)
                if (EX && rpg <= 8 && !(p.flags & 268435456)) hipLaunchKernelGGL((rnn_fwd32_kernel<MODE, 2, 0, EX, EX ? 2 : 0>), grid, block, 0, s, p);
                else hipLaunchKernelGGL((rnn_fwd32_kernel<MODE, 2, 0, EX>), grid, block, 0, s, p);
            }
            else return -1;   // (not reached: refused above, before the exchange buffer was planned)
        } else {
            if (ks == 1) hipLaunchKernelGGL((rnn_fwd32_kernel<MODE, 1, 0, EX>), grid, block, 0, s, p);
            else if (ks == 2) hipLaunchKernelGGL((rnn_fwd32_kernel<MODE, 2, 0, EX>), grid, block, 0, s, p);
            else if (ks == 3) hipLaunchKernelGGL((rnn_fwd32_kernel<MODE, 3, 1, EX>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((rnn_fwd32_kernel<MODE, 4, 2, EX>), grid, block, 0, s, p);
        }
        AAS_LAUNCH_CHECK(name);
    }
    return 0;
}

template <int MODE>
int run_fwd_any(const char* name, RnnP p, hipStream_t s) {
    aas_note_fwd_h_planes(0);
    if (p.xchg && aas_precision_value() == 1) {
        const int rc = run_fwd32<MODE, false>(name, p, s);
        if (rc >= 0) return rc;
    } else if (p.xchg && !(aas_debug_flags_value() & 134217728)) {   // exact fp32 on the same data-is-the-flag kernels
        const int rc = run_fwd32<MODE, true>(name, p, s);
        if (rc >= 0) return rc;
    }
    return run_any<MODE>(name, p, s);
}

}  // namespace
