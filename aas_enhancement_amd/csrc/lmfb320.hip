// log-Mel filterbank features on the matrix cores, specialised for the reference's audio configuration (16 kHz, 20 ms
// hamming window = 320 samples = n_fft, 10 ms hop = 160; AM_training/train.py:39-42,199) - the `[frames, 320] x [320, 322]`
// contraction of SURVEY 7 step 8, with the real-input symmetry of the DFT folded out first.
//
// For a windowed frame x'[0..320) the 161 bins split by the parity of k after two radix-2 style folds (all exact adds):
//     e[j] = x'[j] + x'[320-j], o[j] = x'[j] - x'[320-j]      (1 <= j <= 159; e[0] = x'[0], e[160] = x'[160])
//     ee[j] = e[j] + e[160-j] (0<=j<=80, ee[80] = e[80])   eo[j] = e[j] - e[160-j] (0<=j<=79)
//     oe[j] = o[j] - o[160-j] (1<=j<=79)                   oo[j] = o[j] + o[160-j] (1<=j<=80, oo[80] = o[80])
//     (all four operands are stored at k = j; table rows outside a segment's j range are zero)
//     Re X[2c]   = sum_j ee[j] cos(2 pi j 2c / 320)         Re X[2c+1] = sum_j eo[j] cos(2 pi j (2c+1) / 320)
//     Im X[2c]   = -sum_j oe[j] sin(2 pi j 2c / 320)        Im X[2c+1] = -sum_j oo[j] sin(2 pi j (2c+1) / 320)
// i.e. four products of [frames, <=81] x [<=81, <=81] instead of one [frames, 320] x [320, 322]: 4x fewer MFMA flops.
// The hamming window is applied to the folded samples (w[320-j] = w[j] and w[160+j] = w[160-j], so each first-fold pair shares
// one weight; it is not symmetric under j -> 160-j, so it cannot live in the tables).
//
// One persistent workgroup per CU (256 threads).  Each wave keeps ITS column groups' table fragments (bf16 hi | lo, the
// split-bf16 product hi*hi + lo*hi + hi*lo used everywhere in this library) in REGISTERS for the whole launch - 144 VGPRs
// - so the tables are read once per wave and the loop touches LDS only for the folded frame operands.  Per tile of 16
// frames: stage 2 720 samples (reflect padding at the utterance's own ends) -> window, fold, split into A planes in LDS
// -> 54 MFMAs per wave -> |X|^2 to LDS -> sparse mel filters on the VALU (each bin feeds <= 2 filters) -> log1p -> store.
// Algorithmic HBM bytes: 4 S + 4 n_mels T per utterance (191 KB at S = 31 840, 80 mels).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WIN = 320, HOP = 160, TF = 16;                 // frames per tile = one MFMA M tile
constexpr int SEGK = 96;                                      // k extent of every folded operand (<= 81, zero padded)
constexpr int A_STRIDE = 4 * SEGK * 2 + 16;                   // bytes per frame row of one A plane (+16: bank spread)
constexpr int NSAMP = (TF - 1) * HOP + WIN;                   // 2 720 samples per tile
constexpr int P_STRIDE = 165;                                 // floats per frame row of the power spectrum (odd: no conflicts)
constexpr int MAXW = 24;                                      // max bins per mel filter handled by the sparse form

struct LP {
    const float* wave;       // [N, S]
    const int* lens;         // [N] valid samples per utterance, or null (all S)
    const unsigned short* tab;   // [2 planes][4 segments][96 columns][96 k] bf16 bits
    const float* win;        // [320] hamming
    const int* mel_start;    // [n_mels]
    const int* mel_cnt;      // [n_mels]
    const float* mel_w;      // [n_mels][MAXW]
    float* out;              // [N, n_mels, T]
    int N, S, T, n_mels, tiles_per_utt;
    int flags;               // ablation (aas_set_debug_flags): 1 no staging loads, 2 no fold, 4 no MFMA, 8 no mel, 16 no output stores (timing only)
};

__device__ __forceinline__ void split2(float x, unsigned short& h, unsigned short& l) {
    const __bf16 hb = (__bf16)x;
    h = __builtin_bit_cast(unsigned short, hb);
    const __bf16 lb = (__bf16)(x - __uint_as_float((unsigned)h << 16));
    l = __builtin_bit_cast(unsigned short, lb);
}

// log1p(x) for x >= 0 on the hardware log: u = 1 + x rounds, log(u) * x / (u - 1) undoes that rounding (the classic correction), so
// tiny mel energies keep their relative accuracy; ~8 instructions instead of the ~35 of ocml's log1pf.  Debug bit 32: ocml log1pf.
__device__ __forceinline__ float log1p_fast(float x) {
    const float u = 1.0f + x;
    const float d = u - 1.0f;
    return d == 0.0f ? x : __logf(u) * (x * __frcp_rn(d));
}

__global__ __launch_bounds__(256, 2) void lmfb320_kernel(LP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* a_hi = smem;                                        // [16][A_STRIDE]
    char* a_lo = a_hi + TF * A_STRIDE;
    float* seg = reinterpret_cast<float*>(a_lo + TF * A_STRIDE);   // [NSAMP] windowless samples of the tile
    float* P = seg + NSAMP;                                   // [16][P_STRIDE] power spectrum
    float* wl = P + TF * P_STRIDE;                            // [320] window
    float* mw = wl + WIN;                                     // [n_mels][MAXW]
    int* ms = reinterpret_cast<int*>(mw + p.n_mels * MAXW);   // [n_mels] start, [n_mels] count
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool odd = wave >= 2;                               // waves 0,1: even bins (k = 2c); waves 2,3: odd bins
    const int g0 = (wave & 1) * 3;                            // first 16-column group of this wave
    const int ng = (wave == 3) ? 2 : 3;                       // even: 6 groups (81 columns), odd: 5 groups (80 columns)
    const int seg_re = odd ? 1 : 0, seg_im = odd ? 3 : 2;
    const int ncols = odd ? 80 : 81;

    // ---- one-time: constants to LDS, zero the A planes (the k pads stay zero), table fragments to registers
    for (int i = tid; i < WIN; i += 256) wl[i] = p.win[i];
    for (int i = tid; i < p.n_mels * MAXW; i += 256) mw[i] = p.mel_w[i];
    for (int i = tid; i < p.n_mels; i += 256) { ms[i] = p.mel_start[i]; ms[p.n_mels + i] = p.mel_cnt[i]; }
    for (int i = tid; i < 2 * TF * A_STRIDE / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;
    bf16x8 bre[3][3][2], bim[3][3][2];                        // [group][k-step][hi, lo]
    {
        const int col = lane & 15, kc = (lane >> 4) * 8;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = (g0 + g) * 16 + col;
                    const size_t o_re = (((size_t)h * 4 + seg_re) * 96 + c) * 96 + ks * 32 + kc;
                    const size_t o_im = (((size_t)h * 4 + seg_im) * 96 + c) * 96 + ks * 32 + kc;
                    if (g < ng) {
                        bre[g][ks][h] = *reinterpret_cast<const bf16x8*>(p.tab + o_re);
                        bim[g][ks][h] = *reinterpret_cast<const bf16x8*>(p.tab + o_im);
                    } else {
                        bre[g][ks][h] = bre[0][0][0];
                        bim[g][ks][h] = bre[0][0][0];
                    }
                }
    }
    __syncthreads();

    // fold role of this thread (constant over the launch): j pair and first frame, window weights in registers
    const int fold_fg = tid / 41, fj0 = 2 * (tid - fold_fg * 41);
    const bool fold_on = fold_fg < 6;
    const float fw[4] = {wl[fj0], wl[160 - fj0], wl[fj0 + 1], wl[159 - fj0]};
    const int total = p.N * p.tiles_per_utt;
    // Software pipeline: the samples of tile i+1 are fetched into registers (3 x 16 bytes per thread) while tile i is folded,
    // multiplied and written; they go to LDS at the top of the next iteration.  Interior tiles (no reflection, no batch
    // padding inside) take aligned 16-byte loads; the first / last tiles of an utterance take the scalar reflecting path.
    constexpr int NV = (NSAMP / 4 + 255) / 256;               // float4 loads per thread (680 vectors per tile)
    f32x4 pre[NV];
    bool pre_vec = false;
    auto tile_coords = [&](int tile, int& n, int& t0, int& Sn) {
        n = tile / p.tiles_per_utt;
        t0 = (tile - n * p.tiles_per_utt) * TF;
        Sn = p.lens ? p.lens[n] : p.S;
        if (Sn > p.S) Sn = p.S;
    };
    auto prefetch = [&](int tile) {
        pre_vec = false;
        if (tile >= total || (p.flags & 1)) return;
        int n, t0, Sn;
        tile_coords(tile, n, t0, Sn);
        const int first = t0 * HOP - HOP;                     // wave index of seg[0]
        if (first >= 0 && first + NSAMP <= Sn && ((p.S & 3) == 0)) {
            const f32x4* src = reinterpret_cast<const f32x4*>(p.wave + (int64_t)n * p.S + first);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int idx = tid + v * 256;
                if (idx < NSAMP / 4) pre[v] = src[idx];
            }
            pre_vec = true;
        }
    };
    // each workgroup takes a CONTIGUOUS run of tiles: consecutive tiles of an utterance write the two 64-byte halves of the
    // same output lines and read adjacent (overlapping) sample ranges - close in time on one CU, they merge in its L2
    const int per = (total + gridDim.x - 1) / gridDim.x;
    const int tile_end = min(total, ((int)blockIdx.x + 1) * per);
    prefetch(blockIdx.x * per < tile_end ? blockIdx.x * per : total);
    for (int tile = blockIdx.x * per; tile < tile_end; ++tile) {
        int n, t0, Sn;
        tile_coords(tile, n, t0, Sn);
        const int Tn = 1 + Sn / HOP;                          // this utterance's frames; the rest of [0, T) is zero
        const int nf = min(TF, p.T - t0);
        if (t0 >= Tn || Sn <= HOP) {                          // whole tile in the batch padding
            for (int i = tid; i < p.n_mels * nf; i += 256) {
                const int m = i / nf, f = i - m * nf;
                p.out[((int64_t)n * p.n_mels + m) * p.T + t0 + f] = 0.f;
            }
            prefetch(tile + 1 < tile_end ? tile + 1 : total);
            continue;
        }
        // ---- stage: samples t0*160 - 160 .. of the centre-padded utterance (reflect at both of ITS ends)
        if (pre_vec) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int idx = tid + v * 256;
                if (idx < NSAMP / 4) reinterpret_cast<f32x4*>(seg)[idx] = pre[v];
            }
        } else if (!(p.flags & 1)) {
            const float* w = p.wave + (int64_t)n * p.S;
            for (int i = tid; i < NSAMP; i += 256) {
                int j = t0 * HOP + i - HOP;
                if (j < 0) j = -j;
                if (j >= Sn) j = 2 * (Sn - 1) - j;
                seg[i] = (j >= 0 && j < Sn) ? w[j] : 0.f;
            }
        }
        prefetch(tile + 1 < tile_end ? tile + 1 : total);
        __syncthreads();
        // ---- window + two folds + bf16 hi/lo split -> A planes [frame][segment][k = j]
        // thread -> (pair of neighbouring j: j0 = 2 jp, j1 = j0 + 1; frames fg, fg+6, fg+12): the window weights sit in
        // registers, the four segments are written as 4-byte hi / lo pairs.  Values at k positions whose table row is zero
        // (k = 0 of the sine segments, k = 80 of segments 1 and 2, k = 81) are harmless and left as computed.
        if (!(p.flags & 2) && fold_on) {
            for (int f = fold_fg; f < TF; f += 6) {
                const float* x = seg + f * HOP;
                // the periodic hamming window is symmetric in both folds' pairs: w[320-j] = w[j], w[160+j] = w[160-j]
                const float a0 = x[fj0], c0 = x[160 - fj0];
                const float b0 = fj0 >= 1 ? x[320 - fj0] : 0.f, d0 = fj0 >= 1 ? x[160 + fj0] : 0.f;
                const float a1 = x[fj0 + 1], b1 = x[319 - fj0], c1 = x[159 - fj0], d1 = x[161 + fj0];
                float v0[4], v1[4];
                {
                    const float e1 = (a0 + b0) * fw[0], e2 = (c0 + d0) * fw[1], o1 = (a0 - b0) * fw[0], o2 = (c0 - d0) * fw[1];
                    const bool mid = fj0 == 80;                 // the pair (80, 240) folds onto itself
                    v0[0] = mid ? e1 : e1 + e2; v0[1] = e1 - e2; v0[2] = o1 - o2; v0[3] = mid ? o1 : o1 + o2;
                }
                {
                    const float e1 = (a1 + b1) * fw[2], e2 = (c1 + d1) * fw[3], o1 = (a1 - b1) * fw[2], o2 = (c1 - d1) * fw[3];
                    v1[0] = e1 + e2; v1[1] = e1 - e2; v1[2] = o1 - o2; v1[3] = o1 + o2;
                }
                char* rh = a_hi + f * A_STRIDE + fj0 * 2;
                char* rl = a_lo + f * A_STRIDE + fj0 * 2;
#pragma unroll
                for (int sgm = 0; sgm < 4; ++sgm) {
                    unsigned short h0, l0, h1, l1;
                    split2(v0[sgm], h0, l0);
                    split2(v1[sgm], h1, l1);
                    *reinterpret_cast<unsigned*>(rh + sgm * SEGK * 2) = (unsigned)h0 | ((unsigned)h1 << 16);
                    *reinterpret_cast<unsigned*>(rl + sgm * SEGK * 2) = (unsigned)l0 | ((unsigned)l1 << 16);
                }
            }
        }
        __syncthreads();
        // ---- MFMA: C[frame][column] for this wave's column groups, real and imaginary part in the same lane
        if (!(p.flags & 4)) {
            const int fr = lane & 15, kc = (lane >> 4) * 16;  // A fragment: frame row, 16-byte k chunk
            // k-step outermost: the four A fragments of ONE k-step are live at a time (16 VGPRs instead of 48 for all three - the
            // kernel spilled 8 VGPRs before); every accumulator sees its products in the same order as before (ks ascending; hi x lo,
            // lo x hi, hi x hi), so the results are bit-identical
            f32x4 re0 = {0.f, 0.f, 0.f, 0.f}, re1 = re0, re2 = re0, im0 = re0, im1 = re0, im2 = re0;
#define AAS_LMFB_STEP(G, RE, IM)                                                             \
    RE = __builtin_amdgcn_mfma_f32_16x16x32_bf16(are0, bre[G][ks][1], RE, 0, 0, 0);          \
    RE = __builtin_amdgcn_mfma_f32_16x16x32_bf16(are1, bre[G][ks][0], RE, 0, 0, 0);          \
    RE = __builtin_amdgcn_mfma_f32_16x16x32_bf16(are0, bre[G][ks][0], RE, 0, 0, 0);          \
    IM = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aim0, bim[G][ks][1], IM, 0, 0, 0);          \
    IM = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aim1, bim[G][ks][0], IM, 0, 0, 0);          \
    IM = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aim0, bim[G][ks][0], IM, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const bf16x8 are0 = *reinterpret_cast<const bf16x8*>(a_hi + fr * A_STRIDE + (seg_re * SEGK + ks * 32) * 2 + kc);
                const bf16x8 are1 = *reinterpret_cast<const bf16x8*>(a_lo + fr * A_STRIDE + (seg_re * SEGK + ks * 32) * 2 + kc);
                const bf16x8 aim0 = *reinterpret_cast<const bf16x8*>(a_hi + fr * A_STRIDE + (seg_im * SEGK + ks * 32) * 2 + kc);
                const bf16x8 aim1 = *reinterpret_cast<const bf16x8*>(a_lo + fr * A_STRIDE + (seg_im * SEGK + ks * 32) * 2 + kc);
                AAS_LMFB_STEP(0, re0, im0)
                AAS_LMFB_STEP(1, re1, im1)
                if (ng > 2) { AAS_LMFB_STEP(2, re2, im2) }
            }
#undef AAS_LMFB_STEP
            // lane holds column c = (g0+g)*16 + lane%16 of frames 4*(lane/16) .. +3
            auto emit = [&](int g, const f32x4& re, const f32x4& im) {
                const int c = (g0 + g) * 16 + (lane & 15);
                if (c < ncols) {
                    const int k = odd ? 2 * c + 1 : 2 * c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) P[((lane >> 4) * 4 + r) * P_STRIDE + k] = re[r] * re[r] + im[r] * im[r];
                }
            };
            emit(0, re0, im0);
            emit(1, re1, im1);
            if (ng > 2) emit(2, re2, im2);
        }
        __syncthreads();
        // ---- mel (sparse triangular filters) + log1p + store: thread -> (frame = tid % 16, mel = tid / 16 + 16 i)
        {
            const int f = tid & 15;
            const bool live = f < nf && (t0 + f) < Tn;
            for (int m = tid >> 4; m < p.n_mels; m += 16) {
                float acc = 0.f;
                if (live && !(p.flags & 8)) {
                    const int s0 = ms[m], cnt = ms[p.n_mels + m];
                    const float* pr = P + f * P_STRIDE + s0;
                    const float* wr = mw + m * MAXW;
                    for (int q = 0; q < cnt; ++q) acc = fmaf(pr[q], wr[q], acc);
                }
                if (f < nf && !(p.flags & 16))
                    p.out[((int64_t)n * p.n_mels + m) * p.T + t0 + f] = live ? ((p.flags & 32) ? log1pf(acc) : log1p_fast(acc)) : 0.f;
            }
        }
        // (the next tile's staging writes `seg`, its fold writes the A planes: both were last read before the barrier above;
        //  P is rewritten only after the next two barriers)
    }
}

}  // namespace

extern "C" int aas_lmfb320_fwd(aasStream_t stream, const float* wave, const int* d_lens, int N, int S, int n_mels,
                               const void* tables, const float* window, const int* mel_start, const int* mel_cnt,
                               const float* mel_w, int mel_maxw, float* out) {
    AAS_CHECK(wave && tables && window && mel_start && mel_cnt && mel_w && out, "aas_lmfb320_fwd: null pointer");
    AAS_CHECK(N > 0 && S > HOP && n_mels > 0 && n_mels <= 256, "aas_lmfb320_fwd: bad sizes (S must exceed 160 samples for reflect padding)");
    AAS_CHECK(mel_maxw == MAXW, "aas_lmfb320_fwd: mel weight rows must be %d wide", MAXW);
    AAS_CHECK((reinterpret_cast<uintptr_t>(tables) & 15) == 0, "aas_lmfb320_fwd: tables must be 16-byte aligned");
    LP p;
    p.wave = wave; p.lens = d_lens; p.tab = (const unsigned short*)tables; p.win = window; p.mel_start = mel_start; p.mel_cnt = mel_cnt;
    p.flags = aas_debug_flags_value();
    p.mel_w = mel_w; p.out = out; p.N = N; p.S = S; p.T = 1 + S / HOP; p.n_mels = n_mels; p.tiles_per_utt = cdiv(p.T, TF);
    const size_t lds = 2 * (size_t)TF * A_STRIDE + sizeof(float) * (NSAMP + TF * P_STRIDE + WIN + (size_t)n_mels * MAXW) + sizeof(int) * 2 * n_mels;
    static unsigned char attr_done[AAS_MAX_DEV];
    if (aas_raise_dynamic_lds_once(attr_done, reinterpret_cast<const void*>(&lmfb320_kernel), 96 * 1024)) return 2;
    AAS_CHECK(lds <= 96 * 1024, "aas_lmfb320_fwd: LDS budget exceeded");
    const int total = N * p.tiles_per_utt;
    const int cus = aas_device_cus();
    const int grid = total < 2 * cus ? total : 2 * cus;       // persistent: ~2 workgroups per CU (LDS 50 KB, 1 wave / SIMD each)
    hipLaunchKernelGGL(lmfb320_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    AAS_LAUNCH_CHECK("aas_lmfb320_fwd");
    return 0;
}
