// log-Mel filterbank features: frame (centre=True, reflect pad) -> hamming window -> 320-point real
// DFT as a [frames x win] x [win x 2*nbins] contraction (window folded into the table) -> power ->
// mel (Slaney, area-normalised) -> log1p.  One workgroup per (utterance, 16 frames); frames and
// power spectra are staged in LDS; table reads are coalesced across bins.
// Reference conventions: AM_training/train.py:39-42,55-60,199; Speech_enhancement_by_AAS/model.py:194-198
// (the extractor source itself is absent from the reference - SURVEY.md 0.10).
#include "common.h"

namespace {
constexpr int TF = 16;  // frames per workgroup

__global__ __launch_bounds__(256) void lmfb_kernel(const float* __restrict__ wave, int S, int win, int hop, int nbins,
                                                   int n_mels, int T, const float* __restrict__ dft,
                                                   const float* __restrict__ melT, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* seg = reinterpret_cast<float*>(smem);              // [(TF-1)*hop + win] samples
    float* pw = seg + ((TF - 1) * hop + win);                 // [TF][nbins]
    const int n = blockIdx.y, t0 = blockIdx.x * TF, tid = threadIdx.x;
    const int nf = (T - t0) < TF ? (T - t0) : TF;
    const int seglen = (nf - 1) * hop + win;
    const float* w = wave + (int64_t)n * S;
    const int pad = win / 2;
    for (int i = tid; i < seglen; i += 256) {
        int j = t0 * hop + i - pad;  // index into the un-padded wave, reflect at both ends
        if (j < 0) j = -j;
        if (j >= S) j = 2 * (S - 1) - j;
        seg[i] = (j >= 0 && j < S) ? w[j] : 0.f;
    }
    __syncthreads();
    if (tid < nbins) {
        float re[TF], im[TF];
#pragma unroll
        for (int f = 0; f < TF; ++f) re[f] = im[f] = 0.f;
        for (int j = 0; j < win; ++j) {
            const float c = dft[(int64_t)j * 2 * nbins + tid];
            const float s = dft[(int64_t)j * 2 * nbins + nbins + tid];
#pragma unroll
            for (int f = 0; f < TF; ++f) {
                const float x = (f < nf) ? seg[f * hop + j] : 0.f;
                re[f] = fmaf(x, c, re[f]);
                im[f] = fmaf(x, s, im[f]);
            }
        }
#pragma unroll
        for (int f = 0; f < TF; ++f) pw[f * nbins + tid] = re[f] * re[f] + im[f] * im[f];
    }
    __syncthreads();
    for (int i = tid; i < n_mels * TF; i += 256) {
        const int m = i / TF, f = i % TF;
        if (f >= nf) continue;
        float acc = 0.f;
        for (int b = 0; b < nbins; ++b) acc = fmaf(pw[f * nbins + b], melT[(int64_t)b * n_mels + m], acc);
        out[((int64_t)n * n_mels + m) * T + t0 + f] = log1pf(acc);
    }
}
}  // namespace

extern "C" int aas_lmfb_fwd(aasStream_t stream, const float* wave, int N, int S, int win, int hop, int n_mels,
                            const float* dft, const float* melT, float* out) {
    AAS_CHECK(wave && dft && melT && out, "aas_lmfb_fwd: null pointer");
    AAS_CHECK(N > 0 && S > win / 2 && win > 0 && hop > 0 && n_mels > 0, "aas_lmfb_fwd: bad sizes (S must exceed win/2 for reflect padding)");
    const int nbins = win / 2 + 1;
    AAS_CHECK(nbins <= 256, "aas_lmfb_fwd: win=%d too long (nbins > 256)", win);
    const int T = 1 + S / hop;
    const size_t lds = sizeof(float) * ((size_t)(TF - 1) * hop + win + (size_t)TF * nbins);
    AAS_CHECK(lds <= 64 * 1024, "aas_lmfb_fwd: LDS budget exceeded");
    dim3 grid(cdiv(T, TF), N);
    hipLaunchKernelGGL(lmfb_kernel, grid, dim3(256), lds, (hipStream_t)stream, wave, S, win, hop, nbins, n_mels, T, dft, melT, out);
    AAS_LAUNCH_CHECK("aas_lmfb_fwd");
    return 0;
}
