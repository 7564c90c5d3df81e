// Greedy CTC decoding on the device + host-side edit distance (validation pass of the trainers).
// Reference: AM_training/decoder.py:146-201 GreedyDecoder.decode (argmax over the alphabet, collapse repeats, drop
// blanks) called from Speech_enhancement_by_AAS/trainer_AAS.py:321-323 and AM_training/train.py:377-381; the
// `Levenshtein` C extension behind Decoder.wer / Decoder.cer (decoder.py:45-74).
#include <vector>

#include "common.h"

namespace {

// One wavefront per utterance, 64 frames per pass: lane = frame.  argmax over the C scores of the frame (first
// maximum, like torch.max), keep a frame when its label is not blank and differs from the PREVIOUS FRAME's label
// (decoder.py:176: the comparison is with sequence[i-1], blanks included), compact with a ballot prefix count.
__global__ __launch_bounds__(64) void greedy_decode_kernel(const float* __restrict__ probs, const int* __restrict__ sizes, int T,
                                                           int N, int C, int blank, int* __restrict__ out,
                                                           int* __restrict__ offs, int* __restrict__ out_len) {
    const int n = blockIdx.x, lane = threadIdx.x;
    int len = sizes[n];
    if (len > T) len = T;
    int count = 0, prev_last = -1;
    for (int base = 0; base < len; base += 64) {
        const int t = base + lane;
        int lab = -1;
        if (t < len) {
            const float* row = probs + ((int64_t)t * N + n) * C;
            float best = row[0];
            lab = 0;
            for (int c = 1; c < C; ++c) {
                const float v = row[c];
                if (v > best) { best = v; lab = c; }
            }
        }
        int prev = __shfl_up(lab, 1, 64);
        if (lane == 0) prev = prev_last;
        const bool keep = t < len && lab != blank && !(t > 0 && lab == prev);
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
            out[(int64_t)n * T + pos] = lab;
            if (offs) offs[(int64_t)n * T + pos] = t;
        }
        count += __popcll(m);
        prev_last = __shfl(lab, 63, 64);
    }
    if (lane == 0) out_len[n] = count;
}

}  // namespace

extern "C" int aas_greedy_decode(aasStream_t stream, const float* probs, const int* d_sizes, int T, int N, int C, int blank,
                                 int* out_labels, int* out_offsets, int* out_lens) {
    AAS_CHECK(probs && d_sizes && out_labels && out_lens && T > 0 && N > 0 && C > 0 && blank >= 0 && blank < C,
              "aas_greedy_decode: bad arguments");
    hipLaunchKernelGGL(greedy_decode_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, probs, d_sizes, T, N, C, blank, out_labels,
                       out_offsets, out_lens);
    AAS_LAUNCH_CHECK("aas_greedy_decode");
    return 0;
}

// Host function: Levenshtein distance between two integer sequences (two-row dynamic programme).
extern "C" int aas_edit_distance(const int* h_a, int na, const int* h_b, int nb) {
    if (na < 0 || nb < 0 || (na > 0 && !h_a) || (nb > 0 && !h_b)) return -1;
    if (na < nb) { const int* tp = h_a; h_a = h_b; h_b = tp; const int ti = na; na = nb; nb = ti; }
    std::vector<int> prev(nb + 1), cur(nb + 1);
    for (int j = 0; j <= nb; ++j) prev[j] = j;
    for (int i = 1; i <= na; ++i) {
        cur[0] = i;
        const int ca = h_a[i - 1];
        for (int j = 1; j <= nb; ++j) {
            const int sub = prev[j - 1] + (ca != h_b[j - 1]);
            const int del = prev[j] + 1, ins = cur[j - 1] + 1;
            cur[j] = sub < del ? (sub < ins ? sub : ins) : (del < ins ? del : ins);
        }
        prev.swap(cur);
    }
    return prev[nb];
}
