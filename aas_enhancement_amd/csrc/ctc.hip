// CTC loss + gradient wrt pre-softmax activations (replaces warpctc_pytorch.CTCLoss, a third-party
// C++/CUDA op the reference calls at Speech_enhancement_by_AAS/trainer_AAS.py:168,349 and
// AM_training/train.py:319).  One workgroup per utterance; threads span the S = 2L+1 states of the
// blank-extended label sequence; alpha is kept in global scratch for the beta/gradient sweep;
// per-class occupancies are accumulated in LDS in linear space (normalised by the utterance
// likelihood, so they are <= 1).  Latency-bound: T' sequential steps per sweep.
#include <vector>

#include "common.h"

namespace {

// alpha/beta recursions run in fp64 (the op is latency-bound, not throughput-bound): keeps the
// gradient at fp32 round-off instead of the ~1e-4 an fp32 log-space sweep over 85+ frames gives.
typedef double real;
#define NEG_INF (-(real)INFINITY)
__device__ __forceinline__ real lse2(real a, real b) {
    if (a == NEG_INF) return b;
    if (b == NEG_INF) return a;
    const real m = fmax(a, b);
    return m + log1p(exp(-fabs(a - b)));
}

__global__ __launch_bounds__(256) void ctc_kernel(const float* __restrict__ acts, float* __restrict__ grads,
                                                  const int* __restrict__ labels, const int* __restrict__ lab_off,
                                                  const int* __restrict__ lab_lens, const int* __restrict__ act_lens,
                                                  int C, int N, int Tmax, int Smax, float* __restrict__ costs,
                                                  real* __restrict__ ws, int blank, float gscale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    real* a0 = reinterpret_cast<real*>(smem);
    real* a1 = a0 + Smax;
    float* occ = reinterpret_cast<float*>(a1 + Smax);
    int* ext = reinterpret_cast<int*>(occ + C);
    const int n = blockIdx.x, tid = threadIdx.x;
    const int L = lab_lens[n];
    int Tn = act_lens[n];
    if (Tn > Tmax) Tn = Tmax;
    const int S = 2 * L + 1;
    real* alpha = ws + (int64_t)n * ((int64_t)Tmax * Smax + Tmax);
    real* lse = alpha + (int64_t)Tmax * Smax;
    const int off = lab_off[n];
    auto act = [&](int t, int c) { return (real)acts[((int64_t)t * N + n) * C + c]; };

    for (int s = tid; s < S; s += 256) ext[s] = (s & 1) ? labels[off + (s >> 1)] : blank;
    for (int t = tid; t < Tn; t += 256) {
        real m = NEG_INF;
        for (int c = 0; c < C; ++c) m = fmax(m, act(t, c));
        real sum = 0;
        for (int c = 0; c < C; ++c) sum += exp(act(t, c) - m);
        lse[t] = m + log(sum);
    }
    __syncthreads();
    bool feasible = (Tn >= 1) && (S <= Smax);
    real ll = NEG_INF;
    if (feasible) {
        // ---- alpha sweep ----
        for (int s = tid; s < S; s += 256) {
            real a = NEG_INF;
            if (s < 2) a = act(0, ext[s]) - lse[0];
            a0[s] = a;
            alpha[s] = a;
        }
        __syncthreads();
        for (int t = 1; t < Tn; ++t) {
            const real* prev = (t & 1) ? a0 : a1;
            real* cur = (t & 1) ? a1 : a0;
            const real l_t = lse[t];
            for (int s = tid; s < S; s += 256) {
                real x = prev[s];
                if (s >= 1) x = lse2(x, prev[s - 1]);
                const int e = ext[s];
                if (s >= 2 && e != blank && e != ext[s - 2]) x = lse2(x, prev[s - 2]);
                if (x != NEG_INF) x += act(t, e) - l_t;
                cur[s] = x;
                alpha[(int64_t)t * Smax + s] = x;
            }
            __syncthreads();
        }
        const real* last = alpha + (int64_t)(Tn - 1) * Smax;
        ll = last[S - 1];
        if (S > 1) ll = lse2(ll, last[S - 2]);
        __syncthreads();
        feasible = (ll != NEG_INF);
    }
    if (tid == 0) costs[n] = feasible ? (float)(-ll) : INFINITY;
    if (!grads) return;
    if (!feasible) {
        for (int i = tid; i < Tmax * C; i += 256) grads[((int64_t)(i / C) * N + n) * C + (i % C)] = 0.f;
        return;
    }
    // ---- beta sweep fused with the gradient ----
    {
        const int t = Tn - 1;
        for (int s = tid; s < S; s += 256) a0[s] = (s >= S - 2) ? act(t, ext[s]) - lse[t] : NEG_INF;
    }
    __syncthreads();
    int flip = 0;
    for (int t = Tn - 1; t >= 0; --t) {
        const real* bcur = flip ? a1 : a0;
        real* bnext = flip ? a0 : a1;
        for (int k = tid; k < C; k += 256) occ[k] = 0.f;
        __syncthreads();
        const real l_t = lse[t];
        for (int s = tid; s < S; s += 256) {
            const real al = alpha[(int64_t)t * Smax + s], be = bcur[s];
            if (al != NEG_INF && be != NEG_INF) {
                const int e = ext[s];
                atomicAdd(&occ[e], (float)exp(al + be - (act(t, e) - l_t) - ll));
            }
        }
        if (t > 0) {
            const real l_p = lse[t - 1];
            for (int s = tid; s < S; s += 256) {
                real x = bcur[s];
                if (s + 1 < S) x = lse2(x, bcur[s + 1]);
                const int e = ext[s];
                if (s + 2 < S && ext[s + 2] != blank && ext[s + 2] != e) x = lse2(x, bcur[s + 2]);
                if (x != NEG_INF) x += act(t - 1, e) - l_p;
                bnext[s] = x;
            }
        }
        __syncthreads();
        for (int k = tid; k < C; k += 256)
            grads[((int64_t)t * N + n) * C + k] = gscale * ((float)exp(act(t, k) - l_t) - occ[k]);
        __syncthreads();
        flip ^= 1;
    }
    for (int i = tid; i < (Tmax - Tn) * C; i += 256) grads[((int64_t)(Tn + i / C) * N + n) * C + (i % C)] = 0.f;
}


// ---------------------------------------------------------------------------------------------------------------
// Single-wavefront CTC (S = 2L+1 <= 64 states: the AAS batches, L <= 31): one 64-lane wavefront per utterance, lane s
// owns state s for the whole sweep.  No workgroup barrier and no LDS on the recursion's dependency chain: the
// neighbours alpha[s-1], alpha[s-2] (beta[s+1], beta[s+2]) come from wavefront shuffles.
//
// The recursions run in LINEAR space with per-step power-of-two rescaling (Rabiner scaling with c_t = 2^e_t):
//     x_t[s]   = (a_{t-1}[s] + a_{t-1}[s-1] + [skip allowed] a_{t-1}[s-2]) * p_t[ext[s]]          (fp64)
//     e_t      = exponent of the fp32 wave sum of x_t;   a_t = x_t * 2^-e_t  (EXACT in fp64: only the exponent moves)
//     b_t[s]   = (q_{t+1}[s] + q_{t+1}[s+1] + [skip allowed] q_{t+1}[s+2]) * 2^-e_{t+1},  q_{t+1}[s] = b_{t+1}[s] p_{t+1}[ext[s]]
//     log-likelihood = ln2 * sum_t e_t + log(F),  F = a_T[S-1] + a_T[S-2];   posterior(t, s) = a_t[s] b_t[s] / F
// which needs 2 adds + 1 multiply per state and step instead of the 4 fp64 transcendentals of a log-space lse2 chain,
// and is MORE accurate (no log/exp round trips; the scaling is exact).  The fp32 softmax table p[t][c] and the fp64
// alphas live in LDS (T * (4 C + 512) bytes); utterances whose tables do not fit, or with S > 64, take ctc_kernel above.
__global__ __launch_bounds__(64) void ctc_wave_kernel(const float* __restrict__ acts, float* __restrict__ grads,
                                                      const int* __restrict__ labels, const int* __restrict__ lab_off,
                                                      const int* __restrict__ lab_lens, const int* __restrict__ act_lens,
                                                      int C, int N, int Tmax, float* __restrict__ costs, int blank, float gscale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* alpha = reinterpret_cast<double*>(smem);                       // [Tmax][64]
    float* P = reinterpret_cast<float*>(alpha + (size_t)Tmax * 64);        // [Tmax][C] softmax probabilities
    int* esc = reinterpret_cast<int*>(P + (size_t)Tmax * C);               // [Tmax] scaling exponents e_t
    float* occ = reinterpret_cast<float*>(esc + Tmax);                     // [C] per-class posterior mass of one frame
    const int n = blockIdx.x, lane = threadIdx.x;
    const int L = lab_lens[n];
    int Tn = act_lens[n];
    if (Tn > Tmax) Tn = Tmax;
    const int S = 2 * L + 1;
    const int off = lab_off[n];
    const int e_s = (lane < S) ? ((lane & 1) ? labels[off + (lane >> 1)] : blank) : blank;   // ext[s]
    const int e_m2 = __shfl_up(e_s, 2, 64), e_p2 = __shfl_down(e_s, 2, 64);
    const bool live = lane < S;
    const bool skip_in = live && lane >= 2 && e_s != blank && e_s != e_m2;                  // s-2 -> s allowed
    const bool skip_out = lane + 2 < S && e_p2 != blank && e_p2 != e_s;                      // s -> s+2 allowed
    // ---- stage the utterance's activations, then turn every row into its softmax (lane t owns row t)
    for (int i = lane; i < Tn * C; i += 64) {
        const int t = i / C, k = i - t * C;
        P[i] = acts[((int64_t)t * N + n) * C + k];
    }
    if (lane < C) occ[lane] = 0.f;
    __syncthreads();
    for (int t = lane; t < Tn; t += 64) {
        float* row = P + (size_t)t * C;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, row[c]);
        double sum = 0.0;
        for (int c = 0; c < C; ++c) sum += exp((double)row[c] - (double)m);
        const double lse = (double)m + log(sum);
        // (clamped to the smallest normal fp32: a class 87+ nats below the maximum keeps a representable, negligible mass
        //  instead of flushing to zero and turning a feasible alignment into an infinite cost)
        for (int c = 0; c < C; ++c) row[c] = fmaxf((float)exp((double)row[c] - lse), 1.17549435e-38f);
    }
    __syncthreads();
    bool feasible = Tn >= 1;
    double ll = -INFINITY, F = 0.0;
    if (feasible) {
        // ---- alpha sweep
        long long esum = 0;
        double a = (live && lane < 2) ? (double)P[e_s] : 0.0;
        float pn = (Tn > 1) ? P[(size_t)C + e_s] : 0.f;           // p_{t}[ext[s]] for the next step, fetched one step ahead
        for (int t = 0;; ++t) {
            const float sum = wave_sum((float)a);
            int e = 0;
            if (sum > 0.f) e = (int)((__float_as_uint(sum) >> 23) & 255u) - 126;
            a = ldexp(a, -e);                                         // exact
            esum += e;
            alpha[(size_t)t * 64 + lane] = a;
            if (lane == 0) esc[t] = e;
            if (t + 1 >= Tn) break;
            const double p = (double)pn;
            if (t + 2 < Tn) pn = P[(size_t)(t + 2) * C + e_s];
            const double a1 = __shfl_up(a, 1, 64), a2 = __shfl_up(a, 2, 64);
            double x = a;
            if (lane >= 1) x += a1;
            if (skip_in) x += a2;
            a = live ? x * p : 0.0;
        }
        const double f1 = __shfl(a, S - 1, 64), f2 = (S > 1) ? __shfl(a, S - 2, 64) : 0.0;
        F = f1 + f2;
        feasible = F > 0.0;
        if (feasible) ll = 0.6931471805599453 * (double)esum + log(F);
    }
    if (lane == 0) costs[n] = feasible ? (float)(-ll) : INFINITY;
    if (!grads) return;
    if (!feasible) {
        for (int i = lane; i < Tmax * C; i += 64) grads[((int64_t)(i / C) * N + n) * C + (i % C)] = 0.f;
        return;
    }
    __syncthreads();
    // ---- beta sweep fused with the gradient: grad[t][k] = gscale * (p_t[k] - sum_{s: ext[s] = k} posterior(t, s))
    const double invF = 1.0 / F;
    double b = (live && lane >= S - 2) ? 1.0 : 0.0;
    for (int t = Tn - 1; t >= 0; --t) {
        const double al = alpha[(size_t)t * 64 + lane];
        const float g = (float)(al * b * invF);
        // blanks (even states) by a wavefront reduction, labels by LDS atomics (collisions only for repeated letters)
        const float gb = wave_sum((lane & 1) ? 0.f : g);
        if (live && (lane & 1) && g != 0.f) atomicAdd(&occ[e_s], g);
        if (lane == 0) atomicAdd(&occ[blank], gb);
        __syncthreads();
        if (lane < C) {
            grads[((int64_t)t * N + n) * C + lane] = gscale * (P[(size_t)t * C + lane] - occ[lane]);
            occ[lane] = 0.f;
        }
        if (t > 0) {
            const double q = b * (double)P[(size_t)t * C + e_s];
            const double q1 = __shfl_down(q, 1, 64), q2 = __shfl_down(q, 2, 64);
            double x = q;
            if (lane + 1 < S) x += q1;
            if (skip_out) x += q2;
            b = live ? ldexp(x, -esc[t]) : 0.0;
        }
        __syncthreads();
    }
    for (int i = lane; i < (Tmax - Tn) * C; i += 64) grads[((int64_t)(Tn + i / C) * N + n) * C + (i % C)] = 0.f;
}

// workspace in 4-byte units (the scratch itself is fp64)
size_t ws_floats(int minibatch, int max_T, int smax) { return 2 * (size_t)minibatch * ((size_t)max_T * smax + max_T); }

}  // namespace

extern "C" int aas_ctc_loss_async(aasStream_t stream, const float* activations, float* gradients, const int* d_labels,
                                  const int* d_label_offsets, const int* d_label_lens, const int* d_act_lens,
                                  int alphabet, int minibatch, int max_T, int max_label_len, float* costs,
                                  void* workspace, int blank, float grad_scale) {
    AAS_CHECK(activations && d_labels && d_label_offsets && d_label_lens && d_act_lens && costs && workspace,
              "aas_ctc_loss_async: null pointer");
    AAS_CHECK(alphabet > 0 && minibatch > 0 && max_T > 0 && max_label_len >= 0 && blank >= 0 && blank < alphabet,
              "aas_ctc_loss_async: bad sizes");
    const int smax = 2 * max_label_len + 1;
    // single-wavefront kernel when every utterance's states fit one wavefront and its tables fit the LDS
    const size_t wlds = (size_t)max_T * (64 * sizeof(double) + alphabet * sizeof(float) + sizeof(int)) + alphabet * sizeof(float);
    if (smax <= 64 && alphabet <= 64 && wlds <= 150 * 1024 && !(aas_debug_flags_value() & 16384)) {
        static unsigned char attr_done[AAS_MAX_DEV];
        if (aas_raise_dynamic_lds_once(attr_done, reinterpret_cast<const void*>(&ctc_wave_kernel), 150 * 1024)) return 2;
        hipLaunchKernelGGL(ctc_wave_kernel, dim3(minibatch), dim3(64), wlds, (hipStream_t)stream, activations, gradients, d_labels,
                           d_label_offsets, d_label_lens, d_act_lens, alphabet, minibatch, max_T, costs, blank, grad_scale);
        AAS_LAUNCH_CHECK("aas_ctc_loss_async");
        return 0;
    }
    const size_t lds = sizeof(real) * 2 * smax + sizeof(float) * alphabet + sizeof(int) * smax;
    AAS_CHECK(lds <= 64 * 1024, "aas_ctc_loss_async: label length %d too long for the LDS state arrays", max_label_len);
    hipLaunchKernelGGL(ctc_kernel, dim3(minibatch), dim3(256), lds, (hipStream_t)stream, activations, gradients, d_labels,
                       d_label_offsets, d_label_lens, d_act_lens, alphabet, minibatch, max_T, smax, costs,
                       (real*)workspace, blank, grad_scale);
    AAS_LAUNCH_CHECK("aas_ctc_loss_async");
    return 0;
}

// workspace layout for the warp-ctc-shaped synchronous entry point:
//   [float scratch | costs N | labels sumL | offsets N | label_lens N | act_lens N]
extern "C" int aas_ctc_get_workspace_size(const int* h_label_lens, const int* h_act_lens, int alphabet, int minibatch,
                                          int max_T, size_t* bytes) {
    AAS_CHECK(h_label_lens && h_act_lens && bytes && minibatch > 0 && alphabet > 0 && max_T > 0,
              "aas_ctc_get_workspace_size: bad args");
    int maxl = 0;
    size_t suml = 0;
    for (int i = 0; i < minibatch; ++i) {
        AAS_CHECK(h_label_lens[i] >= 0 && h_act_lens[i] >= 0, "aas_ctc_get_workspace_size: negative length");
        if (h_label_lens[i] > maxl) maxl = h_label_lens[i];
        suml += h_label_lens[i];
    }
    *bytes = sizeof(float) * (ws_floats(minibatch, max_T, 2 * maxl + 1) + minibatch) + sizeof(int) * (suml + 3 * (size_t)minibatch) + 64;
    return 0;
}

extern "C" int aas_compute_ctc_loss(aasStream_t stream, const float* activations, float* gradients,
                                    const int* h_flat_labels, const int* h_label_lens, const int* h_act_lens,
                                    int alphabet, int minibatch, int max_T, float* h_costs, void* workspace,
                                    int blank) {
    AAS_CHECK(activations && h_flat_labels && h_label_lens && h_act_lens && h_costs && workspace,
              "aas_compute_ctc_loss: null pointer");
    hipStream_t s = (hipStream_t)stream;
    int maxl = 0;
    size_t suml = 0;
    std::vector<int> offs(minibatch);
    for (int i = 0; i < minibatch; ++i) {
        offs[i] = (int)suml;
        suml += h_label_lens[i];
        if (h_label_lens[i] > maxl) maxl = h_label_lens[i];
    }
    float* wsf = (float*)workspace;
    float* d_costs = wsf + ws_floats(minibatch, max_T, 2 * maxl + 1);
    int* d_lab = (int*)(d_costs + minibatch);
    int* d_off = d_lab + suml;
    int* d_ll = d_off + minibatch;
    int* d_al = d_ll + minibatch;
    if (suml) AAS_HIP(hipMemcpyAsync(d_lab, h_flat_labels, sizeof(int) * suml, hipMemcpyHostToDevice, s));
    AAS_HIP(hipMemcpyAsync(d_off, offs.data(), sizeof(int) * minibatch, hipMemcpyHostToDevice, s));
    AAS_HIP(hipMemcpyAsync(d_ll, h_label_lens, sizeof(int) * minibatch, hipMemcpyHostToDevice, s));
    AAS_HIP(hipMemcpyAsync(d_al, h_act_lens, sizeof(int) * minibatch, hipMemcpyHostToDevice, s));
    AAS_HIP(hipStreamSynchronize(s));  // offs is a stack vector: the copies must have consumed it
    int rc = aas_ctc_loss_async(stream, activations, gradients, d_lab, d_off, d_ll, d_al, alphabet, minibatch, max_T,
                                maxl, d_costs, workspace, blank, 1.0f);
    if (rc) return rc;
    AAS_HIP(hipMemcpyAsync(h_costs, d_costs, sizeof(float) * minibatch, hipMemcpyDeviceToHost, s));
    AAS_HIP(hipStreamSynchronize(s));
    return 0;
}
