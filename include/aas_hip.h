/*
 * aas_hip.h - C ABI of libaas_hip.so: the MI355X (gfx950) hot path of the AAS training step.
 *
 * The reference (lifelongeek/AAS_enhancement) is pure Python on PyTorch; the native code its
 * hot path executes lives in third-party libraries reached through PyTorch / warp-ctc bindings.
 * Each entry point below names the reference call site (file:line under /root/reference) whose
 * native implementation it replaces.  Plain pointers and sizes only: no torch types.  All
 * pointers are DEVICE pointers unless the parameter name starts with `h_`.  `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  Every function enqueues work on `stream`
 * and returns immediately (exceptions: the two functions documented as synchronous).
 *
 * Return value: 0 on success, non-zero on error; aas_last_error() returns a message for the
 * calling thread's last failure (shape / alignment / capacity / HIP launch errors).
 * Buffers are owned by the caller for the duration of the enqueued work.
 */
#ifndef AAS_HIP_H
#define AAS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* aasStream_t;

/* ---------------------------------------------------------------- launch parameters -----------
 * Queue-time settings exist at two levels.
 * (1) PROCESS settings - aas_set_precision, aas_set_debug_flags, aas_set_rnn_cu_limit, aas_set_rnn_launch_tag, aas_set_gemm_max_steps,
 *     aas_set_wgrad_wg_cap: what a single-threaded host (the reference trainer is one) needs; read when a launch is queued.
 * (2) An aasLaunch: the same settings as a caller-owned struct, plus the row classes of a forward recurrent launch and the h-plane
 *     pitch it reports.  Either passed to ONE call (aas_lstm_fwd_ex / aas_gru_fwd_ex / aas_lstm_bwd_ex / aas_gru_bwd_ex) or installed
 *     for the calling THREAD with aas_launch_scope(): every launch that thread queues while it is installed - through any entry point
 *     of this header - takes each field that is set from the struct and only the unset ones from the process settings.  The struct
 *     stays the caller's: it may change fields between calls (e.g. the CU budget per phase of a training step) without another call
 *     into the library.  Two host threads with a struct each - two trainers, a training and a validation model, autograd's backward
 *     thread and the thread that queued the forward pass - cannot consume or overwrite each other's parameters.
 *     aas_enhancement_amd/ops.py: LaunchState wraps one per trainer. */
typedef struct aasLaunch {
    int size;            /* sizeof(aasLaunch): checked, so that the struct can grow */
    int precision;       /* 0 / 1 / 2 as aas_set_precision; -1 = the process setting */
    int debug_flags;     /* as aas_set_debug_flags; -1 = the process setting */
    int rnn_cu_limit;    /* CUs a persistent recurrent launch may occupy, 0 = the whole device; -1 = the process setting */
    int rnn_tag;         /* >= 1: tag of the persistent recurrent launches (sticky error word); 0 = the process setting */
    int gemm_max_steps;  /* as aas_set_gemm_max_steps, 0 = no cap; -1 = the process setting */
    int wgrad_wg_cap;    /* as aas_set_wgrad_wg_cap; -1 = the process setting */
    int cls_n_first;     /* row classes of the NEXT forward recurrent launch under this struct (see aas_set_rnn_row_classes): rows */
    int cls_T_first;     /*   [0, cls_n_first) x cls_T_first frames, the rest x cls_T_rest; cls_n_first = -1: none.  The launch that */
    int cls_T_rest;      /*   takes them sets cls_n_first back to -1. */
    int fwd_h_pitch;     /* OUT: what aas_rnn_last_fwd_h_pitch() reports for the last forward recurrent launch under this struct */
} aasLaunch;
/* Install `l` (NULL: none) as the calling thread's launch scope; *prev (when prev != NULL) receives the one installed before, to be
 * put back with a second call.  The struct must outlive its installation. */
int aas_launch_scope(aasLaunch* l, aasLaunch** prev);

int aas_version(void);
const char* aas_last_error(void);
/* number of CUs of the current device (persistent recurrent kernels size their grids from it) */
int aas_device_cus(void);
/* Debug / A-B bits (process setting, read when a launch is queued; per thread / per call: aasLaunch.debug_flags).
 * Ablation bits - results are WRONG when one is set (profiling, and the timeout test):
 *   persistent recurrent kernels: 1 skip the exchange loads, 2 skip the MFMAs, 4 skip the wait / poll, 8 skip the
 *     publish stores (consumers then run into their bounded-spin timeout), 64 record phase time stamps;
 *   GEMM kernels: 16 skip the MFMAs, 32 skip the stores, 64 skip the loads, 128 epilogue only, 1 (aas_gemm_f32's LDS-DMA kernel) every
 *     k-step refetches the tile's first one (same instruction stream, L2-resident sources).
 * Kernel-selection bits (results unchanged): 32 plain first exchange load in the forward kernels, 256 all-gather
 *   BPTT instead of the reduce-scatter kernel, 512 16-unit slices, 8192 poll before streaming in the reduce-scatter
 *   BPTT, 1024 128x128 tiles instead of 256x256 and 2048 128x256 instead of 128x128 in aas_gemm_planes, 4096 plain
 *   3-D grid instead of the XCD-aware tile order in aas_gemm_planes, 16384 the general (any S) CTC kernel even when S <= 64,
 *   262144 the plain 3-D grid in the persistent recurrent launches instead of the XCD-aware one (exchange sets per XCD class,
 *     L2-resident publish stores once a set is verified co-located), 524288 XCD-aware grid but write-through publish stores,
 *     67108864 plain grid for the forward launches with more than 8 rows per group (default: XCD-aware grid, write-through),
 *   8388608 / 33554432 256x256 / 256x128 tiles for the wide products of aas_gemm_planes_tn (default 128x128),
 *   65536 / 131072 four waves per workgroup (one per SIMD) instead of eight in aas_gemm_planes_tn / the 128x128 aas_gemm_planes,
 *   32768 BatchNorm on the column-per-thread kernels with fp64 atomics (default where C % 4 == 0: 16-byte loads, per-block partial
 *     sums in a workspace, no atomics - same statistics on every run),
 *   134217728 fp32 mode on the counter-based kernels of round 1 instead of the data-is-the-flag ones, 268435456 fp32 mode: 16x16x4
 *     MFMA tiles also for row groups of <= 8 rows (default there: 4x4x1 blocks - same products, k summed in interleaved chains),
 *   536870912 mode 2: the exact (fp32-input MFMA) LSTM BPTT kernel instead of the six-product one,
 *   1073741824 four-wave workgroups in aas_gemm_f32 (default: eight waves where both operands take 16-byte loads). */
int aas_set_debug_flags(int flags);
/* the bits currently set (bench.py refuses to report a headline value while this is non-zero) */
int aas_get_debug_flags(void);
/* Tag (>= 1) of the persistent recurrent launches queued after the call.  Every bounded spin in those kernels gives
 * up after ~0.5 s and stores the tag of its launch in a sticky error word at byte 4096 of the launch's `sync` buffer
 * (never cleared by the library): the host reads it at its next synchronisation point and names the layer. */
int aas_set_rnn_launch_tag(int tag);
/* Row classes of the calling thread's NEXT aas_lstm_fwd / aas_gru_fwd launch (consumed by it; aas_rnn_fwd refuses them and drops
 * them).  Kept for one round beside the argument form - aasLaunch.cls_* through aas_lstm_fwd_ex / aas_gru_fwd_ex - which is what
 * the host side of this repository uses; the pending setting is thread-local.  Batch rows [0, n_first) hold
 * sequences of T_first frames, rows [n_first, N) of T_rest frames, in a launch of T = max(T_first, T_rest) whose input is anything
 * (e.g. zero padding) beyond a row's length.  The shorter class gets exactly what a launch of its own would give it: zero state
 * in front of its first frame in either direction, h = 0 beyond its last, and - through the stored gate values - zero gate
 * gradients there in the matching BPTT launch, whatever dy holds.  The reference's discriminator pass over an enhanced batch and
 * a clean batch of different padded lengths (trainer_AAS.py:94-107 runs D twice) becomes one batched pass this way. */
int aas_set_rnn_row_classes(int n_first, int T_first, int T_rest);
/* Exchange buffers of the persistent recurrent launches (the `xchg` argument of aas_lstm_fwd / _bwd, aas_gru_fwd / _bwd).  By
 * default every launch poison-fills the part of the buffer it uses first (a memset launch of 8-50 MB per recurrent launch).  A
 * buffer handed to aas_rnn_xchg_prepare ONCE (poison-filled there, `bytes` = at least twice what the largest launch needs:
 * 2 * aas_rnn_xchg_bytes(...)) is MANAGED: the library alternates the launches on it between its two halves, every kernel
 * re-poisons inside the kernel what its predecessor dirtied in the other half, and the fp32 forward kernels exchange h_t through a
 * ring of four time slots that its producers clean behind themselves - no memset launch at all between recurrent launches.  The
 * caller must not touch a managed buffer, must use it from one stream at a time, and calls aas_rnn_xchg_forget before freeing it.
 * A launch under hipGraph capture, or one that needs more than half the buffer, ends the management of that buffer (the library
 * falls back to the poison fill per launch from then on): always correct. */
int aas_rnn_xchg_prepare(aasStream_t stream, void* xchg, size_t bytes);
int aas_rnn_xchg_forget(void* xchg);
int aas_rnn_xchg_is_managed(void* xchg);   /* 1 while the buffer is managed (a fallen-back buffer may be prepared again) */
/* (Argument form: aasLaunch.fwd_h_pitch.)  > 0 when the calling thread's LAST aas_lstm_fwd / aas_gru_fwd call left h_t of every time step but each direction's last one in its
 * exchange buffer as operand planes (rows [2][T][N], this many bytes per row, interleaved hi | lo per 32 units, pad units
 * zero; the unpublished rows stay poisoned = NaN): the B operand of that layer's recurrent weight-gradient product
 * (aas_gemm_planes_tn).  0: fp32 kernels ran, or the batch was processed in several launches. */
int aas_rnn_last_fwd_h_pitch(void);
/* Cap on the grid of the aas_gemm_planes_tn launches queued after the call (0 = none, the default: one workgroup per tile):
 * with a cap each workgroup walks several tiles, so the weight-gradient products never hold more than `workgroups` CUs while
 * a persistent recurrent launch waits to become resident (an A/B switch: worth 0.4 ms / step at config 2 before the XCD-aware
 * recurrent launches, nothing since; trainers leave it off). */
int aas_set_wgrad_wg_cap(int workgroups);
/* Matrix-product operand precision of every GEMM and recurrent product queued after the call:
 *   0 (default) = fp32: fp32 operands on fp32-input MFMA with fp32 accumulation - the arithmetic of the reference's cuDNN /
 *                 cuBLAS fp32 path (trainer_AAS.py:69-73 moves the models to the GPU as fp32);
 *   1           = split-bf16 fast mode: each fp32 operand is carried as bf16 hi + bf16 lo and the product as
 *                 hi*hi + lo*hi + hi*lo with fp32 accumulation (16+ mantissa bits per operand, ~2^-17 per product: NARROWER than
 *                 fp32, inside the 1e-3 / 1e-2 parity budget; 3 bf16 MFMAs instead of 16 fp32-MFMA issue slots);
 *   2           = fp32-EQUIVALENT: the small GEMMs and the recurrent products as in mode 0 (fp32-input MFMA), except that aas_lstm_bwd
 *                 runs its partial products as six bf16 products of three-term operands where instantiated (256 < H <= 512); the large GEMMs are
 *                 the caller's to run as six bf16 products of three-term operands (aas_split_planes3 + the plane
 *                 kernels over both plane sets, below): every product keeps all 24 operand bits, dropped cross terms <= 2^-25 relative. */
int aas_set_precision(int mode);
/* cap on the CUs one persistent recurrent launch occupies (0 = whole device): lets two independent chains of recurrent
 * launches (trainer_AAS.py:153-172: discriminator pass and acoustic pass) run side by side on two streams */
int aas_set_rnn_cu_limit(int cus);

/* ---------------------------------------------------------------- dense linear algebra --------
 * fp32 MFMA GEMM  C = op(A) op(B) [+ bias broadcast over rows] [+ addend] [+ C if accumulate].
 * Replaces the cuDNN/cuBLAS GEMMs under nn.LSTM / nn.GRU input projections (model.py:73-74,94-95),
 * nn.Conv1d k=1 (model.py:216-217), nn.Conv1d k=11 as implicit-im2col GEMM (model.py:289,297),
 * nn.Linear (model.py:317) and all their backward (dgrad / wgrad) products.
 *   mode 0 (NT): A[M,K] lda, B[N,K] ldb      (y = x W^T)
 *   mode 1 (NN): A[M,K] lda, B[K,N] ldb      (dx = dy W)
 *   mode 2 (TN): A[K,M] lda, B[K,N] ldb      (dW = dy^T x), reduction row r addressed two-level:
 *                row(r) = (r / kdiv) * kouter + (r % kdiv) * ld   when kdiv > 0 (per operand)
 * batch > 1: independent problems at A + b*strideA, B + b*strideB, C + b*strideC (NT/NN only).
 */
#define AAS_GEMM_NT 0
#define AAS_GEMM_NN 1
#define AAS_GEMM_TN 2
int aas_gemm_f32(aasStream_t stream, int mode, int M, int N, int K,
                 const float* A, int64_t lda, const float* B, int64_t ldb,
                 float* C, int64_t ldc,
                 const float* bias, const float* addend, int64_t ldd, int accumulate,
                 int batch, int64_t strideA, int64_t strideB, int64_t strideC,
                 int kdivA, int64_t kouterA, int kdivB, int64_t kouterB);
/* `n` (<= 4) products of EQUAL shape and mode in one launch, C_i (+)= op(A_i) op(B_i): the four weight-gradient products of a
 * bidirectional recurrent layer (dW_ih, dW_hh per direction: model.py:73-74,94-95 backward) share d(gates) and fill the chip together
 * where each alone needs split-K.  A / B / C: HOST arrays of n device pointers, K: host array of the n reduction extents (the
 * recurrent products of a layer sum one time step less than the input products).  TN only: kdiv > 0 addresses the reduction rows
 * of BOTH operands two-level, row(r) = (r / kdiv) * kouter + (r % kdiv) * ld - the rows of ONE utterance class of a time-major
 * [T, N, ...] tensor are kdiv = class size, kouter = N * ld, from the class's first row; d_alpha (device scalar or null) multiplies
 * the products: the D-step weight gradients of the batched [enhanced; clean] discriminator pass carry the BEGAN factor (-kt) on the
 * enhanced class only (trainer_AAS.py:152-160) - two launches, no scaled copies of x / h.  fp32 arithmetic on the LDS-DMA kernel when
 * every operand takes 16-byte chunks, else n launches of the general kernel. */
int aas_gemm_f32_multi(aasStream_t stream, int mode, int n, int M, int N, const int* K, const float* const* A, int64_t lda,
                       const float* const* B, int64_t ldb, float* const* C, int64_t ldc, int accumulate, int kdiv, int64_t kouterA,
                       int64_t kouterB, const float* d_alpha);
/* Kernel choice of aas_gemm_f32 in fp32 arithmetic: 0 (default) = the LDS-DMA kernel (128 x 128 x 32 tiles, operands by
 * global_load_lds, split-K through partial slabs + a reduce launch) wherever both operands take 16-byte chunks; 1 = always the
 * register-staged kernel (atomic split-K).  Results agree to fp32 summation order.  Environment: AAS_GEMM32=0 selects 1. */
int aas_set_gemm_variant(int variant);
/* Longest life of a workgroup of the LDS-DMA GEMM in k-steps of 32 (~2 us each); 0 = no cap (default; AAS_GEMM32_MAXSTEPS).  Deep
 * products are split further along K (slabs + reduce launch) so that every workgroup hands its CU back within that time: the
 * persistent recurrent launches of the training step (trainer_AAS.py:131-194 on this build's schedule) become resident only when
 * enough CUs are free at once. */
int aas_set_gemm_max_steps(int steps);
int aas_get_gemm_max_steps(void);
/* The library's own scratch (split-K slabs of aas_gemm_f32, BatchNorm partial sums) is one block per (device, stream), grown on
 * demand; a block that is outgrown is retired, NOT freed, because a captured hipGraph may hold its address.  This call frees the
 * retired blocks (device-synchronising): only when no captured graph that ran library launches is going to be replayed again.
 * Returns the number of blocks released.  Synchronous. */
int aas_release_retired_workspaces(void);
/* C[M,N] (+)= sum_r kscale[r % knb] * A[r,M]^T B[r,N]   (fp32 mode): the TN product of aas_gemm_f32 with a per-reduction-row weight
 * applied while A is staged - the D-step weight gradients of a batched [enhanced; clean] discriminator pass carry the BEGAN factor
 * (-kt) on the enhanced utterances only (trainer_AAS.py:152-160); r = (t, n) time-major, knb = utterances per time step. */
int aas_gemm_tn_rowscaled_f32(aasStream_t stream, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                              float* C, int64_t ldc, int accumulate, const float* d_kscale, int knb);

/* Split-bf16 GEMM on pre-split operand planes (same products as aas_gemm_f32 under aas_set_precision(1), with the
 * fp32 -> (hi, lo) bf16 split hoisted out of the k-loop into one HBM-bound pass per operand):
 *   C[M,N] (+)= A[M,K] B[N,K]^T (+ bias[N]) (+ addend[M,N]),  A/B given as interleaved bf16 planes: row r, 32-wide
 *   k-block b = 64 B of hi = rne(x) then 64 B of lo = rne(x - hi) at byte r*ld*4 + b*128 (one 128-byte line per block),
 *   K % 32 == 0 with zero-filled pad, lda/ldb/strides in elements and multiples of 32, 128-byte aligned.
 * aas_split_planes:   planes[r][k] = split(src[r*ld + k] * row_scale[r % nb]), k < K; zeros for K <= k < Kp.
 * aas_split_planes_t: planes[c][t*nbp + n] = split(src[(t*nb + n)*ld + c] * row_scale[n]) for c < C, time-major src
 *                     [T*nb rows]; pads (n >= nb, and [T*nbp, Kp)) are written as zeros.  This is the operand form of the
 *                     weight-gradient products dW = d(gates)^T x (model.py:73-74 backward), with the per-utterance loss
 *                     weights of the batched discriminator pass (trainer_AAS.py:153-167) folded into the split. */
int aas_gemm_planes(aasStream_t stream, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb,
                    float* C, int64_t ldc, const float* bias, const float* addend, int64_t ldd, int accumulate,
                    int batch, int64_t strideA, int64_t strideB, int64_t strideC);
/* out[rows,K] = a + b (+ c) and, in the same pass, its operand planes (what aas_split_planes would make of `out`): the
 * direction sum + residual of one recurrent layer (model.py:85,104,223-226) is the A operand of the next layer's projection. */
int aas_add3_planes_f32(aasStream_t stream, float* out, const float* a, const float* b, const float* c, int64_t rows, int K, int Kp,
                        void* planes);
/* aas_split_planes_t with PLANES as the source: planes[c][t*nbp + n] = split((hi + lo)(row (t, n), column c) * row_scale[n])
 * for the row-major operand planes [T*nb rows][src_ld] that aas_lstm_bwd_planes / aas_gru_bwd_planes write. */
int aas_planes_transpose(aasStream_t stream, const void* src_planes, int64_t src_ld, int T, int nb, int nbp, int C, int64_t Kp,
                         void* planes, const float* row_scale);
/* `count` (<= 4) products of one shape in ONE launch, each with its own plane operands and result (HOST arrays of
 * device pointers); always accumulates: C_i[M,N] += A_i[M,K] B_i[N,K]^T.  The four weight-gradient products of a
 * bidirectional recurrent layer (dW_ih, dW_hh per direction; model.py:73-74,94-95 backward) run as one launch. */
int aas_gemm_planes_multi(aasStream_t stream, int M, int N, int K, int count, const void* const* h_A,
                          const void* const* h_B, float* const* h_C, int64_t lda, int64_t ldb, int64_t ldc);
/* The weight-gradient products of a recurrent layer (dW = d(gates)^T x, d(gates)^T h_{t-+1}; model.py:73-74,94-95 backward)
 * straight from ROW-MAJOR planes, no transposed copies:  C_i[m][n] (+)= alpha_i * sum_{r' < K_i} A_i[ra(r')][acol0_i + m] * B_i[rb(r')][n],
 * with the row map r' -> (t, n) = (r' / Ns, r' % Ns), ra = (t + ta_i)*Nb + n0_i + n, rb = (t + tb_i)*Nb + n0_i + n: the rows of
 * one utterance class of every time step (a batched discriminator pass = two problems with alpha = -kt / 1, trainer_AAS.py:137-
 * 151), time shifts for the recurrent weights.  A_i, B_i: interleaved planes (layout of aas_split_planes), lda / ldb BYTES per
 * row (multiples of 128), acols / bcols valid plane columns; rows m < msplit_i go to C0_i, the rest to C1_i (the two directions'
 * weights are separate tensors); alpha_i: device scalar or NULL (= 1); all tables are HOST arrays of `count` (<= 8) entries;
 * zero512: >= 512 zero bytes on the device (source of masked rows / columns). */
int aas_gemm_planes_tn(aasStream_t stream, int count, const void* const* h_A, const void* const* h_B, float* const* h_C0,
                       float* const* h_C1, const float* const* h_alpha, const int* h_M, const int* h_N, const int* h_K,
                       const int* h_msplit, const int* h_acol0, const int* h_n0, const int* h_ta, const int* h_tb,
                       const int64_t* h_lda, const int* h_acols, const int64_t* h_ldb, const int* h_bcols,
                       const int64_t* h_ldc, int Ns, int Nb, const void* zero512, int accumulate);
int aas_split_planes(aasStream_t stream, const float* src, int64_t ld, int64_t rows, int K, int Kp, void* planes,
                     const float* row_scale, int nb);
int aas_split_planes_t(aasStream_t stream, const float* src, int64_t ld, int T, int nb, int nbp, int C, int64_t Kp,
                       void* planes, const float* row_scale);
/* aas_split_planes_t with the T blocks of nb source rows `tstride` (non-zero, may be negative) elements apart (row (t, n) at
 * src + t*tstride + n*ld):
 * [W_ih ; W_ih_reverse] - two tensors of one flat parameter buffer - as the transposed operand of dx = d(gates) W_ih. */
int aas_split_planes_t2(aasStream_t stream, const float* src, int64_t ld, int64_t tstride, int T, int nb, int nbp, int C,
                        int64_t Kp, void* planes, const float* row_scale);

/* ---------------------------------------------------------------- layout / elementwise --------
 * out[b, c, r] = in[b, r, c]  with element strides (in: isb, isr, c contiguous; out: osb, osc, r
 * contiguous).  Replaces the .transpose() chains at model.py:222,227,328,331. */
int aas_transpose_f32(aasStream_t stream, const float* in, float* out, int B, int R, int C,
                      int64_t isb, int64_t isr, int64_t osb, int64_t osc);
/* out[b, c, r] += in[b, r, c], same addressing: a weight gradient computed in the GEMM's [M, KW*F] layout ADDED into the parameter's
 * .grad in the module's [M, F, KW] layout (nn.Conv1d weight, model.py:289,297) - what autograd's gradient accumulation would do
 * with a transposed copy and an add. */
int aas_transpose_add_f32(aasStream_t stream, const float* in, float* out, int B, int R, int C,
                          int64_t isb, int64_t isr, int64_t osb, int64_t osc);
/* out[b, a, :] = in[a, b, :]  ([A,B,C] -> [B,A,C], C contiguous): N,T,C <-> T,N,C (model.py:328,331) */
int aas_swap01_f32(aasStream_t stream, const float* in, float* out, int A, int B, int C);
/* out = a + b (+ c if c != NULL), n elements.  Direction sum + residual (model.py:85,104,223-226). */
int aas_add3_f32(aasStream_t stream, float* out, const float* a, const float* b, const float* c, int64_t n);
/* out[(t,n), :] = in[(t,n), :] * scale[n] for time-major rows (row = t*Nb + n); out may alias in.
 * Per-utterance weighting of the weight-gradient products when D(enhanced) and D(clean) share one
 * batched pass: parameter gradients of the enhanced half carry (-kt) (trainer_AAS.py:152-160). */
int aas_scale_rows_f32(aasStream_t stream, float* out, const float* in, const float* scale, int64_t rows, int Nb, int C);
/* y[i] = ref[i] >= 0 ? x[i] : slope * x[i].  nn.LeakyReLU(negative_slope = map) on its own - the first activation of
 * AM_training/model.py:364-367 DeepSpeech_ken(include_first_BN=False), where no BatchNorm launch is there to fuse it into:
 * forward with ref = x, backward with x = dy and ref = the forward input (y may alias x). */
int aas_leaky_relu_f32(aasStream_t stream, float* y, const float* x, const float* ref, float slope, int64_t n);
/* y = alpha * x + beta * y  (gradient scaling / accumulation; y may alias x) */
int aas_axpby_f32(aasStream_t stream, float* y, const float* x, float alpha, float beta, int64_t n);
/* y = (alpha * d_alpha[0]) * x with the factor read from device memory (y may alias x) */
int aas_scale_dev_f32(aasStream_t stream, float* y, const float* x, const float* d_alpha, float alpha, int64_t n);
/* out[c] (+)= sum_r x[r, c]   (bias gradients) */
int aas_colsum_f32(aasStream_t stream, const float* x, int64_t R, int C, int64_t ld, float* out, int accumulate);
/* acc[0] += sum x^2 (fp64 accumulator on device). trainer_AAS.py:353-361 get_gradient_norm. */
int aas_sqsum_f32(aasStream_t stream, const float* x, int64_t n, double* acc);

/* ---------------------------------------------------------------- recurrent layers ------------
 * Bidirectional, bias-free LSTM layer recurrence (cuDNN RNN under model.py:94-95,102).
 *   pre   [T,N,2,4H]  input projections x W_ih^T for both directions (d=0 forward, d=1 reverse),
 *                     gate order i,f,g,o (PyTorch)
 *   w_hh, w_hh_rev  [4H,H] each (weight_hh_l0, weight_hh_l0_reverse)
 *   hout  [2,T,N,H]   h_t per direction (time index = position in the sequence, both directions)
 *   gact  [2,T,N,H,4]  post-nonlinearity gates i,f,g,o per unit (saved for backward); cst [2,T,N,H] cell states
 *   sync  >= aas_rnn_sync_bytes() bytes of zero-initialisable device scratch (zeroed by the call)
 * Persistent kernel: one workgroup per (unit slice, batch group, direction); W_hh slices stay in registers for all T
 * steps.  `sync` holds the arrival counters of the counter-based fallback kernels (re-zeroed per launch) and, at word 1024, a
 * sticky timeout flag that every bounded spin of every kernel can raise.  With an exchange buffer, the forward launches
 * all-gather h_t through poison-tagged 128-byte lines (32 fp32 values, or bf16 hi|lo halves under aas_set_precision(1); 16- or
 * 32-unit slices per workgroup, chosen from the CU budget) and the backward launches run BPTT as a per-step reduce-scatter of
 * K-split partial dh through a tagged two-slot ring (csrc/rnn_bwd_rs_kernel.h), in BOTH precisions: fp32-input MFMA on the
 * exchanged fp32 values, or three bf16 MFMA products on their hi / lo halves; aas_set_rnn_cu_limit() bounds the grid of the
 * launches queued after it.
 */
size_t aas_rnn_sync_bytes(void);
/* bytes of the exchange scratch `xchg` (gates = 4 LSTM / 3 GRU; one buffer per stream that runs recurrent launches); pass
 * xchg = NULL to force the counter-based fp32 kernels of round 1 (csrc/rnn_kernel.h).  With xchg != NULL the exchanged values
 * are fp32 words or bf16 hi/lo pairs (forward) and tagged fp32 partial sums (BPTT; the 2-bit step tag replaces the two low
 * mantissa bits of a partial, rounded to nearest in the fp32 mode). */
size_t aas_rnn_xchg_bytes(int T, int N, int H, int gates);
int aas_lstm_fwd(aasStream_t stream, int T, int N, int H, const float* pre, const float* w_hh,
                 const float* w_hh_rev, float* hout, float* gact, float* cst, void* sync, void* xchg);
/* BPTT.  dy [T,N,H] is the gradient wrt (h_fwd + h_bwd) (shared by both directions);
 * dgates [T,N,2,4H] receives d(loss)/d(pre) (same layout as pre). */
int aas_lstm_bwd(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh,
                 const float* w_hh_rev, const float* gact, const float* cst, float* dgates, void* sync, void* xchg);

/* aas_lstm_bwd with d(gates) written straight in the operand form of the layer's GEMMs instead of fp32: interleaved
 * bf16 hi|lo planes [T*N rows][Kp] (Kp = 2*4*H rounded up to 32; column k = d*4H + g*H + unit; pad columns zeroed) - what
 * aas_split_planes would produce from the fp32 tensor, without that tensor or the pass.  Returns 3 (and does nothing)
 * when the split-bf16 reduce-scatter kernel cannot be used (exact-fp32 mode, no exchange buffer, unsupported H): the
 * caller then uses aas_lstm_bwd + aas_split_planes.  aas_gru_bwd_planes likewise (Kp = 2*3*H rounded up to 32). */
int aas_lstm_bwd_planes(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev,
                        const float* gact, const float* cst, void* dgates_planes, int Kp, void* sync, void* xchg);
int aas_gru_bwd_planes(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev,
                       const float* hout, const float* gact, void* dgx_planes, void* dgh_planes, int Kp, void* sync, void* xchg);
/* aas_lstm_bwd in the fp32-equivalent mode (precision 2) with d(gates) written as the two three-term plane sets of the six-product
 * GEMMs - what aas_split_planes3 would make of the fp32 tensor with row_pitch_bytes = 8 Kp (set Q1 at the start of a row, set Q2 at
 * byte 4 Kp), without that tensor or the pass.  Returns 3 (and does nothing) where the six-product BPTT kernel is not instantiated
 * (other modes, no exchange buffer, H outside (256, 512]): the caller then uses aas_lstm_bwd + aas_split_planes3. */
int aas_lstm_bwd_planes3(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev,
                         const float* gact, const float* cst, void* dgates_sets, int Kp, void* sync, void* xchg);

/* nn.RNN(nonlinearity='tanh', bias=False, bidirectional=True) - the `rnn` entry of supported_rnns (model.py:12-17), selectable with
 * --rnn_type rnn (config.py:44):  h_t = tanh(pre_t + W_hh h_{t-1}).
 *   pre [T,N,2,H] (input projections of both directions), w_hh / w_hh_rev [H,H], hout [2,T,N,H]; gact [2,T,N,H,4] scratch that the
 *   backward pass reads back (slot 0 = h_t).  aas_rnn_bwd: dy [T,N,H] = gradient wrt (h_fwd + h_bwd); dpre [T,N,2,H] receives
 *   d(loss)/d(pre).  fp32-input MFMA in both precision modes (counter-based persistent kernel). */
int aas_rnn_fwd(aasStream_t stream, int T, int N, int H, const float* pre, const float* w_hh, const float* w_hh_rev, float* hout,
                float* gact, void* sync);
int aas_rnn_bwd(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev, const float* gact,
                float* dpre, void* sync);

/* Bidirectional bias-free GRU (cuDNN RNN under model.py:73-74,83), gate order r,z,n:
 *   pre [T,N,2,3H]; w_hh, w_hh_rev [3H,H]; hout [2,T,N,H];
 *   gact [2,T,N,H,4] saves r, z, n and hn = (W_hn h_{t-1}) per unit for backward. */
int aas_gru_fwd(aasStream_t stream, int T, int N, int H, const float* pre, const float* w_hh,
                const float* w_hh_rev, float* hout, float* gact, void* sync, void* xchg);
/* dgx [T,N,2,3H] = d/d(pre) (for dW_ih, dx); dgh [T,N,2,3H] = d/d(W_hh h) (for dW_hh). */
int aas_gru_bwd(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh,
                const float* w_hh_rev, const float* hout, const float* gact, float* dgx, float* dgh, void* sync,
                void* xchg);

/* The four launches with their parameters as an ARGUMENT: *launch carries the row classes of the forward launch (consumed by it),
 * the CU budget, the launch tag, the kernel-selection bits and the arithmetic mode FOR THIS CALL (unset fields: the process
 * settings) and receives fwd_h_pitch; nothing process-wide is written.  launch = NULL: exactly the plain entry point.  What
 * nn.LSTM / nn.GRU's per-call arguments are to the reference (model.py:73-74,94-95,102-104): no state between calls. */
int aas_lstm_fwd_ex(aasStream_t stream, int T, int N, int H, const float* pre, const float* w_hh, const float* w_hh_rev, float* hout,
                    float* gact, float* cst, void* sync, void* xchg, aasLaunch* launch);
int aas_lstm_bwd_ex(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev,
                    const float* gact, const float* cst, float* dgates, void* sync, void* xchg, aasLaunch* launch);
int aas_gru_fwd_ex(aasStream_t stream, int T, int N, int H, const float* pre, const float* w_hh, const float* w_hh_rev, float* hout,
                   float* gact, void* sync, void* xchg, aasLaunch* launch);
int aas_gru_bwd_ex(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev,
                   const float* hout, const float* gact, float* dgx, float* dgh, void* sync, void* xchg, aasLaunch* launch);
/* The LSTM forward launch WITH the layer's input projection inside: x [T,N,I] and w_ih / w_ih_rev [4H,I] instead of `pre` - the whole of
 * nn.LSTM's forward for one bias-free bidirectional layer (model.py:73-74,83,101-105) in one launch, no [T N, 8H] pre-activation tensor.
 * -> 0 launched; 3 = not covered in the present settings (needs: fp32 mode, H and I <= 512, I % 4 == 0, <= 8 batch rows per workgroup
 * on the CU budget with the whole batch in one launch - e.g. the enhancement network's N = 30, H = 500 over the whole chip): NOTHING
 * was launched or consumed, the caller forms `pre` with aas_gemm_f32 and calls aas_lstm_fwd_ex; 1 = error.  Outputs as aas_lstm_fwd. */
int aas_lstm_fwd_x_ex(aasStream_t stream, int T, int N, int H, int I, const float* x, const float* w_ih, const float* w_ih_rev,
                      const float* w_hh, const float* w_hh_rev, float* hout, float* gact, float* cst, void* sync, void* xchg,
                      aasLaunch* launch);

/* ---------------------------------------------------------------- batch norm (train mode) -----
 * Rows-by-channels BatchNorm with batch statistics (nn.BatchNorm1d in train mode: model.py:72,82
 * via SequenceWise :44-49, :290,298, :316; the reference never calls ASR.eval()).
 * x,y [R,C] (ld = C).  stats [4,C] fp32: mean, invstd, (bwd) sum_dy, sum_dy_xhat.
 * slope != 1 fuses LeakyReLU(negative_slope = slope) after the affine (model.py:291,299: slope=map).
 * running_mean/var (may be NULL) are updated with `momentum` and the unbiased variance; num_batches_tracked (device int64, may be
 * NULL) is incremented by the apply launch (nn.BatchNorm1d's counter - no separate launch on the chain).
 * wsd: >= 2*C doubles of scratch (used by the general kernels only: with C % 4 == 0 and 16-byte aligned tensors the per-block
 * partial sums go to a per-stream workspace the library owns - two launches, no zero-fill, no atomics). */
int aas_bn_fwd(aasStream_t stream, const float* x, float* y, int64_t R, int C,
               const float* gamma, const float* beta, float eps, float slope,
               float* stats, float* running_mean, float* running_var, float momentum, double* wsd,
               long long* num_batches_tracked);
int aas_bn_bwd(aasStream_t stream, const float* x, const float* dy, float* dx, int64_t R, int C,
               const float* gamma, const float* beta, float slope, float* stats,
               float* dgamma, float* dbeta, int accumulate, double* wsd);

/* The same two passes as separate launches, so that a data-parallel caller can all-reduce the per-channel sums between
 * them (SyncBN over the global batch): aas_bn_stats leaves wsd[0..C) = sum x, wsd[C..2C) = sum x^2 of THIS
 * rank's R rows; the caller sums wsd (and the row count, as a double at d_rows[0]) over the ranks; aas_bn_apply then
 * normalises with the global statistics (d_rows = NULL: the local R).  Backward likewise: aas_bn_bwd_reduce leaves
 * sum dy / sum dy*xhat; aas_bn_bwd_apply takes the all-reduced sums in `wsd` and this rank's own in `wsd_local`
 * (NULL = wsd) - dgamma/dbeta are local sums, because the gradient all-reduce adds the ranks up. */
int aas_bn_stats(aasStream_t stream, const float* x, int64_t R, int C, double* wsd);
int aas_bn_apply(aasStream_t stream, const float* x, float* y, int64_t R, int C, const float* gamma, const float* beta,
                 float eps, float slope, float* stats, float* running_mean, float* running_var, float momentum,
                 const double* wsd, const double* d_rows, long long* num_batches_tracked);
int aas_bn_bwd_reduce(aasStream_t stream, const float* x, const float* dy, int64_t R, int C, const float* gamma,
                      const float* beta, float slope, const float* stats, double* wsd);
int aas_bn_bwd_apply(aasStream_t stream, const float* x, const float* dy, float* dx, int64_t R, int C, const float* gamma,
                     const float* beta, float slope, float* stats, float* dgamma, float* dbeta, int accumulate,
                     const double* wsd, const double* wsd_local, const double* d_rows);
/* Eval-mode BatchNorm (running statistics; model.eval() in AM_training/train.py:357 validation) and the row softmax of
 * InferenceBatchSoftmax in eval mode (model.py:58-64; C <= 64 classes, one wavefront per row). */
int aas_bn_eval(aasStream_t stream, const float* x, float* y, int64_t R, int C, const float* gamma, const float* beta,
                const float* running_mean, const float* running_var, float eps, float slope);
int aas_softmax_rows(aasStream_t stream, const float* x, float* y, int64_t R, int C);

/* conv1d backward-data helper: dx[n,t,f] = sum_{kk} dcol[n,(t-kk)/s,kk,f] over valid (t-kk)%s==0.
 * dcol [N,T1,KW,F] (from the NN GEMM dOut x W2), dx [N,T,F] channels-last.  model.py:289,297. */
int aas_col2im_f32(aasStream_t stream, const float* dcol, float* dx, int N, int T, int T1, int F, int KW, int stride);

/* ---------------------------------------------------------------- losses ----------------------
 * L1Loss_mask (model.py:19-31): loss_sum[0] (+)= sum |a-b| over ALL elements (mask not applied,
 * as in the reference); the caller divides by nElement = #unmasked (n,t). fp64 accumulator. */
int aas_l1_fwd(aasStream_t stream, const float* a, const float* b, int64_t n, double* loss_sum);
/* ga = s*sign(a-b) (if ga), gb = -s*sign(a-b) (if gb), s = scale * (d_scale ? d_scale[0] : 1): the upstream
 * gradient can stay on the device (no host sync in backward); `accumulate` adds into ga/gb. */
int aas_l1_bwd(aasStream_t stream, const float* a, const float* b, int64_t n, float scale, const float* d_scale,
               float* ga, float* gb, int accumulate);
/* The same two on `rows` rows of `cols` valid elements with row pitches (lda, ldb, ldga, ldgb >= cols): the L1 sum of ONE utterance
 * class of a batched pass over a noisy / clean pair of different padded length (trainer_AAS.py:146-147,176-177 on batches from two
 * loaders) - its own frames inside a tensor padded to the longer class's length.  aas_l1_bwd2d also writes zeros to columns
 * [cols, ga_cols) of ga: no gradient reaches the padding. */
int aas_l1_fwd2d(aasStream_t stream, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int cols, double* loss_sum);
int aas_l1_bwd2d(aasStream_t stream, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int cols, float scale,
                 const float* d_scale, float* ga, int64_t ldga, int ga_cols, float* gb, int64_t ldgb);

/* CTC (warpctc_pytorch.CTCLoss; call site trainer_AAS.py:168).  Mirrors warp-ctc's C ABI
 * (get_workspace_size / compute_ctc_loss): activations [T,N,C] pre-softmax, gradients [T,N,C]
 * (may be NULL), labels/lengths on the HOST (int32), blank index `blank`.
 * aas_compute_ctc_loss is SYNCHRONOUS like warp-ctc's gpu path: h_costs[N] is on the host.
 * aas_ctc_loss_async keeps everything on the device: d_labels (flat), d_label_offsets[N],
 * d_label_lens[N], d_act_lens[N], costs[N] device; max_label_len bounds S = 2L+1. */
int aas_ctc_get_workspace_size(const int* h_label_lens, const int* h_act_lens, int alphabet, int minibatch,
                               int max_T, size_t* bytes);
int aas_compute_ctc_loss(aasStream_t stream, const float* activations, float* gradients,
                         const int* h_flat_labels, const int* h_label_lens, const int* h_act_lens,
                         int alphabet, int minibatch, int max_T, float* h_costs, void* workspace, int blank);
int aas_ctc_loss_async(aasStream_t stream, const float* activations, float* gradients,
                       const int* d_labels, const int* d_label_offsets, const int* d_label_lens,
                       const int* d_act_lens, int alphabet, int minibatch, int max_T, int max_label_len,
                       float* costs, void* workspace, int blank, float grad_scale);

/* Greedy CTC decoding (AM_training/decoder.py:186-201 GreedyDecoder.decode; call sites trainer_AAS.py:321-323,
 * AM_training/train.py:377-381): probs [T,N,C] scores (any monotone transform of the posteriors), d_sizes[N] valid frames
 * per utterance.  Per utterance: argmax per frame, drop blanks and frames equal to the previous FRAME's label.
 * out_labels / out_offsets [N,T] (first out_lens[n] entries valid; out_offsets may be NULL), out_lens [N]; all device. */
int aas_greedy_decode(aasStream_t stream, const float* probs, const int* d_sizes, int T, int N, int C, int blank,
                      int* out_labels, int* out_offsets, int* out_lens);
/* HOST function (no device work): Levenshtein distance of two int sequences - the python-Levenshtein C extension behind
 * Decoder.wer / Decoder.cer (AM_training/decoder.py:45-74).  Returns the distance, or -1 on bad arguments. */
int aas_edit_distance(const int* h_a, int na, const int* h_b, int nb);

/* ---------------------------------------------------------------- fp32-equivalent products on the bf16 pipes (precision 2) ----
 * Three-term operands: x = h + m + l exactly (h = rne_bf16(x), m = rne_bf16(x - h), l = rne_bf16(x - h - m): 24 significant bits).
 * The six products hh' + hm' + mh' + mm' + hl' + lh' (dropped terms <= 2^-25 |x y|: below fp32's rounding unit) are the
 * three-product plane kernels (aas_gemm_planes / aas_gemm_planes_tn) over two plane sets of each operand, both in the plane layout
 * of aas_split_planes:  set Q1 = (m | h)  ->  m m' + h m' + m h',   set Q2 = (h | l)  ->  h h' + l h' + h l';  the two partial
 * results add up (in the accumulators of one launch, or by a second, accumulating launch).  These entry points write both sets in one pass over the fp32 source.
 * `row_pitch_bytes` (0 = 4 Kp): bytes between the rows of each set.  With ONE buffer of pitch 8 Kp, set Q1 at its start and set Q2 at
 * byte 4 Kp of every row, the two sets are the two halves of a k-extent of 2 Kp: an NT product then takes ONE launch of
 * aas_gemm_planes with K = 2 Kp (both passes accumulate in registers); the row-major weight-gradient product (reduction over rows)
 * still takes two launches, on the two column halves. */
int aas_split_planes3(aasStream_t stream, const float* src, int64_t ld, int64_t rows, int K, int Kp, void* planes_q1, void* planes_q2,
                      int64_t row_pitch_bytes);
int aas_add3_planes3_f32(aasStream_t stream, float* out, const float* a, const float* b, const float* c, int64_t rows, int K, int Kp,
                         void* planes_q1, void* planes_q2, int64_t row_pitch_bytes);
/* transposed sets (aas_split_planes_t2's addressing): src[(t*tstride) + n*ld + c] -> planes[c][t*nbp + n] */
int aas_split_planes_t3(aasStream_t stream, const float* src, int64_t ld, int64_t tstride, int T, int nb, int nbp, int C, int64_t Kp,
                        void* planes_q1, void* planes_q2, int64_t row_pitch_bytes);

/* ---------------------------------------------------------------- optimiser -------------------
 * torch.optim.Adam(amsgrad=True/False) element-wise update (trainer_AAS.py:127-129,185-188;
 * AM_training/train.py:246-247), torch 2.x semantics:
 *   m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; vmax = max(vmax, v);
 *   p -= (lr/(1-b1^t)) * m / (sqrt(vmax)/sqrt(1-b2^t) + eps)          (vmax = v if !amsgrad) */
int aas_adam_f32(aasStream_t stream, float* p, const float* g, float* m, float* v, float* vmax,
                 int64_t n, float lr, float beta1, float beta2, float eps, int step, int amsgrad,
                 float grad_scale);
/* torch.optim.SGD(momentum, nesterov=True) element-wise update on a flat buffer (AM_training/train.py:172-174,247-249, `--optim sgd`):
 *   buf = momentum buf + g  (buf zero-initialised: the first step's "buf = g");  p -= lr (g + momentum buf).  No step-dependent
 * scalar, so the same launch serves the host-synchronous and the device-resident step. */
int aas_sgd_nesterov_f32(aasStream_t stream, float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum,
                         float grad_scale);

/* Same update with the step-dependent scalars read from device memory: d_hyper[0] = lr/(1-b1^t),
 * d_hyper[1] = sqrt(1-b2^t) - lets the whole training step be captured in a hipGraph and replayed. */
int aas_adam_dev_f32(aasStream_t stream, float* p, const float* g, float* m, float* v, float* vmax,
                     int64_t n, float beta1, float beta2, float eps, const float* d_hyper, int amsgrad,
                     float grad_scale);

/* d_step[0] += 1 (device-resident step counter, double) and d_hyper = {lr/(1-b1^t), sqrt(1-b2^t)} for aas_adam_dev_f32,
 * evaluated in double from double hyper-parameters like torch.optim.Adam's host-side bias corrections. */
int aas_adam_tick(aasStream_t stream, double* d_step, double lr, double beta1, double beta2, float* d_hyper);
/* BEGAN proportional controller on the device (trainer_AAS.py:190-194): kt <- clip(kt + lambda_k (gamma L_cl - L_ny_G), 0, 1)
 * from device-resident loss scalars; d_out6 = [L_ny_G, L_cl, L_ctc, kt, sum += L_ctc * n_batch, sum += n_batch] (the
 * last two feed the running CTC average of the log line, :169-170,198-203). */
int aas_began_step(aasStream_t stream, const float* d_l_adv_ny_G, const float* d_l_adv_cl, const float* d_l_ctc, double* d_kt,
                   double* d_out6, double gamma, double lambda_k, double n_batch);
/* The same controller from the step's RAW device sums, so that no scaling / summing launch sits between the loss kernels and it:
 * d_l1_sums[2] = sum|D(E(x)) - E(x)|, sum|D(c) - c| (aas_l1_fwd accumulators), d_ctc_costs[n_costs] = per-utterance CTC costs
 * (aas_ctc_loss_async); L_ny = scale_ny * sums[0] (scale = w_adversarial / nElement, model.py:30, trainer_AAS.py:147), L_cl likewise,
 * L_ctc = scale_ctc * sum(costs) (w_acoustic / N, :168); each rounded to fp32 as the tensors of the reference are. */
int aas_began_step_raw(aasStream_t stream, const double* d_l1_sums, double scale_ny, double scale_cl, const float* d_ctc_costs, int n_costs,
                       double scale_ctc, double* d_kt, double* d_out6, double gamma, double lambda_k, double n_batch);
/* The controller from THREE raw device sums (trainer_FSEGAN.py:175-179: the third logged loss is the DCE sum; data parallel: the sums
 * all-reduced over the ranks and the normalisers device scalars): L_0 = s0 f_0 d_l1_sums[0] (noisy adversarial), L_1 = s1 f_1 d_l1_sums[1]
 * (clean adversarial), L_2 = s2 f_2 d_third[0] (the third loss), f_i = d_scales3 ? d_scales3[i] : 1; n_batch from d_n_batch[0] when that is
 * not NULL.  d_out6 as aas_began_step. */
int aas_began_step_sums(aasStream_t stream, const double* d_l1_sums, const double* d_third, double s0, double s1, double s2,
                        const float* d_scales3, double* d_kt, double* d_out6, double gamma, double lambda_k, double n_batch,
                        const double* d_n_batch);
/* d_out3 = [d_l1_sums[0], d_l1_sums[1], sum(d_ctc_costs[0 .. n_costs))]: the three raw loss sums of an AAS step (trainer_AAS.py:146-177) in
 * one buffer - what a data-parallel step all-reduces before the controller (aas_began_step_sums).  d_l1_sums = NULL reads as zeros
 * (a step without a discriminator: AM_training/train.py:297-349, trainer_acoustic.py:120-142). */
int aas_loss_pack(aasStream_t stream, const double* d_l1_sums, const float* d_ctc_costs, int n_costs, double* d_out3);
/* d_out3 = [d_a2[0], d_a2[1], d_b1[0]] (a NULL source reads as zeros): raw fp64 loss sums that live in separate accumulators, in one
 * buffer for ONE all-reduce (FSEGAN: the two adversarial L1 sums and the DCE sum, trainer_FSEGAN.py:139-171; DCE: the one sum). */
int aas_sums_pack(aasStream_t stream, const double* d_a2, const double* d_b1, double* d_out3);
/* d_out[i] = (float)(weights[i] / d_counts[index[i]]), i < n <= 4: the loss normalisers (w_adversarial / nElement, w_acoustic / N:
 * model.py:30, trainer_AAS.py:147,168,177) from device-resident (all-reduced) counts, in one launch; weights / index are host arrays. */
int aas_scales_from_counts(aasStream_t stream, const double* d_counts, int n, const double* weights, const int* index, float* d_out);
/* Start of a training step in ONE launch: zero up to 8 device buffers (16-byte aligned; the flat gradient buffers = zero_grad_all,
 * trainer_AAS.py:134, and the loss accumulators) and, when rs != NULL, write the per-utterance weights of the batched
 * [enhanced; clean] discriminator pass: rs[0 .. n_neg) = -(float) d_kt[0] (the D-step factor, :156-160), rs[n_neg .. n_neg + n_one) = 1. */
int aas_step_prologue(aasStream_t stream, int n, void* const* bufs, const size_t* bytes, float* rs, int n_neg, int n_one, const double* d_kt);

/* ---------------------------------------------------------------- features --------------------
 * log-Mel filterbank: wave [N,S] -> out [N,n_mels,T] with T = 1 + S/hop; hamming(periodic) window
 * of `win` samples, centre=True reflect padding, |DFT|^2 -> mel -> log1p.  The caller supplies the constant
 * tables (aas_enhancement_amd/lmfb.py builds them): dft [win, 2*nbins] (cos | -sin, window folded in),
 * melT [nbins, n_mels].
 * (AM_training/train.py:39-42,55-60,199; model.py:194-198.) */
int aas_lmfb_fwd(aasStream_t stream, const float* wave, int N, int S, int win, int hop, int n_mels,
                 const float* dft, const float* melT, float* out);

/* The same features on the matrix cores, specialised for the reference's audio configuration (win = n_fft = 320,
 * hop = 160; AM_training/train.py:39-42) and for batches of utterances of DIFFERENT lengths: d_lens[n] (device, may be
 * NULL = S) valid samples of utterance n; frames t >= 1 + d_lens[n]/160 are written as zeros, reflect padding is at the
 * utterance's own ends.  Constant tables from the caller (aas_enhancement_amd/lmfb.py):
 *   tables     [2 (bf16 hi, lo)][4 segments][96 columns c][96 k = j] bf16: cos(2 pi j 2c/320) (j <= 80), cos(2 pi j (2c+1)/320)
 *              (j <= 79), sin(2 pi j 2c/320) (1 <= j <= 79), sin(2 pi j (2c+1)/320) (1 <= j <= 80), zero elsewhere - the
 *              twiddles of the twice-folded real DFT;
 *   window     [320] hamming (periodic);  mel_start / mel_cnt [n_mels] first bin and width of each triangular filter,
 *   mel_w      [n_mels][mel_maxw = 24] its weights. */
int aas_lmfb320_fwd(aasStream_t stream, const float* wave, const int* d_lens, int N, int S, int n_mels,
                    const void* tables, const float* window, const int* mel_start, const int* mel_cnt,
                    const float* mel_w, int mel_maxw, float* out);

#ifdef __cplusplus
}
#endif
#endif /* AAS_HIP_H */
