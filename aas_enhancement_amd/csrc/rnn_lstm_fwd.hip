#include "rnn_fwd32_kernel.h"

extern "C" size_t aas_rnn_sync_bytes(void) { return SYNC_BYTES; }
// the larger of: hi + lo arrays of the widest all-gathered vector (2*T*N rows x G*Hp bf16, Hp <= H + 15), and the
// BPTT reduce-scatter ring (2 slots x 2 directions x N rows x P consumers x P producers x 64 B)
extern "C" size_t aas_rnn_xchg_bytes(int T, int N, int H, int gates) {
    const size_t gather = (size_t)8 * T * N * ((size_t)gates * (H + 16) + 32);
    const size_t P = (size_t)(H + 15) / 16;                // 16-unit slices: the larger of the two ring shapes
    const size_t ring = (size_t)4 * N * P * P * 64;
    return (gather > ring ? gather : ring) + 8192;         // + the XCC table of the XCD-aware launches (rnn_split_kernel.h)
}

extern "C" int aas_lstm_fwd(aasStream_t stream, int T, int N, int H, const float* pre, const float* w_hh, const float* w_hh_rev, float* hout,
                            float* gact, float* cst, void* sync, void* xchg) {
    AAS_RNN_LAUNCH_LOCK();
    RnnP p = {};
    // (first: the one-shot setting is consumed by THIS call whatever happens next)
    if (aas_rnn_row_classes_take("aas_lstm_fwd", T, N, &p.cls_n, &p.cls_t0, &p.cls_t1)) return 1;
    AAS_CHECK(pre && w_hh && w_hh_rev && hout && gact && cst && sync, "aas_lstm_fwd: null pointer");
    p.T = T; p.N = N; p.H = H; p.pre = pre; p.w_hh = w_hh; p.w_hh_r = w_hh_rev; p.hout = hout; p.gact = gact; p.cst = cst;
    p.sync = (unsigned*)sync; p.xchg = (unsigned*)xchg;
    return run_fwd_any<LSTM_FWD>("aas_lstm_fwd", p, (hipStream_t)stream);
}

// The same launch with its parameters as an ARGUMENT (include/aas_hip.h: aasLaunch): row classes, CU budget, launch tag, kernel-
// selection bits and arithmetic mode come from *launch for this call only; launch->fwd_h_pitch receives what aas_rnn_last_fwd_h_pitch()
// would report.  Nothing process-wide is read for a field that is set, nothing process-wide is written.
extern "C" int aas_lstm_fwd_ex(aasStream_t stream, int T, int N, int H, const float* pre, const float* w_hh, const float* w_hh_rev, float* hout,
                               float* gact, float* cst, void* sync, void* xchg, aasLaunch* launch) {
    if (aas_scope_check(launch, "aas_lstm_fwd_ex")) return 1;
    AasScopeGuard guard(launch);
    return aas_lstm_fwd(stream, T, N, H, pre, w_hh, w_hh_rev, hout, gact, cst, sync, xchg);
}

// The forward launch WITH the layer's input projection inside (rnn_split_kernel.h: XF): x [T,N,I], W_ih / W_ih_rev [4H, I] instead of
// the pre-activation tensor.  -> 0 launched; 3 = this shape / mode / CU budget is not covered, NOTHING was launched or consumed: the
// caller forms `pre` with a GEMM and calls aas_lstm_fwd_ex.  Replaces the W_ih half of cuDNN's LSTM under
// Speech_enhancement_by_AAS/model.py:73-74,83 for the enhancement network's layers (N = 30 over the whole chip).
extern "C" int aas_lstm_fwd_x_ex(aasStream_t stream, int T, int N, int H, int I, const float* x, const float* w_ih, const float* w_ih_rev,
                                 const float* w_hh, const float* w_hh_rev, float* hout, float* gact, float* cst, void* sync, void* xchg,
                                 aasLaunch* launch) {
    if (aas_scope_check(launch, "aas_lstm_fwd_x_ex")) return 1;
    AasScopeGuard guard(launch);
    AAS_RNN_LAUNCH_LOCK();
    if (!split_xf_covers(T, N, H, I, xchg)) return 3;
    RnnP p = {};
    if (aas_rnn_row_classes_take("aas_lstm_fwd_x_ex", T, N, &p.cls_n, &p.cls_t0, &p.cls_t1)) return 1;
    AAS_CHECK(x && w_ih && w_ih_rev && w_hh && w_hh_rev && hout && gact && cst && sync, "aas_lstm_fwd_x_ex: null pointer");
    p.T = T; p.N = N; p.H = H; p.w_hh = w_hh; p.w_hh_r = w_hh_rev; p.hout = hout; p.gact = gact; p.cst = cst;
    p.xin = x; p.w_ih = w_ih; p.w_ih_r = w_ih_rev; p.I = I;
    p.sync = (unsigned*)sync; p.xchg = (unsigned*)xchg;
    aas_note_fwd_h_planes(0);
    const int rc = run_split<LSTM_FWD, true>("aas_lstm_fwd_x_ex", p, (hipStream_t)stream);
    AAS_CHECK(rc >= 0, "aas_lstm_fwd_x_ex: split_xf_covers() admitted a shape that run_split refused (T=%d N=%d H=%d I=%d)", T, N, H, I);
    return rc;
}
