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
