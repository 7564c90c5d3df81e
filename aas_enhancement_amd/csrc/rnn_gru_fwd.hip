#include "rnn_fwd32_kernel.h"

extern "C" int aas_gru_fwd(aasStream_t stream, int T, int N, int H, const float* pre, const float* w_hh, const float* w_hh_rev, float* hout,
                           float* gact, void* sync, void* xchg) {
    AAS_RNN_LAUNCH_LOCK();
    RnnP p = {};
    // (first: the one-shot setting is consumed by THIS call whatever happens next)
    if (aas_rnn_row_classes_take("aas_gru_fwd", T, N, &p.cls_n, &p.cls_t0, &p.cls_t1)) return 1;
    AAS_CHECK(pre && w_hh && w_hh_rev && hout && gact && sync, "aas_gru_fwd: null pointer");
    p.T = T; p.N = N; p.H = H; p.pre = pre; p.w_hh = w_hh; p.w_hh_r = w_hh_rev; p.hout = hout; p.gact = gact;
    p.sync = (unsigned*)sync; p.xchg = (unsigned*)xchg;
    return run_fwd_any<GRU_FWD>("aas_gru_fwd", p, (hipStream_t)stream);
}

extern "C" int aas_gru_fwd_ex(aasStream_t stream, int T, int N, int H, const float* pre, const float* w_hh, const float* w_hh_rev, float* hout,
                              float* gact, void* sync, void* xchg, aasLaunch* launch) {
    if (aas_scope_check(launch, "aas_gru_fwd_ex")) return 1;
    AasScopeGuard guard(launch);
    return aas_gru_fwd(stream, T, N, H, pre, w_hh, w_hh_rev, hout, gact, sync, xchg);
}
