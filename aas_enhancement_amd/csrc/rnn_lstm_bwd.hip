#include "rnn_bwd_rs_kernel.h"

extern "C" int aas_lstm_bwd(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev,
                            const float* gact, const float* cst, float* dgates, void* sync, void* xchg) {
    AAS_RNN_LAUNCH_LOCK();
    AAS_CHECK(dy && w_hh && w_hh_rev && gact && cst && dgates && sync, "aas_lstm_bwd: null pointer");
    RnnP p = {};
    p.T = T; p.N = N; p.H = H; p.dy = dy; p.w_hh = w_hh; p.w_hh_r = w_hh_rev; p.gact = (float*)gact; p.cst = (float*)cst; p.dg1 = dgates;
    p.sync = (unsigned*)sync; p.xchg = (unsigned*)xchg;
    return run_bwd_any<LSTM_BWD>("aas_lstm_bwd", p, (hipStream_t)stream);
}


extern "C" int aas_lstm_bwd_planes3(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev,
                                    const float* gact, const float* cst, void* dgates_sets, int Kp, void* sync, void* xchg) {
    AAS_RNN_LAUNCH_LOCK();
    AAS_CHECK(dy && w_hh && w_hh_rev && gact && cst && dgates_sets && sync, "aas_lstm_bwd_planes3: null pointer");
    AAS_CHECK(Kp % 32 == 0 && Kp >= 8 * H && Kp < 8 * H + 32, "aas_lstm_bwd_planes3: Kp must be 2*4*H rounded up to 32 (got %d)", Kp);
    RnnP p = {};
    p.T = T; p.N = N; p.H = H; p.dy = dy; p.w_hh = w_hh; p.w_hh_r = w_hh_rev; p.gact = (float*)gact; p.cst = (float*)cst;
    p.dgp1 = (unsigned short*)dgates_sets; p.dgKp = Kp; p.dgsets = 3;
    p.sync = (unsigned*)sync; p.xchg = (unsigned*)xchg;
    return run_bwd_any<LSTM_BWD>("aas_lstm_bwd_planes3", p, (hipStream_t)stream);
}

extern "C" int aas_lstm_bwd_planes(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev,
                                   const float* gact, const float* cst, void* dgates_planes, int Kp, void* sync, void* xchg) {
    AAS_RNN_LAUNCH_LOCK();
    AAS_CHECK(dy && w_hh && w_hh_rev && gact && cst && dgates_planes && sync, "aas_lstm_bwd_planes: null pointer");
    AAS_CHECK(Kp % 32 == 0 && Kp >= 8 * H && Kp < 8 * H + 32, "aas_lstm_bwd_planes: Kp must be 2*4*H rounded up to 32 (got %d)", Kp);
    RnnP p = {};
    p.T = T; p.N = N; p.H = H; p.dy = dy; p.w_hh = w_hh; p.w_hh_r = w_hh_rev; p.gact = (float*)gact; p.cst = (float*)cst;
    p.dgp1 = (unsigned short*)dgates_planes; p.dgKp = Kp;
    p.sync = (unsigned*)sync; p.xchg = (unsigned*)xchg;
    return run_bwd_any<LSTM_BWD>("aas_lstm_bwd_planes", p, (hipStream_t)stream);
}

extern "C" int aas_lstm_bwd_ex(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev,
                               const float* gact, const float* cst, float* dgates, void* sync, void* xchg, aasLaunch* launch) {
    if (aas_scope_check(launch, "aas_lstm_bwd_ex")) return 1;
    AasScopeGuard guard(launch);
    return aas_lstm_bwd(stream, T, N, H, dy, w_hh, w_hh_rev, gact, cst, dgates, sync, xchg);
}
