#include "rnn_bwd_rs_kernel.h"

extern "C" int aas_gru_bwd(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev,
                           const float* hout, const float* gact, float* dgx, float* dgh, void* sync, void* xchg) {
    AAS_RNN_LAUNCH_LOCK();
    AAS_CHECK(dy && w_hh && w_hh_rev && hout && gact && dgx && dgh && sync, "aas_gru_bwd: null pointer");
    RnnP p = {};
    p.T = T; p.N = N; p.H = H; p.dy = dy; p.w_hh = w_hh; p.w_hh_r = w_hh_rev; p.hout = (float*)hout; p.gact = (float*)gact;
    p.dg1 = dgh; p.dg2 = dgx; p.sync = (unsigned*)sync; p.xchg = (unsigned*)xchg;
    return run_bwd_any<GRU_BWD>("aas_gru_bwd", p, (hipStream_t)stream);
}

extern "C" int aas_gru_bwd_planes(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev,
                                  const float* hout, const float* gact, void* dgx_planes, void* dgh_planes, int Kp, void* sync, void* xchg) {
    AAS_RNN_LAUNCH_LOCK();
    AAS_CHECK(dy && w_hh && w_hh_rev && hout && gact && dgx_planes && dgh_planes && sync, "aas_gru_bwd_planes: null pointer");
    AAS_CHECK(Kp % 32 == 0 && Kp >= 6 * H && Kp < 6 * H + 32, "aas_gru_bwd_planes: Kp must be 2*3*H rounded up to 32 (got %d)", Kp);
    RnnP p = {};
    p.T = T; p.N = N; p.H = H; p.dy = dy; p.w_hh = w_hh; p.w_hh_r = w_hh_rev; p.hout = (float*)hout; p.gact = (float*)gact;
    p.dgp1 = (unsigned short*)dgh_planes; p.dgp2 = (unsigned short*)dgx_planes; p.dgKp = Kp;
    p.sync = (unsigned*)sync; p.xchg = (unsigned*)xchg;
    return run_bwd_any<GRU_BWD>("aas_gru_bwd_planes", p, (hipStream_t)stream);
}

extern "C" int aas_gru_bwd_ex(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev,
                              const float* hout, const float* gact, float* dgx, float* dgh, void* sync, void* xchg, aasLaunch* launch) {
    if (aas_scope_check(launch, "aas_gru_bwd_ex")) return 1;
    AasScopeGuard guard(launch);
    return aas_gru_bwd(stream, T, N, H, dy, w_hh, w_hh_rev, hout, gact, dgx, dgh, sync, xchg);
}
