#include "rnn_bwd_rs_kernel.h"

extern "C" int aas_gru_bwd(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev,
                           const float* hout, const float* gact, float* dgx, float* dgh, void* sync, void* xchg) {
    AAS_CHECK(dy && w_hh && w_hh_rev && hout && gact && dgx && dgh && sync, "aas_gru_bwd: null pointer");
    RnnP p = {};
    p.T = T; p.N = N; p.H = H; p.dy = dy; p.w_hh = w_hh; p.w_hh_r = w_hh_rev; p.hout = (float*)hout; p.gact = (float*)gact;
    p.dg1 = dgh; p.dg2 = dgx; p.sync = (unsigned*)sync; p.xchg = (unsigned*)xchg;
    return run_bwd_any<GRU_BWD>("aas_gru_bwd", p, (hipStream_t)stream);
}
