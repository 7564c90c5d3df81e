// Vanilla tanh recurrence (nn.RNN, bias-free, bidirectional: the `rnn` entry of supported_rnns, model.py:12-17) on the
// counter-based persistent kernel of rnn_kernel.h (fp32-input MFMA in both precision modes).
#include "rnn_kernel.h"

extern "C" int aas_rnn_fwd(aasStream_t stream, int T, int N, int H, const float* pre, const float* w_hh, const float* w_hh_rev, float* hout,
                           float* gact, void* sync) {
    AAS_RNN_LAUNCH_LOCK();
    if (aas_rnn_row_classes_reject("aas_rnn_fwd")) return 1;     // (consumed and refused: they must not leak into the next lstm / gru launch)
    AAS_CHECK(pre && w_hh && w_hh_rev && hout && gact && sync, "aas_rnn_fwd: null pointer");
    RnnP p = {};
    p.T = T; p.N = N; p.H = H; p.pre = pre; p.w_hh = w_hh; p.w_hh_r = w_hh_rev; p.hout = hout; p.gact = gact;
    p.sync = (unsigned*)sync;
    return run<RNN_FWD>("aas_rnn_fwd", p, (hipStream_t)stream);
}

extern "C" int aas_rnn_bwd(aasStream_t stream, int T, int N, int H, const float* dy, const float* w_hh, const float* w_hh_rev,
                           const float* gact, float* dpre, void* sync) {
    AAS_RNN_LAUNCH_LOCK();
    AAS_CHECK(dy && w_hh && w_hh_rev && gact && dpre && sync, "aas_rnn_bwd: null pointer");
    RnnP p = {};
    p.T = T; p.N = N; p.H = H; p.dy = dy; p.w_hh = w_hh; p.w_hh_r = w_hh_rev; p.gact = (float*)gact; p.dg1 = dpre;
    p.sync = (unsigned*)sync;
    return run<RNN_BWD>("aas_rnn_bwd", p, (hipStream_t)stream);
}
