#include <stdarg.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include <stdlib.h>

#include "common.h"

static thread_local char g_err[512] = "";

void aas_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int aas_version(void) { return 1; }
extern "C" const char* aas_last_error(void) { return g_err; }
extern "C" int aas_device_cus(void) {
    static int cached[64] = {0};  // per device ordinal; queried once (keeps the launch path free of runtime queries)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (dev >= 0 && dev < 64 && cached[dev] > 0) return cached[dev];
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    if (dev >= 0 && dev < 64) cached[dev] = cus;
    return cus;
}

static int g_debug_flags = 0;
int aas_debug_flags_value() { return g_debug_flags; }
extern "C" int aas_set_debug_flags(int flags) {
    g_debug_flags = flags;
    return 0;
}
extern "C" int aas_get_debug_flags(void) { return g_debug_flags; }

namespace {
struct WsKey {
    int dev, kind;
    hipStream_t s;
    bool operator<(const WsKey& o) const { return dev != o.dev ? dev < o.dev : kind != o.kind ? kind < o.kind : s < o.s; }
};
struct WsBlock {
    void* p = nullptr;
    size_t bytes = 0;
};
std::mutex g_ws_mu;
std::map<WsKey, WsBlock> g_ws;
std::vector<std::pair<int, void*>> g_ws_retired;   // (device, block)
}  // namespace

void* aas_stream_workspace(int kind, hipStream_t s, size_t bytes, size_t floor_bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_ws_mu);
    WsBlock& w = g_ws[WsKey{dev, kind, s}];
    if (w.bytes < bytes) {
        // a stream under hipGraph capture can neither be synchronised nor allocate
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
        size_t want = bytes < floor_bytes ? floor_bytes : bytes;
        if (want < 2 * w.bytes) want = 2 * w.bytes;      // geometric growth: the retired blocks of a stream sum to less than its live one
        void* p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) return nullptr;
        if (w.p) g_ws_retired.emplace_back(dev, w.p);    // still referenced by queued launches / captured graphs: never freed here
        w.p = p;
        w.bytes = want;
    }
    return w.p;
}

extern "C" int aas_release_retired_workspaces(void) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    int prev = 0;
    (void)hipGetDevice(&prev);
    int n = 0;
    for (auto& e : g_ws_retired) {
        (void)hipSetDevice(e.first);
        (void)hipDeviceSynchronize();
        if (hipFree(e.second) == hipSuccess) ++n;
    }
    g_ws_retired.clear();
    (void)hipSetDevice(prev);
    return n;
}

namespace {
struct XchgState {
    size_t bytes = 0;         // whole buffer: two halves of bytes / 2
    unsigned parity = 0;      // half the NEXT launch works in
    size_t dirty[2] = {0, 0}; // bytes of each half that its last launch may have left un-poisoned
};
std::mutex g_xchg_mu;
std::map<void*, XchgState> g_xchg;
}  // namespace

extern "C" int aas_rnn_xchg_prepare(aasStream_t stream, void* xchg, size_t bytes) {
    AAS_CHECK(xchg != nullptr && (reinterpret_cast<uintptr_t>(xchg) & 255) == 0 && bytes >= 8192 && bytes % 512 == 0,
              "aas_rnn_xchg_prepare: a 256-byte aligned buffer of a multiple of 512 bytes");
    AAS_HIP(hipMemsetAsync(xchg, 0xFF, bytes, (hipStream_t)stream));
    std::lock_guard<std::mutex> lk(g_xchg_mu);
    XchgState st;
    st.bytes = bytes;
    g_xchg[xchg] = st;
    return 0;
}

extern "C" int aas_rnn_xchg_forget(void* xchg) {
    std::lock_guard<std::mutex> lk(g_xchg_mu);
    return g_xchg.erase(xchg) ? 0 : 1;
}

int aas_xchg_plan(void* xchg, size_t need_bytes, hipStream_t s, AasXchgPlan* out) {
    out->base = static_cast<unsigned*>(xchg);
    out->clean_ptr = nullptr;
    out->clean_words = 0;
    out->managed = 0;
    std::lock_guard<std::mutex> lk(g_xchg_mu);
    auto it = g_xchg.find(xchg);
    if (it == g_xchg.end()) return 0;
    XchgState& st = it->second;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    need_bytes = (need_bytes + 15) & ~(size_t)15;
    if (capturing || need_bytes > st.bytes / 2) {
        // a captured launch would replay with the parity of capture time, and a launch that does not fit a half takes the whole
        // buffer: either way the alternation ends here - the caller poisons the buffer itself from now on (correct, one launch more)
        g_xchg.erase(it);
        return 0;
    }
    const unsigned h = st.parity;
    char* b = static_cast<char*>(xchg);
    out->base = reinterpret_cast<unsigned*>(b + (size_t)h * (st.bytes / 2));
    out->clean_ptr = reinterpret_cast<unsigned*>(b + (size_t)(h ^ 1) * (st.bytes / 2));
    out->clean_words = (unsigned)(st.dirty[h ^ 1] / 4);
    out->managed = 1;
    st.dirty[h ^ 1] = 0;
    st.dirty[h] = need_bytes;
    st.parity = h ^ 1;
    return 0;
}

int aas_xchg_legacy_fill(void* xchg, size_t bytes, hipStream_t s) {
    {
        std::lock_guard<std::mutex> lk(g_xchg_mu);
        g_xchg.erase(xchg);
    }
    AAS_HIP(hipMemsetAsync(xchg, 0xFF, bytes, s));
    return 0;
}

extern "C" int aas_rnn_xchg_is_managed(void* xchg) {
    std::lock_guard<std::mutex> lk(g_xchg_mu);
    return g_xchg.count(xchg) ? 1 : 0;
}

bool aas_first_use_on_device(unsigned char* flags) {
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= AAS_MAX_DEV) return true;   // (unknown device: set the attribute every time)
    std::lock_guard<std::mutex> lk(mu);
    if (flags[dev]) return false;
    flags[dev] = 1;
    return true;
}

const char* aas_ablation_env(const char* name) {
    static const bool on = getenv("AAS_ABLATION") && atoi(getenv("AAS_ABLATION")) == 1;
    return on ? getenv(name) : nullptr;
}

// 0 (default) = fp32-input MFMA everywhere, the reference's arithmetic; 1 = split-bf16 (hi/lo, 3 MFMAs) fast mode
static int g_precision = 0;
int aas_precision_value() { return g_precision; }
extern "C" int aas_set_precision(int mode) {
    if (mode < 0 || mode > 2) {
        aas_set_error("aas_set_precision: mode must be 0 (fp32), 1 (split-bf16) or 2 (fp32-equivalent: fp32 recurrent products, six-product plane GEMMs)");
        return 1;
    }
    g_precision = mode;
    return 0;
}

// Upper bound on the CUs one persistent recurrent launch may occupy (0 = the whole device).  Two independent chains of
// recurrent launches on two streams (discriminator and acoustic branch of the AAS step) each take half the chip.
static int g_rnn_cu_limit = 0;
int aas_rnn_cus() {
    const int cus = aas_device_cus();
    return (g_rnn_cu_limit > 0 && g_rnn_cu_limit < cus) ? g_rnn_cu_limit : cus;
}
extern "C" int aas_set_rnn_cu_limit(int cus) {
    if (cus < 0) {
        aas_set_error("aas_set_rnn_cu_limit: negative limit");
        return 1;
    }
    g_rnn_cu_limit = cus;
    return 0;
}

// Identifies the persistent recurrent launches queued after the call: a launch that hits its bounded-spin timeout
// stores this value (>= 1) in the sticky error word of its sync buffer, so the host can name the layer.
static int g_rnn_tag = 1;
int aas_rnn_launch_tag_value() { return g_rnn_tag; }
static int g_wgrad_cap = 0;
int aas_wgrad_wg_cap() { return g_wgrad_cap; }
extern "C" int aas_set_wgrad_wg_cap(int workgroups) { g_wgrad_cap = workgroups < 0 ? 0 : workgroups; return 0; }
static int g_fwd_h_pitch = 0;
void aas_note_fwd_h_planes(int pitch_bytes) { g_fwd_h_pitch = pitch_bytes; }
extern "C" int aas_rnn_last_fwd_h_pitch(void) { return g_fwd_h_pitch; }
// Row classes of the NEXT forward recurrent launch (aas_lstm_fwd / aas_gru_fwd), consumed by it: batch rows [0, n_first) carry
// sequences of T_first frames, rows [n_first, N) of T_rest frames, inside a launch of T = max of the two.  What the batched
// discriminator pass over a noisy / clean pair of different padded lengths needs: the shorter class behaves exactly as in a launch
// of its own (zero state before its first frame in either direction, zero output and no gradient beyond its last).
static int g_cls_set = 0, g_cls_n = 0, g_cls_t0 = 0, g_cls_t1 = 0;
extern "C" int aas_set_rnn_row_classes(int n_first, int T_first, int T_rest) {
    if (n_first < 0 || T_first < 1 || T_rest < 1) {
        aas_set_error("aas_set_rnn_row_classes: n_first=%d T_first=%d T_rest=%d", n_first, T_first, T_rest);
        return 1;
    }
    g_cls_set = 1; g_cls_n = n_first; g_cls_t0 = T_first; g_cls_t1 = T_rest;
    return 0;
}
int aas_rnn_row_classes_take(const char* who, int T, int N, int* n, int* t0, int* t1) {
    if (!g_cls_set) { *n = N; *t0 = T; *t1 = T; return 0; }
    g_cls_set = 0;
    if (g_cls_n > N || g_cls_t0 > T || g_cls_t1 > T || (g_cls_t0 != T && g_cls_t1 != T)) {
        aas_set_error("%s: row classes (%d rows x %d frames, the rest x %d) do not fit a launch of N=%d T=%d", who, g_cls_n, g_cls_t0, g_cls_t1, N, T);
        return 1;
    }
    *n = g_cls_n; *t0 = g_cls_t0; *t1 = g_cls_t1;
    return 0;
}
extern "C" int aas_set_rnn_launch_tag(int tag) {
    if (tag < 1) {
        aas_set_error("aas_set_rnn_launch_tag: tag must be >= 1");
        return 1;
    }
    g_rnn_tag = tag;
    return 0;
}
