#include <stdarg.h>

#include <atomic>
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

// ---- queue-time settings --------------------------------------------------------------------------------------------------------
// Two levels.  (1) PROCESS settings (aas_set_precision, aas_set_debug_flags, aas_set_rnn_cu_limit, aas_set_rnn_launch_tag,
// aas_set_gemm_max_steps, aas_set_wgrad_wg_cap): atomics, read when a launch is queued - fine for a single-threaded host.
// (2) A LAUNCH SCOPE (aasLaunch, include/aas_hip.h): a caller-owned struct installed for the calling THREAD (aas_launch_scope) or
// passed to one call (the *_ex entry points); every field that is set in it overrides the process setting for the launches this
// thread queues while it is installed.  Nothing one thread sets through a scope is visible to another thread's launches.
static thread_local aasLaunch* t_scope = nullptr;
static std::atomic<int> g_debug_flags{0};
int aas_debug_flags_value() { return (t_scope && t_scope->debug_flags >= 0) ? t_scope->debug_flags : g_debug_flags.load(std::memory_order_relaxed); }
extern "C" int aas_set_debug_flags(int flags) {
    g_debug_flags.store(flags, std::memory_order_relaxed);
    return 0;
}
extern "C" int aas_get_debug_flags(void) { return aas_debug_flags_value(); }
extern "C" int aas_launch_scope(aasLaunch* l, aasLaunch** prev) {
    if (l != nullptr && l->size != (int)sizeof(aasLaunch)) {
        aas_set_error("aas_launch_scope: aasLaunch.size = %d, this library's is %d", l->size, (int)sizeof(aasLaunch));
        return 1;
    }
    if (prev) *prev = t_scope;
    t_scope = l;
    return 0;
}
AasScopeGuard::AasScopeGuard(aasLaunch* l) : prev_(t_scope), on_(l != nullptr) { if (on_) t_scope = l; }
AasScopeGuard::~AasScopeGuard() { if (on_) t_scope = prev_; }
int aas_scope_check(const aasLaunch* l, const char* who) {
    if (l != nullptr && l->size != (int)sizeof(aasLaunch)) {
        aas_set_error("%s: aasLaunch.size = %d, this library's is %d", who, l->size, (int)sizeof(aasLaunch));
        return 1;
    }
    return 0;
}
std::recursive_mutex& aas_rnn_launch_mutex() {
    static std::recursive_mutex mu;
    return mu;
}
int aas_scope_gemm_max_steps() { return (t_scope && t_scope->gemm_max_steps >= 0) ? t_scope->gemm_max_steps : -1; }

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

int aas_raise_dynamic_lds_once(unsigned char* flags, const void* kernel, int bytes) {
    static std::mutex mu;
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < AAS_MAX_DEV;   // (unknown device: set the attribute every time)
    std::lock_guard<std::mutex> lk(mu);       // held across the attribute call: a second thread launches only after the limit is raised
    if (known && flags[dev]) return 0;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        aas_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed: %s", bytes, hipGetErrorString(e));
        return 2;                             // (not marked done: the next launch tries again)
    }
    if (known) flags[dev] = 1;
    return 0;
}

const char* aas_ablation_env(const char* name) {
    static const bool on = getenv("AAS_ABLATION") && atoi(getenv("AAS_ABLATION")) == 1;
    return on ? getenv(name) : nullptr;
}

// 0 (default) = fp32-input MFMA everywhere, the reference's arithmetic; 1 = split-bf16 (hi/lo, 3 MFMAs) fast mode
static std::atomic<int> g_precision{0};
int aas_precision_value() { return (t_scope && t_scope->precision >= 0) ? t_scope->precision : g_precision.load(std::memory_order_relaxed); }
extern "C" int aas_set_precision(int mode) {
    if (mode < 0 || mode > 2) {
        aas_set_error("aas_set_precision: mode must be 0 (fp32), 1 (split-bf16) or 2 (fp32-equivalent: fp32 recurrent products, six-product plane GEMMs)");
        return 1;
    }
    g_precision.store(mode, std::memory_order_relaxed);
    return 0;
}

// Upper bound on the CUs one persistent recurrent launch may occupy (0 = the whole device).  Two independent chains of
// recurrent launches on two streams (discriminator and acoustic branch of the AAS step) each take half the chip.
static std::atomic<int> g_rnn_cu_limit{0};
int aas_rnn_cus() {
    const int cus = aas_device_cus();
    const int lim = (t_scope && t_scope->rnn_cu_limit >= 0) ? t_scope->rnn_cu_limit : g_rnn_cu_limit.load(std::memory_order_relaxed);
    return (lim > 0 && lim < cus) ? lim : cus;
}
extern "C" int aas_set_rnn_cu_limit(int cus) {
    if (cus < 0) {
        aas_set_error("aas_set_rnn_cu_limit: negative limit");
        return 1;
    }
    g_rnn_cu_limit.store(cus, std::memory_order_relaxed);
    return 0;
}

// Identifies the persistent recurrent launches queued after the call: a launch that hits its bounded-spin timeout
// stores this value (>= 1) in the sticky error word of its sync buffer, so the host can name the layer.
static std::atomic<int> g_rnn_tag{1};
int aas_rnn_launch_tag_value() { return (t_scope && t_scope->rnn_tag >= 1) ? t_scope->rnn_tag : g_rnn_tag.load(std::memory_order_relaxed); }
static std::atomic<int> g_wgrad_cap{0};
int aas_wgrad_wg_cap() { return (t_scope && t_scope->wgrad_wg_cap >= 0) ? t_scope->wgrad_wg_cap : g_wgrad_cap.load(std::memory_order_relaxed); }
extern "C" int aas_set_wgrad_wg_cap(int workgroups) { g_wgrad_cap.store(workgroups < 0 ? 0 : workgroups, std::memory_order_relaxed); return 0; }
// What the last forward recurrent launch OF THIS THREAD left in its exchange buffer (per thread: the launch and the query that follows
// it come from the same host thread); with a launch scope installed the value goes into the scope's fwd_h_pitch as well.
static thread_local int t_fwd_h_pitch = 0;
void aas_note_fwd_h_planes(int pitch_bytes) {
    t_fwd_h_pitch = pitch_bytes;
    if (t_scope) t_scope->fwd_h_pitch = pitch_bytes;
}
extern "C" int aas_rnn_last_fwd_h_pitch(void) { return t_fwd_h_pitch; }
// Row classes of the NEXT forward recurrent launch of THIS THREAD (aas_lstm_fwd / aas_gru_fwd), consumed by it: batch rows [0, n_first)
// carry sequences of T_first frames, rows [n_first, N) of T_rest frames, inside a launch of T = max of the two.  What the batched
// discriminator pass over a noisy / clean pair of different padded lengths needs: the shorter class behaves exactly as in a launch
// of its own (zero state before its first frame in either direction, zero output and no gradient beyond its last).  Per thread
// (a launch queued by another host thread cannot consume them); a launch scope / the *_ex entry points carry them as fields instead.
static thread_local int t_cls_set = 0, t_cls_n = 0, t_cls_t0 = 0, t_cls_t1 = 0;
extern "C" int aas_set_rnn_row_classes(int n_first, int T_first, int T_rest) {
    if (n_first < 0 || T_first < 1 || T_rest < 1) {
        aas_set_error("aas_set_rnn_row_classes: n_first=%d T_first=%d T_rest=%d", n_first, T_first, T_rest);
        return 1;
    }
    t_cls_set = 1; t_cls_n = n_first; t_cls_t0 = T_first; t_cls_t1 = T_rest;
    return 0;
}
int aas_rnn_row_classes_take(const char* who, int T, int N, int* n, int* t0, int* t1) {
    int cn, c0, c1;
    if (t_scope && t_scope->cls_n_first >= 0) {          // the scope's classes win; they are consumed like the one-shot setting
        cn = t_scope->cls_n_first; c0 = t_scope->cls_T_first; c1 = t_scope->cls_T_rest;
        t_scope->cls_n_first = -1;
        t_cls_set = 0;
    } else if (t_cls_set) {
        cn = t_cls_n; c0 = t_cls_t0; c1 = t_cls_t1;
        t_cls_set = 0;
    } else {
        *n = N; *t0 = T; *t1 = T;
        return 0;
    }
    if (cn > N || c0 < 1 || c1 < 1 || c0 > T || c1 > T || (c0 != T && c1 != T)) {
        aas_set_error("%s: row classes (%d rows x %d frames, the rest x %d) do not fit a launch of N=%d T=%d", who, cn, c0, c1, N, T);
        return 1;
    }
    *n = cn; *t0 = c0; *t1 = c1;
    return 0;
}
int aas_rnn_row_classes_reject(const char* who) {
    const bool pending = t_cls_set || (t_scope && t_scope->cls_n_first >= 0);
    t_cls_set = 0;
    if (t_scope) t_scope->cls_n_first = -1;
    if (pending) {
        aas_set_error("%s: row classes are pending for this launch, which does not support them (lstm / gru forward launches only)", who);
        return 1;
    }
    return 0;
}
extern "C" int aas_set_rnn_launch_tag(int tag) {
    if (tag < 1) {
        aas_set_error("aas_set_rnn_launch_tag: tag must be >= 1");
        return 1;
    }
    g_rnn_tag.store(tag, std::memory_order_relaxed);
    return 0;
}
