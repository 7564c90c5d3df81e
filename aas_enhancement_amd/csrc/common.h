// Shared host/device helpers for libaas_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/aas_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

void aas_set_error(const char* fmt, ...);
int aas_debug_flags_value();
// Experiment switches of the kernels are read from the environment only when AAS_ABLATION=1 is exported with them (the
// host side's aas_enhancement_amd/knobs.py applies the same rule): a stray AAS_* variable cannot change a product run.
const char* aas_ablation_env(const char* name);
int aas_precision_value();
int aas_wgrad_wg_cap();                       // 0 = no cap on the grid of aas_gemm_planes_tn
void aas_note_fwd_h_planes(int pitch_bytes);  // what the last forward recurrent launch left in its exchange buffer (0: nothing usable)
int aas_rnn_row_classes_take(const char* who, int T, int N, int* n, int* t0, int* t1);   // one-shot (aas_set_rnn_row_classes); default: all rows live for T
// Per-device "done once" flags (function attributes such as the raised dynamic-LDS limit belong to a device): true exactly once per
// device for a given flag array, thread-safe.  flags: a static unsigned char[AAS_MAX_DEV] of the call site.
constexpr int AAS_MAX_DEV = 64;
// Raise a kernel's dynamic-LDS limit once per device: the flag is set only AFTER hipFuncSetAttribute succeeded and the call holds a
// mutex across it, so a failed call is retried by the next launch and no second thread launches before the limit is raised.
// -> 0 ok (done now or earlier), 2 the attribute call failed (aas_last_error says why).
int aas_raise_dynamic_lds_once(unsigned char* flags, const void* kernel, int bytes);
int aas_rnn_row_classes_reject(const char* who);   // launches that cannot take row classes: consume + refuse a pending setting
// RAII: install an aasLaunch scope for this thread for the duration of one call (the *_ex entry points); nullptr = leave as is
struct AasScopeGuard {
    explicit AasScopeGuard(aasLaunch* l);
    ~AasScopeGuard();
    AasScopeGuard(const AasScopeGuard&) = delete;
    AasScopeGuard& operator=(const AasScopeGuard&) = delete;
  private:
    aasLaunch* prev_;
    bool on_;
};
int aas_scope_check(const aasLaunch* l, const char* who);
// Held by every persistent recurrent entry point from its first bookkeeping step (exchange-buffer plan, sync-buffer use) until its
// kernels are queued: with two host threads launching onto one stream / one managed exchange buffer, the order in which the buffer's
// halves were planned is then the order in which the launches sit in the stream.  Recursive: the *_ex forms call the plain ones.
std::recursive_mutex& aas_rnn_launch_mutex();
#define AAS_RNN_LAUNCH_LOCK() std::lock_guard<std::recursive_mutex> aas_rnn_lock__(aas_rnn_launch_mutex())
int aas_scope_gemm_max_steps();   // >= 0: the installed scope's lifetime cap of GEMM workgroups; -1: none installed / not set
int aas_rnn_launch_tag_value();  // >= 1: what a timed-out persistent launch leaves in its sticky error word
// gemm32.hip: the LDS-DMA fp32 GEMM; -> 0 launched, 1 error, -1 not applicable to these operands (take the general kernel)
int aas_gemm32_try(hipStream_t s, int mode, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                   int64_t ldc, const float* bias, const float* addend, int64_t ldd, int accumulate, int batch, int64_t strideA,
                   int64_t strideB, int64_t strideC, int kdivA, int64_t kouterA, int kdivB, int64_t kouterB, int nmulti,
                   const float* const* Am, const float* const* Bm, float* const* Cm, const int* Km, const float* d_alpha);
int aas_rnn_cus();  // aas_device_cus() capped by aas_set_rnn_cu_limit()
// Scratch block of at least `bytes` for launches on stream `s` of the current device, one per (device, stream, kind); grown on
// demand (at least doubling).  A block that is outgrown is RETIRED, not freed: a captured hipGraph may have its address baked into
// its nodes, and a queued launch may still read it - retired blocks are released by aas_release_retired_workspaces() or at exit.
// -> nullptr when the block would have to grow while `s` is under hipGraph capture (nothing may be allocated there).
// Managed exchange buffers of the persistent recurrent launches (aas_rnn_xchg_prepare): the buffer is two halves; a launch works in
// one half - poisoned by its predecessor - and, inside the kernel, re-poisons what its predecessor dirtied in the OTHER half, so no
// memset launch stands between two persistent launches.  -> managed = 0: the caller poisons the buffer itself (legacy path).
struct AasXchgPlan {
    unsigned* base;          // the half this launch works in
    unsigned* clean_ptr;     // 16-byte aligned region of the other half to poison from inside the kernel
    unsigned clean_words;    // (multiple of 4)
    int managed;
};
int aas_xchg_plan(void* xchg, size_t need_bytes, hipStream_t s, AasXchgPlan* out);
// the legacy path: poison `bytes` of the buffer with a memset launch; a managed buffer stops being managed (its halves no longer
// hold what the protocol assumes)
int aas_xchg_legacy_fill(void* xchg, size_t bytes, hipStream_t s);
enum { AAS_WS_GEMM_SLABS = 0, AAS_WS_BN_PARTIALS = 1 };
void* aas_stream_workspace(int kind, hipStream_t s, size_t bytes, size_t floor_bytes);

#define AAS_CHECK(cond, ...)            \
    do {                                \
        if (!(cond)) {                  \
            aas_set_error(__VA_ARGS__); \
            return 1;                   \
        }                               \
    } while (0)

#define AAS_LAUNCH_CHECK(name)                                              \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            aas_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return 2;                                                       \
        }                                                                   \
    } while (0)

#define AAS_HIP(call)                                                       \
    do {                                                                    \
        hipError_t e__ = (call);                                            \
        if (e__ != hipSuccess) {                                            \
            aas_set_error("%s failed: %s", #call, hipGetErrorString(e__));  \
            return 2;                                                       \
        }                                                                   \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// sigmoid / tanh on the hardware exp + reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each); absolute error ~1e-7,
// far inside the 1e-3 parity budget, and ~5x fewer instructions than ocml tanhf on the serial critical path
__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    const float e = __expf(-2.0f * fabsf(x));            // in (0, 1]: no overflow
    const float t = (1.0f - e) * __frcp_rn(1.0f + e);
    return copysignf(t, x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
