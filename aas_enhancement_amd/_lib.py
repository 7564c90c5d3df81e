"""ctypes binding of libaas_hip.so (the C ABI declared in include/aas_hip.h).

PyTorch is plumbing here: it owns device memory (``tensor.data_ptr()``) and the HIP stream
(``torch.cuda.current_stream().cuda_stream``); every compute op of the hot path is a call into the
HIP library.  There is NO fallback: if the library is missing or a call fails, a RuntimeError is
raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libaas_hip.so")

_lib = None

c_int, c_i64, c_f32, c_vp, c_sz = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "aas_version": [],
    "aas_last_error": [],
    "aas_device_cus": [],
    "aas_set_debug_flags": [c_int],
    "aas_get_debug_flags": [],
    "aas_get_gemm_max_steps": [],
    "aas_rnn_xchg_prepare": [c_vp, c_vp, c_sz],
    "aas_rnn_xchg_forget": [c_vp],
    "aas_rnn_xchg_is_managed": [c_vp],
    "aas_release_retired_workspaces": [],
    "aas_launch_scope": [c_vp, c_vp],
    "aas_lstm_fwd_ex": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "aas_lstm_fwd_x_ex": [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "aas_lstm_bwd_ex": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "aas_gru_fwd_ex": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "aas_gru_bwd_ex": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "aas_set_precision": [c_int],
    "aas_set_rnn_launch_tag": [c_int],
    "aas_set_rnn_row_classes": [c_int, c_int, c_int],
    "aas_set_rnn_cu_limit": [c_int],
    "aas_gemm_f32": [c_vp, c_int, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_int,
                     c_int, c_i64, c_i64, c_i64, c_int, c_i64, c_int, c_i64],
    "aas_gemm_f32_multi": [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_i64, c_i64, c_vp],
    "aas_set_gemm_variant": [c_int],
    "aas_set_gemm_max_steps": [c_int],
    "aas_gemm_tn_rowscaled_f32": [c_vp, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_int],
    "aas_gemm_planes": [c_vp, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_int,
                        c_int, c_i64, c_i64, c_i64],
    "aas_gemm_planes_multi": [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64],
    "aas_rnn_last_fwd_h_pitch": [],
    "aas_set_wgrad_wg_cap": [c_int],
    "aas_gemm_planes_tn": [c_vp, c_int] + [c_vp] * 18 + [c_int, c_int, c_vp, c_int],
    "aas_split_planes": [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_int],
    "aas_lstm_bwd_planes3": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp],
    "aas_split_planes3": [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_i64],
    "aas_add3_planes3_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_i64],
    "aas_split_planes_t3": [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_int, c_i64, c_vp, c_vp, c_i64],
    "aas_split_planes_t": [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_i64, c_vp, c_vp],
    "aas_split_planes_t2": [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_int, c_i64, c_vp, c_vp],
    "aas_transpose_f32": [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64],
    "aas_transpose_add_f32": [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64],
    "aas_swap01_f32": [c_vp, c_vp, c_vp, c_int, c_int, c_int],
    "aas_add3_planes_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp],
    "aas_add3_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64],
    "aas_scale_rows_f32": [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int],
    "aas_axpby_f32": [c_vp, c_vp, c_vp, c_f32, c_f32, c_i64],
    "aas_leaky_relu_f32": [c_vp, c_vp, c_vp, c_vp, c_f32, c_i64],
    "aas_step_prologue": [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp],
    "aas_began_step_raw": [c_vp, c_vp, ctypes.c_double, ctypes.c_double, c_vp, c_int, ctypes.c_double, c_vp, c_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double],
    "aas_began_step_sums": [c_vp, c_vp, c_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_vp, c_vp, c_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_vp],
    "aas_loss_pack": [c_vp, c_vp, c_vp, c_int, c_vp],
    "aas_sums_pack": [c_vp, c_vp, c_vp, c_vp],
    "aas_scales_from_counts": [c_vp, c_vp, c_int, c_vp, c_vp, c_vp],
    "aas_scale_dev_f32": [c_vp, c_vp, c_vp, c_vp, c_f32, c_i64],
    "aas_colsum_f32": [c_vp, c_vp, c_i64, c_int, c_i64, c_vp, c_int],
    "aas_sqsum_f32": [c_vp, c_vp, c_i64, c_vp],
    "aas_rnn_sync_bytes": [],
    "aas_rnn_xchg_bytes": [c_int, c_int, c_int, c_int],
    "aas_lstm_fwd": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "aas_lstm_bwd": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "aas_lstm_bwd_planes": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp],
    "aas_gru_bwd_planes": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp],
    "aas_planes_transpose": [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_i64, c_vp, c_vp],
    "aas_gru_fwd": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "aas_gru_bwd": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "aas_rnn_fwd": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "aas_rnn_bwd": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "aas_bn_fwd": [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp],
    "aas_bn_bwd": [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_int, c_vp],
    "aas_bn_stats": [c_vp, c_vp, c_i64, c_int, c_vp],
    "aas_bn_apply": [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp],
    "aas_bn_bwd_reduce": [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_f32, c_vp, c_vp],
    "aas_bn_bwd_apply": [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp],
    "aas_bn_eval": [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32],
    "aas_softmax_rows": [c_vp, c_vp, c_vp, c_i64, c_int],
    "aas_col2im_f32": [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int],
    "aas_l1_fwd": [c_vp, c_vp, c_vp, c_i64, c_vp],
    "aas_l1_bwd": [c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_vp, c_int],
    "aas_l1_fwd2d": [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_vp],
    "aas_l1_bwd2d": [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_f32, c_vp, c_vp, c_i64, c_int, c_vp, c_i64],
    "aas_ctc_get_workspace_size": [c_vp, c_vp, c_int, c_int, c_int, ctypes.POINTER(c_sz)],
    "aas_compute_ctc_loss": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_int],
    "aas_ctc_loss_async": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_f32],
    "aas_greedy_decode": [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp],
    "aas_edit_distance": [c_vp, c_int, c_vp, c_int],
    "aas_adam_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_int, c_int, c_f32],
    "aas_sgd_nesterov_f32": [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32],
    "aas_adam_dev_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_vp, c_int, c_f32],
    "aas_adam_tick": [c_vp, c_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_vp],
    "aas_began_step": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double],
    "aas_lmfb320_fwd": [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp],
    "aas_lmfb_fwd": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp],
}
_RESTYPES = {"aas_last_error": ctypes.c_char_p, "aas_rnn_sync_bytes": c_sz, "aas_rnn_xchg_bytes": c_sz}


def lib():
    """Load (once) and return the ctypes handle; raises loudly when the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "aas_enhancement_amd: HIP library %s is missing - build it with "
                "`python -m aas_enhancement_amd.build` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, c_int)
        from . import knobs                               # ablation switches: defaults unless AAS_ABLATION=1 (knobs.py)
        if knobs.get("WGRAD_WGS"):                        # grid cap of the row-major weight-gradient GEMM (0 = none)
            L.aas_set_wgrad_wg_cap(int(knobs.get("WGRAD_WGS")))
        if knobs.get("DEBUG_FLAGS"):                      # A/B kernel-selection bits of aas_set_debug_flags (include/aas_hip.h)
            L.aas_set_debug_flags(int(knobs.get("DEBUG_FLAGS")))
        # The host side above this binding is the TRAINING STEP: its GEMMs run beside persistent recurrent launches that become
        # resident only when enough CUs are free at once, so a GEMM workgroup lives at most 48 k-steps (~0.1 ms) here - deeper
        # products are split further along K.  Same box, config-2 fp32 step: no cap 28.6 ms, 48: 27.55, 24: 27.7.  (The C library's
        # own default is no cap: the best choice for a product alone on the chip.)
        L.aas_set_gemm_max_steps(int(knobs.get("GEMM32_MAXSTEPS")))
        if os.environ.get("AAS_PRECISION") in ("0", "1", "2"):  # 0 = fp32 MFMA (library default), 1 = split-bf16 fast mode
            L.aas_set_precision(int(os.environ["AAS_PRECISION"]))
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().aas_last_error()
        raise RuntimeError("libaas_hip %s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a CUDA float tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("aas_enhancement_amd ops run on the MI355X only: got a %s tensor (no CPU fallback)" % t.device)
