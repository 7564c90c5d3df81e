"""Layer-level forward/backward ops of the AAS hot path, each a thin host wrapper over the C ABI of
libaas_hip.so, plus the ``torch.autograd.Function`` shells that let the reference-compatible
modules (model.py) participate in autograd.  All tensors are fp32 CUDA (HIP) tensors.

Layouts used internally:  sequences [T,N,H] row-major ("TNH"); conv front-end
channels-last [N,T,C]; the module boundary keeps the reference's [N,C,T].
"""

import torch

from . import knobs
from ._lib import check, lib, ptr, require_cuda, stream

NT, NN, TN = 0, 1, 2

_scratch = {}
_scratch_lock = __import__("threading").RLock()     # two trainers in two host threads may size the same stream's scratch at once


def _locked(fn):
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        with _scratch_lock:
            return fn(*a, **k)
    return wrapped


@_locked
def _sync_buf(dev):
    """zero-initialised arrival-counter scratch for the persistent RNN kernels (per device+stream)."""
    key = ("sync", dev, torch.cuda.current_stream().cuda_stream)
    b = _scratch.get(key)
    if b is None:
        b = torch.zeros(int(lib().aas_rnn_sync_bytes()), dtype=torch.uint8, device=dev)
        _scratch[key] = b
    return b


_layer_names = {}   # launch tag // 2 -> layer name (tag 1 = a launch of an unregistered layer)


def register_layer(name):
    """-> id of a recurrent layer; its launches carry tag 2*id (forward) / 2*id+1 (BPTT) into the sticky error word."""
    lid = len(_layer_names) + 1
    _layer_names[lid] = name
    return lid


def name_layers(module, prefix):
    """Give every recurrent layer of `module` a readable name ("G.rnn2.rnn") for timeout diagnostics."""
    for n, m in module.named_modules():
        lid = getattr(m, "_aas_layer_id", None)
        if lid is not None:
            _layer_names[lid] = prefix + "." + n


def rnn_timeout_layers():
    """Names of the persistent launches whose bounded spins timed out since the flags were last cleared (host sync)."""
    out = []
    for k, b in _scratch.items():
        if k[0] == "sync":
            tag = int(b.view(torch.int32)[1024].item())
            if tag:
                out.append("%s %s" % (_layer_names.get(tag // 2, "unregistered layer"), "BPTT" if tag & 1 else "forward"))
    return out


def rnn_timeout_flag(dev=None):
    """True if any persistent RNN launch on this device hit its bounded-spin timeout."""
    return len(rnn_timeout_layers()) > 0


def clear_rnn_timeout():
    """Lower the sticky word after a reported exchange timeout.  The managed exchange buffers are handed back too: a launch that gave
    up may have left its half in any state; they are poison-filled afresh at their next use (`_xchg_buf`)."""
    for k, b in _scratch.items():
        if k[0] == "sync":
            b.view(torch.int32)[1024:1025].zero_()
        elif k[0] in ("xchg_fwd", "xchg_bwd"):
            lib().aas_rnn_xchg_forget(ptr(b))
            _xchg_meta.pop(k, None)        # (not a fallback of the library's: prepare afresh at the next use)


def check_rnn_health(scalars=()):
    """Call at a host synchronisation point.  A raised sticky word names the layer whose exchange timed out - its
    gradients are garbage and training must not continue; a non-finite loss without it is reported as divergence."""
    import math
    # (a NaN that training itself produced is published as a quiet-NaN bit pattern, never as the exchange's poison word
    #  0xFFFFFFFF / tag 3, so it cannot raise the sticky word: the word means a launch really waited ~0.5 s in vain)
    bad = rnn_timeout_layers()
    if bad:
        raise RuntimeError("persistent recurrent kernel: cross-CU exchange timed out in %s (results of this step are invalid)"
                           % ", ".join(bad))
    if any(not math.isfinite(float(v)) for v in scalars):
        raise FloatingPointError("training diverged: non-finite loss scalars %r" % (list(scalars),))


@_locked
def _xchg_buf(dev, T, N, H, G, kind="any"):
    """Exchange scratch of the persistent RNN kernels, per device + stream (+ kind), grown on demand.
    kind "fwd" / "bwd": a MANAGED buffer (include/aas_hip.h: aas_rnn_xchg_prepare) - twice the size a launch needs, poison-filled
    once here; the library then alternates the launches between its halves and the kernels re-poison behind themselves, so no memset
    launch precedes a persistent launch.  Forward and BPTT launches keep separate buffers (their exchange protocols differ).  When the
    library had to fall back on a buffer (hipGraph capture, an A/B kernel that takes the whole buffer) it is prepared again here."""
    if kind == "any":
        need = int(lib().aas_rnn_xchg_bytes(T, N, H, G))
        key = ("xchg", dev, torch.cuda.current_stream().cuda_stream)
        b = _scratch.get(key)
        if b is None or b.numel() < need:
            b = torch.empty(need, dtype=torch.uint8, device=dev)
            _scratch[key] = b
        return b
    if kind == "fwd":       # rows [2][T][N] of ceil(Hp / 32) 128-byte chunks (the legacy form is what sizes the buffer) + the XCC table
        need = 2 * T * N * ((H + 31) // 32 + 1) * 128 + 8192
    else:
        need = int(lib().aas_rnn_xchg_bytes(T, N, H, G))
    need = (2 * need + 511) // 512 * 512
    key = ("xchg_" + kind, dev, torch.cuda.current_stream().cuda_stream)
    b = _scratch.get(key)
    grown = b is None or b.numel() < need
    if grown:
        if b is not None:
            lib().aas_rnn_xchg_forget(ptr(b))
        b = torch.empty(need, dtype=torch.uint8, device=dev)
        _scratch[key] = b
        _xchg_meta[key] = dict(last=None, fell=set())
    meta = _xchg_meta.setdefault(key, dict(last=None, fell=set()))
    managed = bool(lib().aas_rnn_xchg_is_managed(ptr(b)))
    shape = (T, N, H, G, state().rnn_cu_limit)
    if not grown and not managed and meta["last"] is not None:
        # the previous launch on this buffer ended its management: the library fell back to its own poison fill for that shape (a
        # launch of fewer than 4 steps, a batch split over several launches, ...).  Preparing the buffer again for the same shape
        # would only add a whole-buffer memset in front of the library's own: hand it out as it is until the shape changes.
        meta["fell"].add(meta["last"])
    if (grown or not managed) and shape not in meta["fell"] and not torch.cuda.is_current_stream_capturing():
        check(lib().aas_rnn_xchg_prepare(stream(), ptr(b), b.numel()), "aas_rnn_xchg_prepare")
    meta["last"] = shape
    return b


_xchg_meta = {}


@_locked
def _wsd(dev, n):
    key = ("wsd", dev, torch.cuda.current_stream().cuda_stream)
    b = _scratch.get(key)
    if b is None or b.numel() < n:
        b = torch.empty(max(n, 4096), dtype=torch.float64, device=dev)
        _scratch[key] = b
    return b


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ---- launch state: the queue-time settings of ONE trainer (or of the process, when none is installed) --------------------------------
import ctypes as _ct
import threading as _threading


class _CLaunch(_ct.Structure):
    """include/aas_hip.h: aasLaunch"""
    _fields_ = [(n, _ct.c_int) for n in ("size", "precision", "debug_flags", "rnn_cu_limit", "rnn_tag", "gemm_max_steps", "wgrad_wg_cap",
                                        "cls_n_first", "cls_T_first", "cls_T_rest", "fwd_h_pitch")]


def _new_claunch():
    return _CLaunch(_ct.sizeof(_CLaunch), -1, -1, -1, 0, -1, -1, -1, 0, 0, 0)


class LaunchState(object):
    """Everything the library and this module read when a launch is QUEUED, owned by one trainer: the arithmetic mode, the lifetime
    cap of GEMM workgroups, the kernel-selection (debug) bits, the CU budget of the persistent recurrent launches, and the host-side
    switches of a training step - held-back weight-gradient products (`defer_wgrad`, `defer_lids`, the queue `deferred`), the
    data-parallel hook run behind a layer's weight-gradient products (`wgrad_hook`), the SyncBN context (`sync_bn`).

    `c` is an `aasLaunch` (include/aas_hip.h).  `launch_state(st)` installs the state for the calling THREAD - in this module
    (`state()`) and in the library (`aas_launch_scope`) - and puts the previous one back; the trainers' entry points run under it
    (`with_trainer_precision`), and every autograd function below records the state its forward ran under and runs its backward
    under the same one, on whichever thread autograd executes it.  The recurrent launches receive the struct as an argument
    (`aas_*_ex`).  So two trainers in one process - in one thread or in two - never queue a launch under each other's settings;
    nothing here is a process-wide cell that one call sets and a later call consumes.  An unset field (None / -1) falls through to
    the process setting (`ops.set_precision`, `aas_set_debug_flags`, ...), which is what code outside any trainer sees."""

    def __init__(self, gemm_max_steps=None, debug_flags=None, precision=None):
        self.c = _new_claunch()
        self.gemm_max_steps, self.debug_flags, self.precision = gemm_max_steps, debug_flags, precision
        self.defer_wgrad, self.defer_lids, self.deferred = False, set(), []
        self.wgrad_hook = None      # callable(list of .grad views) run on the side stream after a layer's products are queued
        self.sync_bn = None         # a dist.DPContext: train-mode BatchNorm statistics are all-reduced over the ranks (SyncBN)

    def _field(name):   # noqa: N805  (a property per struct field: None <-> "unset")
        unset = 0 if name == "rnn_tag" else -1
        return property(lambda self: (None if getattr(self.c, name) == unset else getattr(self.c, name)),
                        lambda self, v: setattr(self.c, name, unset if v is None else int(v)))
    gemm_max_steps, debug_flags, precision, rnn_cu_limit = _field("gemm_max_steps"), _field("debug_flags"), _field("precision"), _field("rnn_cu_limit")
    del _field


_PROCESS = LaunchState()     # "no trainer installed": every field unset - the process settings apply
_process_precision = [int(__import__('os').environ.get('AAS_PRECISION', '0'))]
_tls = _threading.local()


def state():
    """The LaunchState installed for this thread (`launch_state`), else the process-level one."""
    return getattr(_tls, "state", None) or _PROCESS


def _launch_arg(tag, row_len=None):
    """-> ctypes reference to the aasLaunch a recurrent launch is called with: the installed state's struct, or this thread's scratch
    struct (all settings unset: the process settings apply) - with the launch tag and the row classes of THIS launch filled in."""
    st = state()
    if st is _PROCESS:
        c = getattr(_tls, "scratch", None)
        if c is None:
            c = _tls.scratch = _new_claunch()
    else:
        c = st.c
    c.rnn_tag = int(tag)
    if row_len is not None:
        c.cls_n_first, c.cls_T_first, c.cls_T_rest = int(row_len[0]), int(row_len[1]), int(row_len[2])
    else:
        c.cls_n_first = -1
    return c


class launch_state(object):
    """`with launch_state(st):` - install `st` for this thread (None / the process-level state: nothing installed), previous one back
    on exit.  Re-entrant; cheap (two library calls)."""

    def __init__(self, st):
        self.st = None if st is _PROCESS else st

    def __enter__(self):
        self.prev = getattr(_tls, "state", None)
        if self.st is not self.prev:
            _tls.state = self.st
            check(lib().aas_launch_scope(_ct.byref(self.st.c) if self.st is not None else None, None), "aas_launch_scope")
        return self

    def __exit__(self, *exc):
        if self.st is not self.prev:
            _tls.state = self.prev
            lib().aas_launch_scope(_ct.byref(self.prev.c) if self.prev is not None else None, None)
        return False


class _PrecisionCell(object):
    """`_precision[0]`: the arithmetic mode launches are queued in right now, on this thread (the installed state's, else the process's)."""

    def __getitem__(self, i):
        p = state().precision
        return _process_precision[0] if p is None else p

    def __setitem__(self, i, v):
        set_precision(v)


_precision = _PrecisionCell()


def set_precision(mode):
    """0 (library default) = fp32 operands on fp32-input MFMA with fp32 accumulation in every GEMM and recurrent product - the
    arithmetic class of the reference's cuDNN / cuBLAS fp32 path, not bit-identical to it: summation orders differ as between any two
    GEMM tilings, and the reduce-scatter BPTT exchanges its partial sums as fp32 words whose two low mantissa bits carry the step tag
    (rounded to nearest there: a 22-bit exchange, |error| <= 2 ulp per partial; the guarantee is the fp64-error tests, not "exact");
    1 = split-bf16 fast mode (operands as bf16 hi/lo, 3 MFMAs per product, ~2^-17 per product: narrower than fp32, inside the parity
    budget); 2 = fp32-EQUIVALENT: small GEMMs and recurrent products as in mode 0, EXCEPT the 500-unit LSTM's BPTT, which runs the
    six-product bf16 kernel; the large GEMMs as six bf16 products of three-term operands (all 24 operand bits, dropped cross terms
    <= 2^-25: `gemm_planes6`).  `--precision` of main.py / am_train.py and the AAS_PRECISION environment variable select it for a whole
    run; a trainer keeps the mode it was built in (`Trainer.precision`, `ops.precision`).  Inside an installed LaunchState the call
    changes THAT state only; outside, the process setting."""
    mode = int(mode)
    if mode not in (0, 1, 2):
        raise RuntimeError("set_precision: mode must be 0 (fp32), 1 (split-bf16) or 2 (fp32-equivalent), got %r" % (mode,))
    st = state()
    if st is _PROCESS:
        check(lib().aas_set_precision(mode), "aas_set_precision")
        _process_precision[0] = mode
    else:
        st.precision = mode


def get_precision():
    return _precision[0]


class precision(object):
    """`with ops.precision(mode):` - the arithmetic mode of everything queued inside the block, the previous mode back on exit (the
    library reads the mode when a launch is queued).  The trainers wrap their step / validation entry points in it with the mode they
    were built for (`Trainer.precision`), so two trainers of different modes in one process, or a caller that flips the process-wide
    setting in between, cannot run a step in the wrong arithmetic.  mode None: leave the current setting alone."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.st = state()
        self.prev = self.st.precision if self.st is not _PROCESS else _process_precision[0]
        if self.mode is not None and int(self.mode) != self.prev:
            set_precision(self.mode)
        return self

    def __exit__(self, *exc):
        if self.st is _PROCESS:
            if _process_precision[0] != self.prev:
                set_precision(self.prev)
        else:
            self.st.precision = self.prev
        return False


def with_trainer_precision(fn):
    """Decorator for trainer entry points: run under the trainer's own launch state (`self.launch`, a LaunchState) in its own
    arithmetic mode (`self.precision`)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        with launch_state(getattr(self, "launch", None)):      # (the state carries the trainer's arithmetic mode: TrainerContext.precision)
            return fn(self, *a, **k)
    return wrapped


def _scoped(cls):
    """Class decorator for the autograd functions of this module: forward records the LaunchState it ran under, backward runs under
    the same one - autograd executes backward nodes on its own thread, where nothing is installed."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *a, **k):
        ctx.launch = state()
        return fwd(ctx, *a, **k)

    def backward(ctx, *g):
        st = ctx.launch
        if st is state():
            return bwd(ctx, *g)
        with launch_state(st):
            return bwd(ctx, *g)
    forward.__doc__, backward.__doc__ = fwd.__doc__, bwd.__doc__
    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


class RowWeights(object):
    """Per-utterance parameters of a batched pass, as ONE explicit argument (`wgrad_row_scale=` of the modules, `rs=` of the layer
    ops): `w` - a device vector [N] of weights applied to the PARAMETER gradients only (None: all ones); `classes` - the utterance
    classes [(first row, rows, weight as a device scalar or None), ...] the vector consists of, for the weight-gradient products that
    run once per class with the weight as their alpha; `row_len` = (n_first, T_first, T_rest) - two row classes of different sequence
    length inside the recurrent launches (ragged noisy / clean pair, include/aas_hip.h: aasLaunch.cls_*)."""
    __slots__ = ("w", "classes", "row_len")

    def __init__(self, w=None, classes=None, row_len=None):
        self.w, self.classes, self.row_len = w, classes, (tuple(int(v) for v in row_len) if row_len is not None else None)

    @staticmethod
    def of(x):
        return x if (x is None or isinstance(x, RowWeights)) else RowWeights(x)


class TrainerContext(object):
    """What every trainer class shares about its library context: the arithmetic mode it was built in, its launch settings, and the
    one way to switch the mode that also refreshes the cached weight operand planes of its networks."""

    def _init_context(self):
        self.launch = LaunchState(precision=get_precision())     # the mode in force when the trainer is built

    @property
    def launch(self):
        """This trainer's LaunchState (every entry point runs under it)."""
        return self._launch

    @launch.setter
    def launch(self, st):
        old = getattr(self, "_launch", None)
        if old is not None and st.precision is None:      # a replacement state that names no arithmetic mode keeps the trainer's
            st.precision = old.precision
        self._launch = st

    @property
    def precision(self):
        """The arithmetic mode this trainer runs in (a field of its launch state)."""
        return self.launch.precision

    @precision.setter
    def precision(self, mode):
        self.launch.precision = int(mode)

    def _context_networks(self):
        return [m for m in (getattr(self, n, None) for n in ("G", "D", "ASR", "model")) if m is not None]

    def set_precision(self, mode):
        """Switch this trainer to another arithmetic mode (0 fp32 / 1 split-bf16 / 2 fp32-equivalent) and bring the cached weight
        operand planes of its networks up to date for it, off the critical path."""
        self.precision = int(mode)
        with launch_state(self.launch):
            for net in self._context_networks():
                if any(True for _ in net.parameters()):
                    refresh_weight_planes(net)


    def _upload_small(self, host, dev):
        """Asynchronous H2D copy of a small host tensor through a ring of REUSED pinned staging buffers (a fresh
        pin_memory() per step costs a pinned allocation, and the copy from pageable memory would block the host)."""
        ring = getattr(self, "_pin_ring", None)
        if ring is None:
            ring = self._pin_ring = dict(i=0, slots=[None] * 8)
        k = ring["i"] % len(ring["slots"])
        ring["i"] += 1
        slot = ring["slots"][k]
        n = host.numel()
        if slot is None or slot[0].numel() < n or slot[0].dtype != host.dtype:
            slot = [torch.empty(max(256, 2 * n), dtype=host.dtype).pin_memory(), None]
            ring["slots"][k] = slot
        elif slot[1] is not None:
            slot[1].synchronize()        # the copy that last used this staging buffer (8 uploads ago) has long finished
        slot[0][:n].copy_(host.reshape(-1))
        out = slot[0][:n].to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        slot[1] = ev
        return out


def set_rnn_cu_limit(cus):
    """Cap the CUs of every persistent recurrent launch queued from now on under the installed launch state (0 / None = whole
    device); without one: the process setting."""
    st = state()
    if st is _PROCESS:
        check(lib().aas_set_rnn_cu_limit(int(cus or 0)), "aas_set_rnn_cu_limit")
    else:
        if cus is not None and int(cus) < 0:
            raise RuntimeError("set_rnn_cu_limit: negative limit")
        st.rnn_cu_limit = int(cus or 0)


def set_wgrad_cap(workgroups):
    """Grid cap of the row-major weight-gradient GEMMs queued from now on (0 = none); include/aas_hip.h: aas_set_wgrad_wg_cap."""
    st = state()
    if st is _PROCESS:
        lib().aas_set_wgrad_wg_cap(int(workgroups))
    else:
        st.c.wgrad_wg_cap = max(0, int(workgroups))


def device_cus():
    return int(lib().aas_device_cus())


# ---- off-critical-path weight gradients --------------------------------------------------------------
# When a parameter's .grad is pre-bound to a flat gradient buffer (dist.FlatBuffers) and DIRECT_WGRAD is on,
# the recurrent layers' weight-gradient GEMMs run on a side HIP stream and ACCUMULATE straight into .grad
# (autograd gets None for those inputs): they are only needed at the optimiser step, so they overlap the next
# layer's persistent BPTT launch instead of sitting on the backward critical path.  sync_wgrad() joins.
DIRECT_WGRAD = [True]    # kill switch; the path is taken only for parameters re-homed by dist.FlatBuffers (_aas_flat_grad)
LINEAR_DIRECT = [knobs.get("LINEAR_DIRECT")]   # pointwise linear layers take the same side-stream path
# (LaunchState.wgrad_hook: callable(list of .grad views) run on the side stream after a layer's products are queued)
_SKIP_WGRAD = [knobs.get("SKIP_WGRAD")]   # timing experiment only (knobs.py): never set in a product run
_wgrad_streams = {}


_chain_streams = {}


_lane_streams = {}
_hip_rt = [None]


def lane_stream(half, dev=None):
    """A HIP stream whose kernels run on ONE HALF of the CUs only (hipExtStreamCreateWithCUMask): bits 0..127 of the mask select 16
    CUs on each of the 8 XCDs, bits 128..255 the other 16 (tools/probe/cumask.cpp), so round-robin workgroup placement over the XCDs
    - what the XCD-aware recurrent launches count on - is unchanged.  Two chains of persistent recurrent launches on the two lanes
    cannot take each other's CUs: a persistent launch becomes resident only when all its workgroups find a CU, and an unmasked GEMM of
    the OTHER chain, dealt over every free CU, made it wait for that GEMM's end (the 1000-unit GRU forward launches of the acoustic
    chain: 1.10 ms beside the discriminator's projections, 0.78 alone).  half: 0 = low CU half, 1 = high."""
    import ctypes
    if dev is None:
        dev = torch.cuda.current_device()
    dev = torch.device("cuda", dev) if isinstance(dev, int) else dev
    key = (dev, int(half))
    st = _lane_streams.get(key)
    if st is None:
        if _hip_rt[0] is None:
            _hip_rt[0] = ctypes.CDLL("libamdhip64.so")
        words = (device_cus() + 31) // 32
        lo = words // 2
        mask = (ctypes.c_uint32 * words)(*[(0xFFFFFFFF if ((i < lo) == (half == 0)) else 0) for i in range(words)])
        h = ctypes.c_void_p()
        with torch.cuda.device(dev):
            rc = _hip_rt[0].hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), mask)
        if rc != 0 or not h.value:
            raise RuntimeError("hipExtStreamCreateWithCUMask failed (rc=%d)" % rc)
        st = _lane_streams[key] = torch.cuda.ExternalStream(h.value, device=dev)
    return st


CHAIN_LANES = [knobs.get("CHAIN_LANES")]   # the two chains of the AAS step on CU-masked lane streams


def chain_stream(dev=None):
    """THE second stream for a chain of persistent recurrent launches, one per device and process: every trainer shares it.
    (Streams are multiplexed onto four hardware queues; a second trainer with a side stream of its own made five busy streams
    and its steps took 23 ms instead of 17.)  AAS_CHAIN_PRIO=1 creates it with the highest priority (measured: slower)."""
    if dev is None:
        dev = torch.cuda.current_device()
    dev = torch.device("cuda", dev) if isinstance(dev, int) else dev
    if CHAIN_LANES[0]:
        return lane_stream(1, dev)
    s = _chain_streams.get(dev)
    if s is None:
        prio = 0
        if knobs.get("CHAIN_PRIO"):
            try:
                prio = min(torch.cuda.Stream.priority_range())
            except Exception:  # noqa: BLE001
                prio = 0
        s = _chain_streams[dev] = torch.cuda.Stream(device=dev, priority=prio)
    return s


def wgrad_stream(dev):
    s = _wgrad_streams.get(dev)
    if s is None and knobs.get("WGRAD_LANE") in ("0", "1"):
        # (experiment: the weight-gradient products confined to one CU half - a masked stream of its own)
        import ctypes
        if _hip_rt[0] is None:
            _hip_rt[0] = ctypes.CDLL("libamdhip64.so")
        half = int(knobs.get("WGRAD_LANE"))
        words = (device_cus() + 31) // 32
        mask = (ctypes.c_uint32 * words)(*[(0xFFFFFFFF if ((i < words // 2) == (half == 0)) else 0) for i in range(words)])
        h = ctypes.c_void_p()
        rc = _hip_rt[0].hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), mask)
        if rc != 0:
            raise RuntimeError("hipExtStreamCreateWithCUMask failed (rc=%d)" % rc)
        s = _wgrad_streams[dev] = torch.cuda.ExternalStream(h.value, device=dev)
    if s is None:
        # lowest priority: when a persistent recurrent launch and queued weight-gradient blocks compete for CUs, the
        # recurrent grid (which must become fully resident) is dispatched first
        prio = 0
        if knobs.get("WGRAD_PRIO"):
            try:
                prio = max(torch.cuda.Stream.priority_range())
            except Exception:  # noqa: BLE001
                prio = 0
        s = torch.cuda.Stream(device=dev, priority=prio)
        _wgrad_streams[dev] = s
    return s


# A trainer may hold back the weight-gradient products of a backward pass (DEFER_WGRAD) and release them later with
# flush_deferred_wgrad(): the products are HBM-heavy and, queued beside a latency-bound BPTT chain that is on the step's
# critical path, they slow its cross-CU exchange (D's BPTT launches at N=60: 0.96 ms alone, up to 1.9 ms beside them).
# LaunchState.defer_wgrad / .defer_lids (layer ids - model.py: _aas_layer_id - whose products are held back even when defer_wgrad is
# off) / .deferred (the queue).


def flush_deferred_wgrad():
    """Queue the held-back weight-gradient products of the installed launch state (on the weight-gradient stream, in the order they
    were produced)."""
    st = state()
    todo, st.deferred[:] = list(st.deferred), []
    for fn in todo:
        fn()


def sync_wgrad():
    """Make the current stream wait for every side-stream weight-gradient product issued so far."""
    flush_deferred_wgrad()
    for s in _wgrad_streams.values():
        torch.cuda.current_stream().wait_stream(s)


class _wgrad_gemm_cap(object):
    """The weight-gradient products (side stream) under their own workgroup-lifetime cap (knobs.WGRAD_MAXSTEPS; None = the same cap as
    every other GEMM): they run beside persistent launches that wait for whole CUs, the chain's own GEMMs run between them."""

    def __enter__(self):
        cap = knobs.get("WGRAD_MAXSTEPS")
        self.st = None
        if cap is not None:
            st = state()
            if st is _PROCESS:
                self.prev = int(lib().aas_get_gemm_max_steps())
                lib().aas_set_gemm_max_steps(int(cap))
            else:
                self.prev, st.gemm_max_steps = st.gemm_max_steps, int(cap)
            self.st = st
        return self

    def __exit__(self, *exc):
        if self.st is _PROCESS:
            lib().aas_set_gemm_max_steps(self.prev)
        elif self.st is not None:
            self.st.gemm_max_steps = self.prev
        return False


class Profiler:
    """Optional HIP-event timing of individual launches on the launching stream (bench.py roofline).
    `classes` selects which launch classes are bracketed with events: "rnn" and/or "gemm"."""
    enabled = False
    classes = ("rnn",)
    records = []  # (name, algorithmic_flops, start_event, end_event, T or 0)

    @classmethod
    def start(cls, classes=("rnn",)):
        cls.enabled, cls.classes, cls.records = True, tuple(classes), []

    @classmethod
    def stop(cls):
        """-> {name: dict(count, total_ms, avg_ms, flops_per_launch)}; call after a device sync."""
        cls.enabled = False
        out = {}
        for name, flops, e0, e1, T in cls.records:
            d = out.setdefault(name, dict(count=0, total_ms=0.0, flops=0.0, T=T))
            d["count"] += 1
            d["total_ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / d["count"]
            d["flops_per_launch"] = d["flops"] / d["count"]
        cls.records = []
        return out


class _timed:
    def __init__(self, klass, name, flops, T=0):
        self.on = Profiler.enabled and klass in Profiler.classes
        self.name, self.flops, self.T = name, flops, T

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *a):
        if self.on:
            self.e1.record(torch.cuda.current_stream())
            Profiler.records.append((self.name, self.flops, self.e0, self.e1, self.T))


# --------------------------------------------------------------------------------------- GEMM
def gemm(mode, M, N, K, A, lda, B, ldb, C, ldc, bias=None, addend=None, ldd=0, accumulate=False,
         batch=1, sA=0, sB=0, sC=0, kdivA=0, kouterA=0, kdivB=0, kouterB=0, a_off=0, b_off=0, c_off=0):
    """Raw GEMM on device pointers; *_off are element offsets into A/B/C."""
    with _timed("gemm", "gemm_%s" % ("nt", "nn", "tn")[mode], 2.0 * M * N * K * batch):
        _gemm_raw(mode, M, N, K, A, lda, B, ldb, C, ldc, bias, addend, ldd, accumulate, batch, sA, sB, sC, kdivA, kouterA,
                  kdivB, kouterB, a_off, b_off, c_off)


def _gemm_raw(mode, M, N, K, A, lda, B, ldb, C, ldc, bias, addend, ldd, accumulate, batch, sA, sB, sC, kdivA, kouterA,
              kdivB, kouterB, a_off, b_off, c_off):
    check(lib().aas_gemm_f32(stream(), mode, M, N, K, A.data_ptr() + 4 * a_off, lda, B.data_ptr() + 4 * b_off, ldb,
                             C.data_ptr() + 4 * c_off, ldc, ptr(bias), ptr(addend), ldd, int(accumulate),
                             batch, sA, sB, sC, kdivA, kouterA, kdivB, kouterB), "aas_gemm_f32")


def gemm_multi(mode, M, N, Ks, As, lda, Bs, ldb, Cs, ldc, accumulate=False, kdiv=0, kouterA=0, kouterB=0, alpha=None):
    """<= 4 products of equal shape in ONE launch (include/aas_hip.h: aas_gemm_f32_multi): As / Bs / Cs are device byte addresses,
    Ks the reduction extent of each problem; kdiv / kouter*: two-level reduction rows of both operands (TN); alpha: device scalar."""
    import ctypes
    n = len(Ks)
    vp = lambda xs: (ctypes.c_void_p * n)(*[int(x) for x in xs])
    ks = (ctypes.c_int * n)(*[int(k) for k in Ks])
    with _timed("gemm", "gemm_%s" % ("nt", "nn", "tn")[mode], sum(2.0 * M * N * k for k in Ks)):
        check(lib().aas_gemm_f32_multi(stream(), mode, n, M, N, ks, vp(As), lda, vp(Bs), ldb, vp(Cs), ldc, int(accumulate), int(kdiv), int(kouterA),
                                       int(kouterB), ptr(alpha)), "aas_gemm_f32_multi")


# ---- pre-split operand planes (split-bf16 GEMM with the fp32 -> hi/lo split hoisted out of the k-loop) ----------
class Planes(object):
    """Interleaved bf16 hi/lo planes of a [rows, K] operand (include/aas_hip.h: aas_gemm_planes): `buf` is
    [rows, Kp/32, 2, 32] bf16, pitch Kp (multiple of 32, zero pad)."""
    __slots__ = ("buf", "rows", "K", "Kp")

    def __init__(self, buf, rows, K, Kp):
        self.buf, self.rows, self.K, self.Kp = buf, rows, K, Kp

    def to_float(self):
        """[rows, Kp] fp32 reconstruction hi + lo (tests)."""
        b = self.buf.view(self.rows, self.Kp // 32, 2, 32).float()
        return (b[:, :, 0, :] + b[:, :, 1, :]).reshape(self.rows, self.Kp)


def _kp(K):
    return (K + 31) // 32 * 32


class Planes3(object):
    """The two plane sets of a three-term operand (include/aas_hip.h: aas_split_planes3): q1 = (m | h), q2 = (h | l), each in the
    layout of `Planes`, side by side in ONE buffer `buf` [rows, 2 * 2*Kp] bf16 - set 1 in the first 4 Kp bytes of a row, set 2 in
    the next 4 Kp.  Read as a single operand of k extent 2 Kp the row is (Q1 | Q2), so an NT product of two such operands is ONE
    launch of the three-product plane kernel (gemm_planes6); the row-major weight-gradient products take the column halves."""
    __slots__ = ("buf", "rows", "K", "Kp")

    def __init__(self, buf, rows, K, Kp):
        self.buf, self.rows, self.K, self.Kp = buf, rows, K, Kp

    @property
    def pitch(self):
        return 8 * self.Kp                   # bytes per row of either set

    def set_ptr(self, i, row0=0):
        return self.buf.data_ptr() + row0 * self.pitch + i * 4 * self.Kp

    def to_float(self):
        """[rows, Kp] fp32 reconstruction h + m + l (tests): exact."""
        v = self.buf.view(self.rows, 2, self.Kp // 32, 2, 32).float()
        a, b = v[:, 0], v[:, 1]                                       # (m | h), (h | l)
        return ((a[:, :, 1, :] + a[:, :, 0, :]) + b[:, :, 1, :]).reshape(self.rows, self.Kp)


def _new_planes3(rows, K, dev):
    Kp = _kp(K)
    return Planes3(torch.empty((rows, 4 * Kp), device=dev, dtype=torch.bfloat16), rows, K, Kp)


def split_planes3(x2d, rows, K, ld=None, off=0):
    p3 = _new_planes3(rows, K, x2d.device)
    check(lib().aas_split_planes3(stream(), x2d.data_ptr() + 4 * off, ld if ld is not None else K, rows, K, p3.Kp, p3.set_ptr(0), p3.set_ptr(1),
                                  p3.pitch), "aas_split_planes3")
    return p3


def add3_planes3(a, b, c, K):
    """out = a + b (+ c) plus its three-term plane sets (the next layer's input operand in the fp32-equivalent mode)."""
    out = torch.empty_like(a)
    p3 = _new_planes3(a.numel() // K, K, a.device)
    check(lib().aas_add3_planes3_f32(stream(), ptr(out), ptr(a), ptr(b), ptr(c), p3.rows, K, p3.Kp, p3.set_ptr(0), p3.set_ptr(1), p3.pitch),
          "aas_add3_planes3_f32")
    return out, p3


def gemm_planes6(M, N, K, A3, B3, C, ldc, bias=None, addend=None, ldd=0, accumulate=False):
    """C[M,N] (+)= A[M,K] B[N,K]^T with every product carried to fp32's 24 operand bits, in one launch: over the (m | h) halves of
    the rows the three-product kernel sums the small terms m m' + h m' + m h', over the (h | l) halves h h' + l h' + h l' - the
    k extent it multiplies is 2 K (K = the operands' Kp)."""
    assert K == A3.Kp == B3.Kp
    # (profiling: the algorithmic flops of the ONE logical product, 2 M N K)
    with _timed("gemm", "gemm_planes6", 2.0 * M * N * K):
        check(lib().aas_gemm_planes(stream(), M, N, 2 * K, A3.buf.data_ptr(), 2 * A3.Kp, B3.buf.data_ptr(), 2 * B3.Kp, C.data_ptr(), ldc,
                                    ptr(bias), ptr(addend), ldd, int(accumulate), 1, 0, 0, 0), "aas_gemm_planes")


def split_planes(x2d, rows, K, ld=None, row_scale=None, nb=0, off=0):
    """planes of x2d[r*ld + off + k] (* row_scale[r % nb])."""
    Kp = _kp(K)
    buf = torch.empty((rows, 2 * Kp), device=x2d.device, dtype=torch.bfloat16)
    check(lib().aas_split_planes(stream(), x2d.data_ptr() + 4 * off, ld if ld is not None else K, rows, K, Kp, ptr(buf),
                                 ptr(row_scale), nb), "aas_split_planes")
    return Planes(buf, rows, K, Kp)


def split_planes_t(x3d, T, nb, C, ld=None, row_scale=None, off=0, extra=0):
    """Transposed planes of a time-major [T*nb, C] matrix: planes[c][t*nbp + n], nbp = nb rounded up to 8; `extra` more
    zero k positions behind T*nbp (room for a shifted k window)."""
    nbp = (nb + 7) // 8 * 8
    Kp = _kp(T * nbp + extra)
    buf = torch.empty((C, 2 * Kp), device=x3d.device, dtype=torch.bfloat16)
    check(lib().aas_split_planes_t(stream(), x3d.data_ptr() + 4 * off, ld if ld is not None else C, T, nb, nbp, C, Kp, ptr(buf),
                                   ptr(row_scale)), "aas_split_planes_t")
    return Planes(buf, C, T * nbp, Kp), nbp


def split_planes_t_into(buf_rows, x3d, T, nb, nbp, C, Kp, ld=None, row_scale=None, off=0, tstride=0):
    """aas_split_planes_t[2] into rows [0, C) of an existing plane buffer view `buf_rows` (bf16 [C, 2*Kp])."""
    if tstride:
        check(lib().aas_split_planes_t2(stream(), x3d.data_ptr() + 4 * off, ld if ld is not None else C, tstride, T, nb, nbp, C, Kp,
                                        ptr(buf_rows), ptr(row_scale)), "aas_split_planes_t2")
    else:
        check(lib().aas_split_planes_t(stream(), x3d.data_ptr() + 4 * off, ld if ld is not None else C, T, nb, nbp, C, Kp,
                                       ptr(buf_rows), ptr(row_scale)), "aas_split_planes_t")


def gemm_planes_multi(M, N, K, items, lda, ldb, ldc):
    """items: list (<= 4) of (A_ptr, B_ptr, C_ptr) device byte addresses; C_i[M,N] += A_i[M,K] B_i[N,K]^T in ONE launch."""
    import ctypes
    n = len(items)
    arr = lambda k: (ctypes.c_void_p * n)(*[int(it[k]) for it in items])
    with _timed("gemm", "gemm_planes_wgrad", 2.0 * M * N * K * n):
        check(lib().aas_gemm_planes_multi(stream(), M, N, K, n, arr(0), arr(1), arr(2), lda, ldb, ldc), "aas_gemm_planes_multi")


_zero_blocks = {}


def _zero512(dev):
    z = _zero_blocks.get(dev)
    if z is None:
        z = _zero_blocks[dev] = torch.zeros(1024, dtype=torch.uint8, device=dev)
    return z


def gemm_planes_tn(problems, Ns, Nb, dev, accumulate=True, name="gemm_planes_wgrad", count_flops=True):
    """Weight-gradient style products straight from ROW-MAJOR planes (include/aas_hip.h: aas_gemm_planes_tn), <= 8 per launch.
    Each problem is a dict: A, B (device byte addresses of row 0 of the planes), lda, ldb (bytes per plane row), acols, bcols
    (plane columns), acol0, M, N, K, C0, C1 (device addresses; C1 = 0 when msplit >= M), msplit, ldc, n0, ta, tb, alpha
    (a device float tensor or None)."""
    import ctypes
    n = len(problems)
    vp = lambda k: (ctypes.c_void_p * n)(*[int(pr[k]) if pr.get(k) else None for pr in problems])
    ci = lambda k: (ctypes.c_int * n)(*[int(pr[k]) for pr in problems])
    c64 = lambda k: (ctypes.c_int64 * n)(*[int(pr[k]) for pr in problems])
    al = (ctypes.c_void_p * n)(*[pr["alpha"].data_ptr() if pr.get("alpha") is not None else None for pr in problems])
    flops = sum(2.0 * pr["M"] * pr["N"] * pr["K"] for pr in problems) if count_flops else 0.0
    with _timed("gemm", name, flops):
        check(lib().aas_gemm_planes_tn(stream(), n, vp("A"), vp("B"), vp("C0"), vp("C1"), al, ci("M"), ci("N"), ci("K"), ci("msplit"),
                                       ci("acol0"), ci("n0"), ci("ta"), ci("tb"), c64("lda"), ci("acols"), c64("ldb"), ci("bcols"),
                                       c64("ldc"), Ns, Nb, ptr(_zero512(dev)), int(accumulate)), "aas_gemm_planes_tn")


def gemm_planes(M, N, K, A, B, C, ldc, bias=None, addend=None, ldd=0, accumulate=False, batch=1, sA=0, sB=0, sC=0,
                a_off=0, b_off=0, c_off=0, lda=None, ldb=None, name="gemm_planes", count_flops=True):
    """C[M,N] (+)= A[M,K] B[N,K]^T on Planes operands; K is the k extent actually multiplied (multiple of 32);
    a_off/b_off are offsets in elements (multiples of 32 within a row, or whole rows), c_off in elements."""
    with _timed("gemm", name, 2.0 * M * N * K * batch if count_flops else 0.0):
        check(lib().aas_gemm_planes(stream(), M, N, K, A.buf.data_ptr() + 4 * a_off, lda if lda is not None else A.Kp,
                                    B.buf.data_ptr() + 4 * b_off, ldb if ldb is not None else B.Kp,
                                    C.data_ptr() + 4 * c_off, ldc, ptr(bias), ptr(addend), ldd, int(accumulate), batch, sA, sB, sC),
              "aas_gemm_planes")


def linear_fwd(x2d, W, bias=None):
    """y[R,N] = x[R,K] W[N,K]^T + b"""
    R, K = x2d.shape
    Nn = W.shape[0]
    y = torch.empty((R, Nn), device=x2d.device, dtype=torch.float32)
    gemm(NT, R, Nn, K, x2d, K, W, K, y, Nn, bias=bias)
    return y


def linear_bwd(x2d, W, dy, need_dx=True, need_db=False, need_dw=True):
    R, K = x2d.shape
    Nn = W.shape[0]
    dW = None
    if need_dw:
        dW = torch.empty((Nn, K), device=x2d.device, dtype=torch.float32)
        gemm(TN, Nn, K, R, dy, Nn, x2d, K, dW, K)
    dx = None
    if need_dx:
        dx = torch.empty((R, K), device=x2d.device, dtype=torch.float32)
        gemm(NN, R, K, Nn, dy, Nn, W, K, dx, K)
    db = colsum(dy, R, Nn) if need_db else None
    return dx, dW, db


def colsum(x2d, R, C):
    out = torch.empty((C,), device=x2d.device, dtype=torch.float32)
    check(lib().aas_colsum_f32(stream(), ptr(x2d), R, C, C, ptr(out), 0), "aas_colsum_f32")
    return out


def transpose(inp, out, B, R, C, isb, isr, osb, osc):
    check(lib().aas_transpose_f32(stream(), ptr(inp), ptr(out), B, R, C, isb, isr, osb, osc), "aas_transpose_f32")
    return out


def add3(a, b, c=None):
    out = torch.empty_like(a)
    check(lib().aas_add3_f32(stream(), ptr(out), ptr(a), ptr(b), ptr(c), a.numel()), "aas_add3_f32")
    return out


def add3_planes(a, b, c, K):
    """out = a + b (+ c) [..., K] plus its operand planes, attached to the result as `_aas_planes` for the next layer."""
    out = torch.empty_like(a)
    rows = a.numel() // K
    Kp = _kp(K)
    buf = torch.empty((rows, 2 * Kp), device=a.device, dtype=torch.bfloat16)
    check(lib().aas_add3_planes_f32(stream(), ptr(out), ptr(a), ptr(b), ptr(c), rows, K, Kp, ptr(buf)), "aas_add3_planes_f32")
    return out, Planes(buf, rows, K, Kp)


def scale_rows(x, scale, nb, out=None):
    """out[(t,n), :] = x[(t,n), :] * scale[n] (time-major rows); out=None -> new tensor, out=x -> in place."""
    C = x.shape[-1] if x.dim() == 2 else x.numel() // (x.shape[0] * x.shape[1])
    rows = x.numel() // C
    out = torch.empty_like(x) if out is None else out
    check(lib().aas_scale_rows_f32(stream(), ptr(out), ptr(x), ptr(scale), rows, nb, C), "aas_scale_rows_f32")
    return out


def axpby_(y, x, alpha, beta):
    check(lib().aas_axpby_f32(stream(), ptr(y), ptr(x), float(alpha), float(beta), y.numel()), "aas_axpby_f32")
    return y


# ---- layout helpers (each is its own inverse pair) -------------------------------------------
def nct_to_tnc(x):  # [N,C,T] -> [T,N,C]
    N, C, T = x.shape
    out = torch.empty((T, N, C), device=x.device, dtype=torch.float32)
    return transpose(x, out, N, C, T, C * T, T, C, N * C)


def tnc_to_nct(x):  # [T,N,C] -> [N,C,T]
    T, N, C = x.shape
    out = torch.empty((N, C, T), device=x.device, dtype=torch.float32)
    return transpose(x, out, N, T, C, C, N * C, C * T, T)


def nct_to_ntc(x):  # [N,C,T] -> [N,T,C]
    N, C, T = x.shape
    out = torch.empty((N, T, C), device=x.device, dtype=torch.float32)
    return transpose(x, out, N, C, T, C * T, T, T * C, C)


def ntc_to_nct(x):  # [N,T,C] -> [N,C,T]
    N, T, C = x.shape
    out = torch.empty((N, C, T), device=x.device, dtype=torch.float32)
    return transpose(x, out, N, T, C, T * C, C, C * T, T)


def swap01(x):  # [A,B,C] -> [B,A,C]
    A, B, C = x.shape
    out = torch.empty((B, A, C), device=x.device, dtype=torch.float32)
    check(lib().aas_swap01_f32(stream(), ptr(x), ptr(out), A, B, C), "aas_swap01_f32")
    return out


@_scoped
class _Layout(torch.autograd.Function):
    """Differentiable layout change; backward applies the inverse permutation."""
    FWD = {"nct_tnc": nct_to_tnc, "tnc_nct": tnc_to_nct, "nct_ntc": nct_to_ntc, "ntc_nct": ntc_to_nct, "swap01": swap01}
    INV = {"nct_tnc": "tnc_nct", "tnc_nct": "nct_tnc", "nct_ntc": "ntc_nct", "ntc_nct": "nct_ntc", "swap01": "swap01"}

    @staticmethod
    def forward(ctx, x, kind):
        require_cuda(x)
        ctx.kind = kind
        return _Layout.FWD[kind](_c(x))

    @staticmethod
    def backward(ctx, g):
        return _Layout.FWD[_Layout.INV[ctx.kind]](_c(g)), None


def layout(x, kind):
    return _Layout.apply(x, kind)


@_scoped
class _LayoutCatNCT(torch.autograd.Function):
    """[a ; b] along the batch axis and [N,C,T] -> [T,N,C] in one step: two transposing launches straight into the halves of the
    time-major tensor instead of a concatenated copy and a transpose of it (the batched [enhanced; clean] discriminator input).
    Only `a` receives a gradient.  a and b may differ in T: the result has max(Ta, Tb) frames, zero beyond the shorter one's last
    (the batched discriminator pass over a ragged pair, trainer_AAS._batched_D_core)."""

    @staticmethod
    def forward(ctx, a, b):
        require_cuda(a, b)
        a, b = _c(a), _c(b)
        Na, C, Ta = a.shape
        Nb, Tb = b.shape[0], b.shape[2]
        assert b.shape[1] == C
        N, T = Na + Nb, max(Ta, Tb)
        out = (torch.empty if Ta == Tb else torch.zeros)((T, N, C), device=a.device, dtype=torch.float32)
        check(lib().aas_transpose_f32(stream(), ptr(a), ptr(out), Na, C, Ta, C * Ta, Ta, C, N * C), "aas_transpose_f32")
        check(lib().aas_transpose_f32(stream(), ptr(b), out.data_ptr() + 4 * Na * C, Nb, C, Tb, C * Tb, Tb, C, N * C), "aas_transpose_f32")
        ctx.na, ctx.ta = Na, Ta
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        T, N, C = g.shape
        Ta = ctx.ta
        ga = torch.empty((ctx.na, C, Ta), device=g.device, dtype=torch.float32)
        # rows (t < Ta, n < Na) of g back to [Na, C, Ta]: in[b = n][r = t][c] with strides (C, N*C), out[b][c][t]
        check(lib().aas_transpose_f32(stream(), ptr(g), ptr(ga), ctx.na, Ta, C, C, N * C, C * Ta, Ta), "aas_transpose_f32")
        return ga, None


def layout_cat_nct_tnc(a, b):
    return _LayoutCatNCT.apply(a, b)


@_scoped
class _LayoutPairedCat(torch.autograd.Function):
    """The FSEGAN discriminator's batched input in one step: rows [0, N) = (leaf | mixture), rows [N, 2N) = (cleans | mixture) -
    `forward_paired` (model.py:233-238: cat along the feature axis) for both halves of the batch - laid down time-major
    [T, 2N, 2F] by four transposing launches, without the three concatenated copies and the transpose of the result.  Only
    `leaf` receives a gradient."""

    @staticmethod
    def forward(ctx, leaf, mixture, cleans):
        require_cuda(leaf, mixture, cleans)
        leaf, mixture, cleans = _c(leaf), _c(mixture), _c(cleans)
        N, F, T = leaf.shape
        assert tuple(mixture.shape) == (N, F, T) and tuple(cleans.shape) == (N, F, T)
        out = torch.empty((T, 2 * N, 2 * F), device=leaf.device, dtype=torch.float32)
        base = out.data_ptr()
        for src, n0, c0 in ((leaf, 0, 0), (mixture, 0, F), (cleans, N, 0), (mixture, N, F)):
            check(lib().aas_transpose_f32(stream(), ptr(src), base + 4 * (n0 * 2 * F + c0), N, F, T, F * T, T, 2 * F, 2 * N * 2 * F), "aas_transpose_f32")
        ctx.dims = (N, F, T)
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        N, F, T = ctx.dims
        ga = torch.empty((N, F, T), device=g.device, dtype=torch.float32)
        # rows n < N, channels c < F of g [T, 2N, 2F] back to [N, F, T]
        check(lib().aas_transpose_f32(stream(), ptr(g), ptr(ga), N, T, F, 2 * F, 2 * N * 2 * F, F * T, T), "aas_transpose_f32")
        return ga, None, None


def layout_paired_cat(leaf, mixture, cleans):
    return _LayoutPairedCat.apply(leaf, mixture, cleans)


# --------------------------------------------------------------------------------------- linear
@_scoped
class _LinearRows(torch.autograd.Function):
    """y[..., N] = x[..., K] W[N,K]^T (+ b) on the flattened leading dims."""

    @staticmethod
    def forward(ctx, x, W, b, rs=None):
        require_cuda(x, W)
        x = _c(x)
        W2 = _c(W).view(W.shape[0], -1)
        x2 = x.view(-1, x.shape[-1])
        y = linear_fwd(x2, W2, _c(b) if b is not None else None)
        ctx.save_for_backward(x2, W2)
        ctx.params = (W, b)          # the nn.Parameters themselves (direct accumulation into their flat-buffer .grad)
        rs = RowWeights.of(rs)
        ctx.rs, ctx.nb = (rs.w if rs is not None else None), (x.shape[1] if x.dim() == 3 else 1)
        ctx.wshape = W.shape
        ctx.has_b = b is not None
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], W2.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, W2 = ctx.saved_tensors
        gy2 = _c(gy).view(-1, W2.shape[0])
        W, b = ctx.params
        need_dw = ctx.needs_input_grad[1]
        need_db = ctx.has_b and ctx.needs_input_grad[2]
        # Parameter gradients off the critical path: when W (and b) live in a flat gradient buffer, dW = gy^T x and db =
        # colsum(gy) are accumulated straight into .grad on the weight-gradient stream (joined by ops.sync_wgrad before the
        # optimiser reads them); only dx, which the rest of the backward pass waits for, stays on this stream.
        direct = (DIRECT_WGRAD[0] and LINEAR_DIRECT[0] and need_dw and getattr(W, "_aas_flat_grad", False) and W.grad is not None and W.grad.is_contiguous()
                  and (not need_db or (getattr(b, "_aas_flat_grad", False) and b.grad is not None)))
        dx = None
        if ctx.needs_input_grad[0]:
            dx, _, _ = linear_bwd(x2, W2, gy2, need_dx=True, need_db=False, need_dw=False)
        if direct:
            main = torch.cuda.current_stream()
            side = wgrad_stream(gy2.device)
            ev = torch.cuda.Event()
            ev.record(main)
            rs, nb = ctx.rs, ctx.nb
            gW = W.grad.view(W2.shape[0], -1)
            gb = b.grad if need_db else None
            R_, K_, N_ = x2.shape[0], x2.shape[1], W2.shape[0]

            def run():
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    g2 = scale_rows(gy2, rs, nb) if rs is not None else gy2
                    gemm(TN, N_, K_, R_, g2, N_, x2, K_, gW, K_, accumulate=True)
                    if gb is not None:
                        check(lib().aas_colsum_f32(stream(), ptr(g2), R_, N_, N_, ptr(gb), 1), "aas_colsum_f32")
            for t_ in (gy2, x2):
                t_.record_stream(side)
            if state().defer_wgrad:
                state().deferred.append(run)
            else:
                run()
            return (dx.view(ctx.xshape) if dx is not None else None), None, None, None
        if ctx.rs is None:
            _, dW, db = linear_bwd(x2, W2, gy2, need_dx=False, need_db=need_db, need_dw=need_dw)
        else:  # per-utterance weights on the parameter gradients only
            gys = scale_rows(gy2, ctx.rs, ctx.nb)
            _, dW, db = linear_bwd(x2, W2, gys, need_dx=False, need_db=need_db, need_dw=need_dw)
        return (dx.view(ctx.xshape) if dx is not None else None), (dW.view(ctx.wshape) if dW is not None else None), db, None


def linear_rows(x, W, b=None, rs=None):
    return _LinearRows.apply(x, W, b, rs)


# --------------------------------------------------------------------------------------- RNN layers
PLANES_PRE = [knobs.get("PLANES_PRE")]   # input projections of the recurrent layers on the plane GEMM
PLANES_BWD = [knobs.get("PLANES_BWD")]   # their input-gradient and weight-gradient products too
PLANES_EMIT = [knobs.get("PLANES_EMIT")]  # the BPTT kernels write d(gates) as operand planes (no fp32, no split)


def _plane_sig(w_ih, w_ih_r, GH, I, tag):
    """(cacheable, signature) of a weight-plane cache entry.  The planes hang on the weight tensor itself and are valid while
    its storage address and version are unchanged: torch's version counter (bumped by every in-place torch write, e.g.
    load_state_dict) and, for parameters re-homed in dist.FlatBuffers, the buffer's own counter that optim.FlatAdam bumps
    (its HIP kernel writes through raw pointers).  Frozen weights are therefore split once, trainable ones once per
    optimiser step - by refresh_weight_planes(), right after the step and off the critical path.  (Not a table keyed by
    address: addresses are reused by other tensors.)"""
    flat = getattr(w_ih, "_aas_flat_ref", None)
    frozen = not (w_ih.requires_grad or w_ih_r.requires_grad)
    sig = (w_ih.data_ptr(), w_ih._version, w_ih_r.data_ptr(), w_ih_r._version, GH, I, tag, flat.version if flat is not None else None)
    return (frozen or flat is not None) and not torch.cuda.is_current_stream_capturing(), sig


def _wih_planes(w_ih, w_ih_r, GH, I):
    """[W_ih ; W_ih_rev] as planes [2*GH rows][I] (the B operand of the input projections of both directions)."""
    ok, sig = _plane_sig(w_ih, w_ih_r, GH, I, "N")
    ent = getattr(w_ih, "_aas_planes", None) if ok else None
    if ent is not None and ent[0] == sig:
        _planes_ready(w_ih)
        return ent[1]
    Kp = _kp(I)
    buf = torch.empty((2 * GH, 2 * Kp), device=w_ih.device, dtype=torch.bfloat16)
    check(lib().aas_split_planes(stream(), ptr(w_ih), I, GH, I, Kp, ptr(buf), None, 0), "aas_split_planes")
    check(lib().aas_split_planes(stream(), ptr(w_ih_r), I, GH, I, Kp, buf.data_ptr() + GH * Kp * 4, None, 0), "aas_split_planes")
    wb = Planes(buf, 2 * GH, I, Kp)
    if ok:
        try:
            w_ih._aas_planes = (sig, wb)
        except Exception:  # noqa: BLE001  (a tensor type that takes no attributes: just do not cache)
            pass
    return wb


def _wih_planes3(w_ih, w_ih_r, GH, I):
    """[W_ih ; W_ih_rev] as three-term plane sets [2*GH rows][I]."""
    ok, sig = _plane_sig(w_ih, w_ih_r, GH, I, "N3")
    ent = getattr(w_ih, "_aas_planes3", None) if ok else None
    if ent is not None and ent[0] == sig:
        _planes_ready(w_ih)
        return ent[1]
    wb = _new_planes3(2 * GH, I, w_ih.device)
    for d_, w_ in enumerate((w_ih, w_ih_r)):
        check(lib().aas_split_planes3(stream(), ptr(w_), I, GH, I, wb.Kp, wb.set_ptr(0, d_ * GH), wb.set_ptr(1, d_ * GH), wb.pitch),
              "aas_split_planes3")
    if ok:
        try:
            w_ih._aas_planes3 = (sig, wb)
        except Exception:  # noqa: BLE001
            pass
    return wb


def _wih_t_planes3(w_ih, w_ih_r, GH, I):
    """[W_ih ; W_ih_rev]^T as three-term plane sets [I rows][k = d*GH + g]."""
    ok, sig = _plane_sig(w_ih, w_ih_r, GH, I, "T3")
    ent = getattr(w_ih, "_aas_planes3_t", None) if ok else None
    if ent is not None and ent[0] == sig:
        _planes_ready(w_ih)
        return ent[1]
    wt = _new_planes3(I, 2 * GH, w_ih.device)
    dw = (w_ih_r.data_ptr() - w_ih.data_ptr()) // 4
    check(lib().aas_split_planes_t3(stream(), ptr(w_ih), I, dw, 2, GH, GH, I, wt.Kp, wt.set_ptr(0), wt.set_ptr(1), wt.pitch), "aas_split_planes_t3")
    if ok and not torch.cuda.is_current_stream_capturing():
        try:
            w_ih._aas_planes3_t = (sig, wt)
        except Exception:  # noqa: BLE001
            pass
    return wt


_refresh_streams = {}


def refresh_stream(dev):
    """The ONE utility stream per device (weight-plane refresh, the trainer's scalar reductions, the neutral caller of paired
    backward passes).  Streams are multiplexed onto a handful of hardware queues: a fifth busy stream shared a queue with one
    of the step's chains and serialised the step (18.0 -> 22.0 ms), so everything that is off the critical path shares this one."""
    dev = torch.device("cuda", dev) if isinstance(dev, int) else dev
    s = _refresh_streams.get(dev)
    if s is None:
        s = _refresh_streams[dev] = torch.cuda.Stream(device=dev)
    return s


def _planes_ready(w_ih):
    """A cached plane set may still be in flight on the refresh stream: its consumer waits for the layer's own event."""
    ev = getattr(w_ih, "_aas_planes_ready", None)
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)


def refresh_weight_planes(module):
    """Re-split the input weights of every recurrent layer of `module` into their operand planes (forward and transposed)
    on a stream of their own, right after the optimiser step; every layer gets an event that its next use waits for, so the
    next step starts as soon as ITS first layer's planes are there (not after all ~24 small launches of the three networks)."""
    dev = next(module.parameters()).device
    main, side = torch.cuda.current_stream(), refresh_stream(dev)
    ev = torch.cuda.Event()
    ev.record(main)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        for m in module.modules():
            if getattr(m, "_aas_layer_id", None) is None:
                continue
            w, wr = m.weight_ih_l0, m.weight_ih_l0_reverse
            GH, I = w.shape
            if _precision[0] in (1, 2) and PLANES_PRE[0] and I >= 64 and getattr(m, "kind", "") != "rnn":
                w._aas_planes_ready = None        # (the calls below must not wait for the previous refresh on this stream)
                if _precision[0] == 1:
                    bufs = (_wih_planes(w, wr, GH, I).buf,) + ((_wih_t_planes(w, wr, GH, I).buf,) if (PLANES_BWD[0] and w.requires_grad) else ())
                else:
                    p3 = (_wih_planes3(w, wr, GH, I),) + ((_wih_t_planes3(w, wr, GH, I),) if (PLANES_BWD[0] and w.requires_grad) else ())
                    bufs = tuple(x_.buf for x_ in p3)
                for t_ in bufs:
                    t_.record_stream(main)
                done = torch.cuda.Event()
                done.record(side)
                try:
                    w._aas_planes_ready = done
                except Exception:  # noqa: BLE001
                    side.synchronize()


def _wih_t_planes(w_ih, w_ih_r, GH, I):
    """[W_ih ; W_ih_rev]^T as planes [I rows][k = d*GH + g] (the B operand of dx = d(gates) W_ih)."""
    ok, sig = _plane_sig(w_ih, w_ih_r, GH, I, "T")
    frozen = ok
    ent = getattr(w_ih, "_aas_planes_t", None) if ok else None
    if ent is not None and ent[0] == sig:
        _planes_ready(w_ih)
        return ent[1]
    Kp = _kp(2 * GH)
    buf = torch.empty((I, 2 * Kp), device=w_ih.device, dtype=torch.bfloat16)
    dw = (w_ih_r.data_ptr() - w_ih.data_ptr()) // 4
    split_planes_t_into(buf, w_ih, 2, GH, GH, I, Kp, ld=I, tstride=dw)
    wt = Planes(buf, I, 2 * GH, Kp)
    if frozen and not torch.cuda.is_current_stream_capturing():
        try:
            w_ih._aas_planes_t = (sig, wt)
        except Exception:  # noqa: BLE001
            pass
    return wt


def _knob_changed(name, value):
    """knobs.set / knobs.override -> the module-level mirrors the hot path reads, and the library's own setters."""
    global _TN_FOLD
    mirrors = dict(LINEAR_DIRECT=LINEAR_DIRECT, SKIP_WGRAD=_SKIP_WGRAD, CHAIN_LANES=CHAIN_LANES, PLANES_PRE=PLANES_PRE, PLANES_BWD=PLANES_BWD,
                   PLANES_EMIT=PLANES_EMIT, CLASS_WGRAD=CLASS_WGRAD, MULTI_WGRAD=MULTI_WGRAD, TN_WGRAD=TN_WGRAD)
    if name in mirrors:
        mirrors[name][0] = value
    elif name == "TN_FOLD":
        _TN_FOLD = value
    elif name == "DEBUG_FLAGS":
        lib().aas_set_debug_flags(int(value))
    elif name == "WGRAD_WGS":
        lib().aas_set_wgrad_wg_cap(int(value))
    elif name == "GEMM32_MAXSTEPS":
        lib().aas_set_gemm_max_steps(int(value))


knobs.on_change(_knob_changed)

_GATES = {"lstm": 4, "gru": 3, "rnn": 1}
# fp32 mode: per-utterance weights folded into the weight-gradient GEMM (aas_gemm_tn_rowscaled_f32) instead of a scaling pass over
# d(gates).  Measured on one box (tools/ab.sh): 32.5-32.6 ms / step with the fold against 31.8 without - the products then start right
# behind the BPTT launch and take CUs from the input-gradient GEMM on the critical path, and the per-row weight lengthens the TN
# kernel's prefetch - so it is off by default.
_TN_FOLD = knobs.get("TN_FOLD")
CLASS_WGRAD = [knobs.get("CLASS_WGRAD")]   # ... per utterance class with alpha (no scaled copies of x / h)
MULTI_WGRAD = [knobs.get("MULTI_WGRAD")]   # fp32 arithmetic: a layer's four weight-gradient products as one aas_gemm_f32_multi launch
TN_WGRAD = [knobs.get("TN_WGRAD")]   # weight-gradient products from row-major planes (aas_gemm_planes_tn)


def _birnn_fwd(kind, x, w_ih, w_hh, w_ih_r, w_hh_r, lid=0, keep=None, row_len=None):
    """x [T,N,I] -> (hout[2,T,N,H], gact, cst).  row_len = (n_first, T_first, T_rest): two row classes of different sequence length
    in one launch (aasLaunch.cls_*; lstm / gru).  keep: a dict that receives what the layer's weight-gradient products can
    reuse: 'xp' = the input's operand planes, 'hx' / 'hpitch' = the forward launch's exchange buffer (h_t as planes)."""
    T, N, I = x.shape
    G = _GATES[kind]
    H = w_hh.shape[1]
    dev = x.device
    if kind == "lstm" and _precision[0] == 0 and knobs.get("FUSED_XPROJ") and I % 4 == 0 and max(I, H) <= 512:
        # the input projection inside the persistent launch (aas_lstm_fwd_x_ex): no `pre` tensor, no GEMM in front - when the library
        # covers the shape on the present CU budget (the enhancement network over the whole chip); 3 = it does not, nothing happened
        hout = torch.empty((2, T, N, H), device=dev, dtype=torch.float32)
        gact = torch.empty((2, T, N, 4 * H), device=dev, dtype=torch.float32)
        cst = torch.empty((2, T, N, H), device=dev, dtype=torch.float32)
        tag = 2 * lid if lid else 1
        la = _launch_arg(tag, row_len)
        xchg = _xchg_buf(dev, T, N, H, G, "fwd" if knobs.get("MANAGED_XCHG") else "any")
        with _timed("rnn", "lstm_fwdx[N=%d,H=%d]" % (N, H), 2.0 * 2 * T * N * G * H * (H + I), T) as tm:
            rc = lib().aas_lstm_fwd_x_ex(stream(), T, N, H, I, ptr(x), ptr(w_ih), ptr(w_ih_r), ptr(w_hh), ptr(w_hh_r), ptr(hout), ptr(gact),
                                         ptr(cst), ptr(_sync_buf(dev)), ptr(xchg), _ct.byref(la))
            tm.on = tm.on and rc == 0      # (nothing was launched: no record)
        if rc == 0:
            return hout, gact, cst
        if rc != 3:
            check(rc, "aas_lstm_fwd_x_ex")
        del hout, gact, cst
    pre = torch.empty((T, N, 2, G * H), device=dev, dtype=torch.float32)
    x2 = x.view(T * N, I)
    dw = (w_ih_r.data_ptr() - w_ih.data_ptr()) // 4  # element distance between the two directions' W_ih
    if _precision[0] == 2 and kind != "rnn" and PLANES_PRE[0] and T * N >= 1024 and I >= 64:
        # fp32-equivalent mode: three-term plane sets of x and [W_ih; W_ih_rev], one plane-GEMM launch over both (six products)
        GH = G * H
        xa = getattr(x, "_aas_planes3", None)
        if xa is None or xa.rows != T * N or xa.K != I:
            xa = split_planes3(x2, T * N, I)
        if keep is not None:
            keep["xp3"] = xa
        gemm_planes6(T * N, 2 * GH, xa.Kp, xa, _wih_planes3(w_ih, w_ih_r, GH, I), pre, 2 * GH)
    elif _precision[0] == 1 and PLANES_PRE[0] and T * N >= 1024 and I >= 64:
        # plane GEMM: x and [W_ih; W_ih_rev] as pre-split bf16 planes (one HBM-bound pass each; frozen weights are
        # split once), then one LDS-DMA-staged launch for both directions: pre[tn, d*GH + g]
        GH = G * H
        xa = getattr(x, "_aas_planes", None)       # the producing layer's direction sum may have written them already
        if xa is None or xa.rows != T * N or xa.K != I:
            xa = split_planes(x2, T * N, I)
        if keep is not None:
            keep["xp"] = xa
        wb = _wih_planes(w_ih, w_ih_r, GH, I)
        gemm_planes(T * N, 2 * GH, xa.Kp, xa, wb, pre, 2 * GH)
    elif dw > 0 and dw % 4 == 0:
        # both directions in ONE batched launch: same A, B strided by the distance between the two weight tensors
        # (they live in one flat parameter buffer), C = the two column halves of `pre`
        gemm(NT, T * N, G * H, I, x2, I, w_ih, I, pre, 2 * G * H, batch=2, sA=0, sB=dw, sC=G * H)
    else:
        gemm(NT, T * N, G * H, I, x2, I, w_ih, I, pre, 2 * G * H)
        gemm(NT, T * N, G * H, I, x2, I, w_ih_r, I, pre, 2 * G * H, c_off=G * H)
    hout = torch.empty((2, T, N, H), device=dev, dtype=torch.float32)
    gact = torch.empty((2, T, N, 4 * H), device=dev, dtype=torch.float32)
    sync = _sync_buf(dev)
    if kind == "rnn":
        keep = None
    if keep is not None and _precision[0] == 1:   # a buffer of the layer's own: it is read again by the backward pass
        xchg = torch.empty(int(lib().aas_rnn_xchg_bytes(T, N, H, G)), dtype=torch.uint8, device=dev)
    else:
        xchg = _xchg_buf(dev, T, N, H, G, "fwd" if (_precision[0] != 1 and knobs.get("MANAGED_XCHG")) else "any")
    rflops = 2.0 * 2 * T * N * H * G * H  # both directions, T steps of [N,H]x[H,G*H]
    tag = 2 * lid if lid else 1
    if kind == "rnn":
        if row_len is not None:
            raise NotImplementedError("row classes of different length: lstm / gru layers only")
        _set_tag(tag)
        with _timed("rnn", "rnn_fwd[N=%d,H=%d]" % (N, H), rflops, T):
            check(lib().aas_rnn_fwd(stream(), T, N, H, ptr(pre), ptr(w_hh), ptr(w_hh_r), ptr(hout), ptr(gact), ptr(sync)), "aas_rnn_fwd")
        return hout, gact, None
    # the launch parameters travel as an ARGUMENT (include/aas_hip.h: aasLaunch): tag, row classes, and - from the installed state -
    # CU budget, kernel-selection bits, arithmetic mode; the h-plane pitch comes back in the same struct
    la = _launch_arg(tag, row_len)
    if kind == "lstm":
        cst = torch.empty((2, T, N, H), device=dev, dtype=torch.float32)
        with _timed("rnn", "lstm_fwd[N=%d,H=%d]" % (N, H), rflops, T):
            check(lib().aas_lstm_fwd_ex(stream(), T, N, H, ptr(pre), ptr(w_hh), ptr(w_hh_r), ptr(hout), ptr(gact), ptr(cst),
                                        ptr(sync), ptr(xchg), _ct.byref(la)), "aas_lstm_fwd_ex")
    else:
        cst = None
        with _timed("rnn", "gru_fwd[N=%d,H=%d]" % (N, H), rflops, T):
            check(lib().aas_gru_fwd_ex(stream(), T, N, H, ptr(pre), ptr(w_hh), ptr(w_hh_r), ptr(hout), ptr(gact), ptr(sync),
                                       ptr(xchg), _ct.byref(la)), "aas_gru_fwd_ex")
    if keep is not None and _precision[0] == 1:
        pitch = int(la.fwd_h_pitch)
        if pitch > 0:
            keep["hx"], keep["hpitch"] = xchg, pitch
    return hout, gact, cst


def _set_tag(tag):
    """Launch tag of the recurrent launches that do not take an aasLaunch argument (the plane-emitting BPTT variants, nn.RNN)."""
    st = state()
    if st is _PROCESS:
        lib().aas_set_rnn_launch_tag(int(tag))
    else:
        st.c.rnn_tag = int(tag)


def _birnn_bwd(kind, dy, x, w_ih, w_hh, w_ih_r, w_hh_r, hout, gact, cst, residual, need_dx=True, need_dw=True, rs=None,
               direct=None, lid=0, keep=None):
    T, N, I = x.shape
    G = _GATES[kind]
    H = w_hh.shape[1]
    GH = G * H
    dev = x.device
    dy = _c(dy)
    sync = _sync_buf(dev)
    xchg = _xchg_buf(dev, T, N, H, G, "bwd" if knobs.get("MANAGED_XCHG") else "any")
    R = T * N
    use_planes = (_precision[0] == 1 and kind != "rnn" and PLANES_BWD[0] and R >= 1024 and H >= 64 and I >= 64 and GH % 8 == 0
                  and w_ih_r.data_ptr() != w_ih.data_ptr() and (w_ih_r.data_ptr() - w_ih.data_ptr()) % 4 == 0)
    # fp32-equivalent mode: fp32 BPTT (exact kernels), then d(gates) as three-term plane sets for the six-product GEMMs
    use_planes6 = (_precision[0] == 2 and kind != "rnn" and PLANES_BWD[0] and R >= 1024 and H >= 64 and I >= 64 and GH % 8 == 0
                   and w_ih_r.data_ptr() != w_ih.data_ptr() and (w_ih_r.data_ptr() - w_ih.data_ptr()) % 4 == 0)
    rflops = 2.0 * 2 * T * N * H * GH
    tag = 2 * lid + 1 if lid else 1
    _set_tag(tag)
    rs = RowWeights.of(rs)
    rsw = rs.w if rs is not None else None            # the per-utterance weight vector (device) or None
    classes_of = lambda: [(0, N, None)] if rs is None else rs.classes
    # d(gates) straight in the operand form of the layer's GEMMs (row-major bf16 hi|lo planes) when the plane path is taken:
    # no fp32 copy and no split pass between the BPTT launch and the input-gradient GEMM
    dgx = dgh = dgp = dghp = None
    Kpg = _kp(2 * GH)
    # (only when every consumer takes planes: the plane weight-gradient path needs the flat-buffer accumulate form)
    if use_planes and PLANES_EMIT[0] and T > 1 and (direct is not None or not need_dw):
        dgp = torch.empty((R, 2 * Kpg), device=dev, dtype=torch.bfloat16)
        if kind == "lstm":
            with _timed("rnn", "lstm_bwd[N=%d,H=%d]" % (N, H), rflops, T):
                rc = lib().aas_lstm_bwd_planes(stream(), T, N, H, ptr(dy), ptr(w_hh), ptr(w_hh_r), ptr(gact), ptr(cst), ptr(dgp), Kpg,
                                               ptr(sync), ptr(xchg))
            dghp = dgp
        else:
            dghp = torch.empty((R, 2 * Kpg), device=dev, dtype=torch.bfloat16)
            with _timed("rnn", "gru_bwd[N=%d,H=%d]" % (N, H), rflops, T):
                rc = lib().aas_gru_bwd_planes(stream(), T, N, H, ptr(dy), ptr(w_hh), ptr(w_hh_r), ptr(hout), ptr(gact), ptr(dgp), ptr(dghp),
                                              Kpg, ptr(sync), ptr(xchg))
        if rc == 3:          # this shape / mode has no plane-emitting kernel: fp32 d(gates) + split passes below
            dgp = dghp = None
        else:
            check(rc, "aas_%s_bwd_planes" % kind)
    dga3 = None
    if use_planes6 and PLANES_EMIT[0] and kind == "lstm" and T > 1 and (direct is not None or not need_dw):
        # fp32-equivalent mode: the six-product BPTT kernel writes d(gates) as the three-term plane sets of the layer's GEMMs
        dga3 = _new_planes3(R, 2 * GH, dev)
        with _timed("rnn", "lstm_bwd[N=%d,H=%d]" % (N, H), rflops, T):
            rc = lib().aas_lstm_bwd_planes3(stream(), T, N, H, ptr(dy), ptr(w_hh), ptr(w_hh_r), ptr(gact), ptr(cst), dga3.buf.data_ptr(), dga3.Kp,
                                            ptr(sync), ptr(xchg))
        if rc == 3:          # no six-product kernel for this shape: exact BPTT + the split pass below
            dga3 = None
        else:
            check(rc, "aas_lstm_bwd_planes3")
    if dgp is None and dga3 is None:
        dgx = torch.empty((T, N, 2, GH), device=dev, dtype=torch.float32)
        if kind == "rnn":
            with _timed("rnn", "rnn_bwd[N=%d,H=%d]" % (N, H), rflops, T):
                check(lib().aas_rnn_bwd(stream(), T, N, H, ptr(dy), ptr(w_hh), ptr(w_hh_r), ptr(gact), ptr(dgx), ptr(sync)), "aas_rnn_bwd")
            dgh = dgx
        elif kind == "lstm":
            with _timed("rnn", "lstm_bwd[N=%d,H=%d]" % (N, H), rflops, T):
                check(lib().aas_lstm_bwd_ex(stream(), T, N, H, ptr(dy), ptr(w_hh), ptr(w_hh_r), ptr(gact), ptr(cst), ptr(dgx),
                                            ptr(sync), ptr(xchg), _ct.byref(_launch_arg(tag))), "aas_lstm_bwd_ex")
            dgh = dgx
        else:
            dgh = torch.empty((T, N, 2, GH), device=dev, dtype=torch.float32)
            with _timed("rnn", "gru_bwd[N=%d,H=%d]" % (N, H), rflops, T):
                check(lib().aas_gru_bwd_ex(stream(), T, N, H, ptr(dy), ptr(w_hh), ptr(w_hh_r), ptr(hout), ptr(gact), ptr(dgx),
                                           ptr(dgh), ptr(sync), ptr(xchg), _ct.byref(_launch_arg(tag))), "aas_gru_bwd_ex")
    # the weight-gradient products (side stream) only need the BPTT launch's output: their event is recorded here, BEFORE the
    # input-gradient GEMM is queued, so they may start beside it
    ev_bptt = torch.cuda.Event()
    ev_bptt.record(torch.cuda.current_stream())
    x2 = x.view(T * N, I)
    dx = None
    if need_dx:
        dx = torch.empty((T, N, I), device=dev, dtype=torch.float32)
        dw = (w_ih_r.data_ptr() - w_ih.data_ptr()) // 4
        if use_planes6:
            if dga3 is None:
                dga3 = split_planes3(dgx.view(R, 2 * GH), R, 2 * GH)
            gemm_planes6(R, I, dga3.Kp, dga3, _wih_t_planes3(w_ih, w_ih_r, GH, I), dx, I, addend=dy if residual else None, ldd=I)
        elif use_planes:
            # dx[R, I] = d(gates)[R, 2GH] [W_ih ; W_ih_rev]: the NT plane GEMM on a row-major split of d(gates) (one HBM pass)
            # and the transposed weight planes (2 GH x I elements; cached when the weights are frozen)
            dga = Planes(dgp, R, 2 * GH, Kpg) if dgp is not None else split_planes(dgx.view(R, 2 * GH), R, 2 * GH)
            wt = _wih_t_planes(w_ih, w_ih_r, GH, I)
            gemm_planes(R, I, dga.Kp, dga, wt, dx, I, addend=dy if residual else None, ldd=I)
        elif dw > 0 and dw % 4 == 0:
            # dx = [dg_fwd | dg_rev] (K = 2GH) x [W_ih ; W_ih_rev]: one launch, B rows addressed two-level
            gemm(NN, R, I, 2 * GH, dgx, 2 * GH, w_ih, I, dx, I, addend=dy if residual else None, ldd=I, kdivB=GH, kouterB=dw)
        else:
            gemm(NN, R, I, GH, dgx, 2 * GH, w_ih, I, dx, I, addend=dy if residual else None, ldd=I)
            gemm(NN, R, I, GH, dgx, 2 * GH, w_ih_r, I, dx, I, accumulate=True, a_off=GH)
    if not need_dw:
        return dx, None, None, None, None

    def wgrads_planes(out):
        """The four weight-gradient products as ONE multi-problem plane GEMM: d(gates)^T (per-utterance weights folded into
        the transposing split - no scale_rows pass) against x^T and the time-shifted h^T planes, accumulated into `out`."""
        nbp = (N + 31) // 32 * 32                      # k offsets are whole 32-blocks: one time step = nbp k positions
        K = T * nbp
        Kp = _kp(K + nbp)                              # + one zero time step behind the data: room for the shifted windows
        bf = torch.bfloat16
        def transposed(src_f32, src_planes):
            out_ = torch.empty((2 * GH, 2 * Kp), device=dev, dtype=bf)
            if src_planes is not None:
                check(lib().aas_planes_transpose(stream(), ptr(src_planes), Kpg, T, N, nbp, 2 * GH, Kp, ptr(out_), ptr(rsw)), "aas_planes_transpose")
            else:
                split_planes_t_into(out_, src_f32, T, N, nbp, 2 * GH, Kp, ld=2 * GH, row_scale=rsw)
            return out_
        dgT = transposed(dgx, dgp)
        dghT = dgT if (dgh is dgx and dghp is dgp) else transposed(dgh, dghp)
        xT = torch.empty((I, 2 * Kp), device=dev, dtype=bf)
        split_planes_t_into(xT, x2, T, N, nbp, I, Kp, ld=I)
        hT = torch.empty((2 * H, 2 * Kp), device=dev, dtype=bf)
        split_planes_t_into(hT, hout, T, N, nbp, H, Kp, ld=H)                          # h_fwd
        split_planes_t_into(hT[H:], hout, T, N, nbp, H, Kp, ld=H, off=T * N * H)       # h_rev
        row = 4 * Kp                                   # bytes per plane row
        a_f, a_r = dgT.data_ptr(), dgT.data_ptr() + GH * row
        ah_f, ah_r = dghT.data_ptr(), dghT.data_ptr() + GH * row
        shift = nbp * 4                                # one time step inside a row, in bytes (hi | lo interleaved per 32-block)
        ih = [(a_f, xT.data_ptr(), out[0].data_ptr()), (a_r, xT.data_ptr(), out[2].data_ptr())]
        # forward direction: sum_{t>=1} dg[t]^T h_f[t-1] -> A window starts one step in; reverse: dg[t]^T h_r[t+1] -> B window does
        hh = [(ah_f + shift, hT.data_ptr(), out[1].data_ptr()), (ah_r, hT.data_ptr() + H * row + shift, out[3].data_ptr())]
        if I == H:
            gemm_planes_multi(GH, H, K, ih + hh, Kp, Kp, H)
        else:
            gemm_planes_multi(GH, I, K, ih, Kp, Kp, I)
            gemm_planes_multi(GH, H, K, hh, Kp, Kp, H)
        for t_ in (dgT, dghT, xT, hT):
            t_.record_stream(torch.cuda.current_stream())
        for t_ in (dgp, dghp):
            if t_ is not None:
                t_.record_stream(torch.cuda.current_stream())

    def wgrads_tn(out):
        """The same products straight from ROW-MAJOR planes: d(gates) as the BPTT kernel wrote it, the input planes of the
        forward projection, h_t as the forward recurrent launch published it - no transposed copies (aas_gemm_planes_tn).
        Per-utterance weights: one problem set per utterance class, its weight as the product's alpha.  False when the
        layer does not have every operand in that form (the transposed-plane path runs instead)."""
        if not (TN_WGRAD[0] and keep and dgp is not None and "xp" in keep and "hx" in keep):
            return False
        classes = classes_of()
        if not classes:
            return False
        xp, hx, hpitch = keep["xp"], keep["hx"], keep["hpitch"]
        if xp.rows != R or xp.K != I or hpitch < 4 * H or not all(o.is_contiguous() for o in out):
            return False
        a_x, a_h = dgp.data_ptr(), dghp.data_ptr()
        lda = 4 * Kpg
        # one launch per class: the classes accumulate into the SAME results (read-modify-write epilogue), launches are ordered
        for n0, ns, alpha in classes:
            base = dict(lda=lda, acols=Kpg, n0=n0, alpha=alpha)
            probs = [dict(base, A=a_x, B=xp.buf.data_ptr(), ldb=4 * xp.Kp, bcols=xp.Kp, acol0=0, M=2 * GH, N=I, K=T * ns,
                          C0=out[0].data_ptr(), C1=out[2].data_ptr(), msplit=GH, ldc=I, ta=0, tb=0),
                     # forward direction: sum_{t>=1} dg[t]^T h_f[t-1]; reverse: sum_{t<=T-2} dg[t]^T h_r[t+1]
                     dict(base, A=a_h, B=hx.data_ptr(), ldb=hpitch, bcols=hpitch // 4, acol0=0, M=GH, N=H, K=(T - 1) * ns,
                          C0=out[1].data_ptr(), C1=0, msplit=GH, ldc=H, ta=1, tb=0),
                     dict(base, A=a_h, B=hx.data_ptr() + R * hpitch, ldb=hpitch, bcols=hpitch // 4, acol0=GH, M=GH, N=H,
                          K=(T - 1) * ns, C0=out[3].data_ptr(), C1=0, msplit=GH, ldc=H, ta=0, tb=1)]
            gemm_planes_tn(probs, ns, N, dev, accumulate=True)
        cur = torch.cuda.current_stream()
        for t_ in (dgp, dghp, xp.buf, hx):
            t_.record_stream(cur)
        return True

    def wgrads_tn6(out):
        """fp32-equivalent mode: the row-major weight-gradient products as two passes over three-term plane sets - d(gates) (split
        above for the input-gradient product, or here), the input's sets from the forward projection, h_t split here from `hout`."""
        if not (TN_WGRAD[0] and keep and "xp3" in keep):
            return False
        classes = classes_of()
        if not classes:
            return False
        xp3 = keep["xp3"]
        if xp3.rows != R or xp3.K != I or not all(o.is_contiguous() for o in out):
            return False
        dg3 = dga3 if dga3 is not None else split_planes3(dgx.view(R, 2 * GH), R, 2 * GH)
        dgh3 = dg3 if dgh is dgx else split_planes3(dgh.view(R, 2 * GH), R, 2 * GH)
        h3 = split_planes3(hout.view(2 * R, H), 2 * R, H)
        for n0, ns, alpha in classes:
            base = dict(lda=dg3.pitch, acols=dg3.Kp, n0=n0, alpha=alpha)
            for pi_ in (0, 1):                 # pass 1 on the (m | h) sets, pass 2 on the (h | l) sets, into the same gradients
                hb = h3.set_ptr(pi_)
                probs = [dict(base, A=dg3.set_ptr(pi_), B=xp3.set_ptr(pi_), ldb=xp3.pitch, bcols=xp3.Kp, acol0=0, M=2 * GH, N=I, K=T * ns,
                              C0=out[0].data_ptr(), C1=out[2].data_ptr(), msplit=GH, ldc=I, ta=0, tb=0),
                         dict(base, A=dgh3.set_ptr(pi_), B=hb, ldb=h3.pitch, bcols=h3.Kp, acol0=0, M=GH, N=H, K=(T - 1) * ns,
                              C0=out[1].data_ptr(), C1=0, msplit=GH, ldc=H, ta=1, tb=0),
                         dict(base, A=dgh3.set_ptr(pi_), B=hb + R * h3.pitch, ldb=h3.pitch, bcols=h3.Kp, acol0=GH, M=GH, N=H,
                              K=(T - 1) * ns, C0=out[3].data_ptr(), C1=0, msplit=GH, ldc=H, ta=0, tb=1)]
                gemm_planes_tn(probs, ns, N, dev, accumulate=True, name="gemm_planes6_wgrad", count_flops=(pi_ == 0))
        cur = torch.cuda.current_stream()
        for t_ in (dg3.buf, dgh3.buf, xp3.buf, h3.buf):
            t_.record_stream(cur)
        return True

    def wgrads(out, acc):
        if _SKIP_WGRAD[0]:  # timing experiment only (knobs.SKIP_WGRAD): how much of the step the weight-gradient products hold
            return
        nonlocal dgx, dgh
        if use_planes6 and acc and T > 1 and wgrads_tn6(out):
            return
        if dgx is None and dga3 is not None:      # (the BPTT wrote plane sets only and the plane product could not take them: fp32 again)
            dgx = dgh = dga3.to_float()[:, :2 * GH].contiguous().view(T, N, 2, GH)
        if use_planes and acc and T > 1:
            if wgrads_tn(out):
                return
            return wgrads_planes(out)
        tn = lambda *a_, **k_: gemm(TN, *a_, **k_)
        xw, hw = x2, hout
        multi_ok = MULTI_WGRAD[0] and _precision[0] != 1 and not (rs is not None and _precision[0] == 0 and _TN_FOLD) and all(o.is_contiguous() for o in out)
        by_class = multi_ok and rs is not None and CLASS_WGRAD[0] and bool(rs.classes)
        if rs is not None and not by_class:  # per-utterance weights on the parameter gradients only (dx used the unscaled d(gates))
            if _precision[0] == 0 and _TN_FOLD and all(o.is_contiguous() for o in out):
                # fp32 mode, optional: the weight rides on the reduction rows while the GEMM stages them (no pass at all)
                def tn(M_, N_, K_, A_, lda_, B_, ldb_, C_, ldc_, a_off=0, b_off=0, accumulate=False):
                    with _timed("gemm", "gemm_tn", 2.0 * M_ * N_ * K_):
                        check(lib().aas_gemm_tn_rowscaled_f32(stream(), M_, N_, K_, A_.data_ptr() + 4 * a_off, lda_, B_.data_ptr() + 4 * b_off, ldb_,
                                                              C_.data_ptr(), ldc_, int(accumulate), ptr(rsw), N), "aas_gemm_tn_rowscaled_f32")
            else:
                # the weight of reduction row (t, n) may ride on EITHER operand of the product: scale the narrow ones - x [R, I] and
                # h [2R, H] - into copies instead of d(gates) [R, 2 G H] in place (8-16x fewer bytes; d(gates) stays untouched)
                xw = scale_rows(x2, rsw, N)
                if T > 1:
                    hw = scale_rows(hout.view(2 * R, H), rsw, N)
        if multi_ok:
            # the four products of the layer (both directions' dW_ih and dW_hh) share d(gates): ONE launch of 4 x (GH/128 x I/128)
            # tiles fills the chip without split-K (two launches when the input and hidden widths differ).  Per-utterance weights
            # (the batched discriminator pass): one launch per utterance CLASS over that class's reduction rows - two-level row
            # addressing, rows (t, n0 .. n0+ns) of every time step - with the class's weight as the product's alpha: no scaled
            # copies of x / h (10 scale_rows launches, 1 ms of HBM-bound stream time per step before)
            classes = classes_of()
            if classes and (rs is None or by_class):
                a0, ah, b0, bh = dgx.data_ptr(), dgh.data_ptr(), x2.data_ptr(), hout.data_ptr()
                first = True
                for n0, ns, alpha in classes:
                    two = ns != N
                    kw = dict(kdiv=ns, kouterA=N * 2 * GH, kouterB=N * I, alpha=alpha) if two else dict(alpha=alpha)
                    kwh = dict(kw, kouterB=N * H) if two else kw
                    oa, ox, oh = 4 * n0 * 2 * GH, 4 * n0 * I, 4 * n0 * H          # byte offsets of the class's first row
                    ih = ([a0 + oa, a0 + oa + 4 * GH], [b0 + ox, b0 + ox], [out[0].data_ptr(), out[2].data_ptr()], [T * ns, T * ns])
                    Rm = (T - 1) * ns
                    hh = ([ah + oa + 4 * N * 2 * GH, ah + oa + 4 * GH], [bh + oh, bh + oh + 4 * (T * N * H + N * H)],
                          [out[1].data_ptr(), out[3].data_ptr()], [Rm, Rm])
                    a_ = acc or not first
                    if T > 1 and I == H:
                        gemm_multi(TN, GH, I, ih[3] + hh[3], ih[0] + hh[0], 2 * GH, ih[1] + hh[1], I, ih[2] + hh[2], I, accumulate=a_, **kw)
                    else:
                        gemm_multi(TN, GH, I, ih[3], ih[0], 2 * GH, ih[1], I, ih[2], I, accumulate=a_, **kw)
                        if T > 1:
                            gemm_multi(TN, GH, H, hh[3], hh[0], 2 * GH, hh[1], H, hh[2], H, accumulate=a_, **kwh)
                        elif not a_:
                            out[1].zero_()
                            out[3].zero_()
                    first = False
                return
            a0, ah, b0, bh = dgx.data_ptr(), dgh.data_ptr(), xw.data_ptr(), hw.data_ptr()
            ih = ([a0, a0 + 4 * GH], [b0, b0], [out[0].data_ptr(), out[2].data_ptr()], [R, R])
            Rm = (T - 1) * N
            hh = ([ah + 4 * N * 2 * GH, ah + 4 * GH], [bh, bh + 4 * (T * N * H + N * H)], [out[1].data_ptr(), out[3].data_ptr()], [Rm, Rm])
            if T > 1 and I == H:
                gemm_multi(TN, GH, I, ih[3] + hh[3], ih[0] + hh[0], 2 * GH, ih[1] + hh[1], I, ih[2] + hh[2], I, accumulate=acc)
            else:
                gemm_multi(TN, GH, I, ih[3], ih[0], 2 * GH, ih[1], I, ih[2], I, accumulate=acc)
                if T > 1:
                    gemm_multi(TN, GH, H, hh[3], hh[0], 2 * GH, hh[1], H, hh[2], H, accumulate=acc)
                elif not acc:
                    out[1].zero_()
                    out[3].zero_()
            return
        tn(GH, I, R, dgx, 2 * GH, xw, I, out[0], I, accumulate=acc)
        tn(GH, I, R, dgx, 2 * GH, xw, I, out[2], I, a_off=GH, accumulate=acc)
        if T > 1:
            Rm = (T - 1) * N
            # forward direction: sum_{t>=1} dg[t,:,0,:]^T h_f[t-1]   (a_off is a whole number of time steps: row r still belongs to utterance r % N)
            tn(GH, H, Rm, dgh, 2 * GH, hw, H, out[1], H, a_off=N * 2 * GH, accumulate=acc)
            # reverse direction: sum_{t<=T-2} dg[t,:,1,:]^T h_r[t+1]
            tn(GH, H, Rm, dgh, 2 * GH, hw, H, out[3], H, a_off=GH, b_off=T * N * H + N * H, accumulate=acc)
        elif not acc:
            out[1].zero_()
            out[3].zero_()

    if direct is not None:
        main = torch.cuda.current_stream()
        side = wgrad_stream(dev)
        # (plane path only: the fp32 path scales d(gates) IN PLACE for the per-utterance weights, which must not overlap the
        #  input-gradient GEMM that reads them)
        # When the products may start: right behind the BPTT launch (beside the input-gradient GEMM) on the plane paths and for layers
        # without per-utterance weights; after the input-gradient GEMM for the weighted layers of the fp32 path (the discriminator's):
        # nothing mutates d(gates) any more, but started early those products take CUs from the input-gradient GEMM, which is on
        # the critical path (32.5 vs 31.8 ms)
        fold_ok = _precision[0] == 0 and _TN_FOLD and all(o.is_contiguous() for o in direct)
        no_mutation = use_planes or use_planes6 or rs is None or fold_ok
        ev = ev_bptt if (no_mutation and T > 1 and knobs.get("WGRAD_EARLY")) else torch.cuda.Event()
        if ev is not ev_bptt:
            ev.record(main)
        hook = state().wgrad_hook

        def run():
            with torch.cuda.stream(side), _wgrad_gemm_cap():
                side.wait_event(ev)
                wgrads(direct, True)
                if hook is not None:
                    hook(direct)
        for t_ in (dgx, dgh, dgp, dghp, x, hout) + ((keep["xp"].buf if "xp" in keep else None, keep.get("hx")) if keep else ()) + (
                (dga3.buf,) if dga3 is not None else ()):
            if t_ is not None:
                t_.record_stream(side)
        if state().defer_wgrad or lid in state().defer_lids:
            state().deferred.append(run)      # (the closure keeps the layer's operands alive until it runs)
        else:
            run()
        return dx, None, None, None, None
    outs = [torch.empty((GH, I), device=dev, dtype=torch.float32), torch.empty((GH, H), device=dev, dtype=torch.float32),
            torch.empty((GH, I), device=dev, dtype=torch.float32), torch.empty((GH, H), device=dev, dtype=torch.float32)]
    wgrads(outs, False)
    return dx, outs[0], outs[1], outs[2], outs[3]


@_scoped
class _BiRNNLayer(torch.autograd.Function):
    """y = h_fwd + h_rev (+ x if residual)  for a bias-free bidirectional LSTM/GRU layer
    (reference model.py:80-86,101-105 and the residual adds at :223-226)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, w_ih_r, w_hh_r, kind, residual, rs=None, lid=0):
        require_cuda(x, w_ih, w_hh)
        rs = RowWeights.of(rs)
        ctx.rs, ctx.lid = rs, lid
        ctx.params = (w_ih, w_hh, w_ih_r, w_hh_r)  # the nn.Parameters themselves (for the direct-accumulate path)
        x = _c(x)
        w_ih, w_hh, w_ih_r, w_hh_r = _c(w_ih), _c(w_hh), _c(w_ih_r), _c(w_hh_r)
        trainable = any(ctx.needs_input_grad[1:5])    # (grad mode is off inside forward(): ask the context)
        ctx.keep = {} if (TN_WGRAD[0] and trainable and PLANES_BWD[0] and PLANES_EMIT[0]
                          and not torch.cuda.is_current_stream_capturing()) else None
        hout, gact, cst = _birnn_fwd(kind, x, w_ih, w_hh, w_ih_r, w_hh_r, lid, keep=ctx.keep, row_len=rs.row_len if rs is not None else None)
        T_, N_, H_ = hout.shape[1], hout.shape[2], hout.shape[3]
        if _precision[0] == 2 and kind != "rnn" and PLANES_PRE[0] and T_ * N_ >= 1024 and H_ >= 64 and H_ % 4 == 0:
            y, yp = add3_planes3(hout[0], hout[1], x if residual else None, H_)
            y._aas_planes3 = yp
        elif _precision[0] == 1 and PLANES_PRE[0] and T_ * N_ >= 1024 and H_ >= 64 and H_ % 4 == 0:
            y, yp = add3_planes(hout[0], hout[1], x if residual else None, H_)
            y._aas_planes = yp        # consumed by the next recurrent layer's input projection (same Python object is passed on)
        else:
            y = add3(hout[0], hout[1], x if residual else None)
        ctx.kind, ctx.residual = kind, residual
        ctx.save_for_backward(x, w_ih, w_hh, w_ih_r, w_hh_r, hout, gact, cst if cst is not None else hout)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_ih, w_hh, w_ih_r, w_hh_r, hout, gact, cst = ctx.saved_tensors
        direct = None
        if DIRECT_WGRAD[0] and all(ctx.needs_input_grad[1:5]):
            gr = [getattr(p_, "grad", None) if getattr(p_, "_aas_flat_grad", False) else None for p_ in ctx.params]
            if all(g is not None and g.is_contiguous() and g.shape == p_.shape for g, p_ in zip(gr, ctx.params)):
                direct = gr
        dx, a, b, c, d = _birnn_bwd(ctx.kind, dy, x, w_ih, w_hh, w_ih_r, w_hh_r, hout, gact, cst, ctx.residual,
                                    need_dx=ctx.needs_input_grad[0], need_dw=any(ctx.needs_input_grad[1:5]), rs=ctx.rs,
                                    direct=direct, lid=ctx.lid, keep=ctx.keep)
        ctx.keep = None
        return dx, a, b, c, d, None, None, None, None


def birnn_layer(x, w_ih, w_hh, w_ih_r, w_hh_r, kind="lstm", residual=False, rs=None, lid=0):
    return _BiRNNLayer.apply(x, w_ih, w_hh, w_ih_r, w_hh_r, kind, residual, rs, lid)


# --------------------------------------------------------------------------------------- batch norm
# (LaunchState.sync_bn: a dist.DPContext - train-mode statistics are all-reduced over the ranks: SyncBN)


def _direct_small(params):
    """True when every one of `params` (nn.Parameters or None) has its .grad inside a dist.FlatBuffers gradient buffer: a backward
    kernel with an accumulate flag can then add into it on the stream it runs on - what autograd's AccumulateGrad would do with one
    more launch per parameter."""
    return (DIRECT_WGRAD[0] and LINEAR_DIRECT[0]
            and all(p_ is not None and getattr(p_, "_aas_flat_grad", False) and p_.grad is not None and p_.grad.is_contiguous() for p_ in params))


@_scoped
class _BatchNormRows(torch.autograd.Function):
    """Train-mode BatchNorm over the rows of x[..., C] (+ fused LeakyReLU(slope)); updates running stats.
    With the launch state's `sync_bn` set (data parallel, --sync_bn) the per-channel sums and the row count are all-reduced between the
    statistics pass and the apply pass, forward and backward, so the result equals the single-process global batch."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, slope, nbt=None):
        require_cuda(x, gamma)
        x = _c(x)
        C = x.shape[-1]
        R = x.numel() // C
        y = torch.empty_like(x)
        stats = torch.empty((4, C), device=x.device, dtype=torch.float32)
        dp = state().sync_bn
        ctx.dp, ctx.rows = dp, None
        if dp is None:
            wsd = _wsd(x.device, 2 * C)
            check(lib().aas_bn_fwd(stream(), ptr(x), ptr(y), R, C, ptr(gamma), ptr(beta), float(eps), float(slope),
                                   ptr(stats), ptr(running_mean), ptr(running_var), float(momentum), ptr(wsd), ptr(nbt)), "aas_bn_fwd")
        else:
            red = torch.empty(2 * C + 1, device=x.device, dtype=torch.float64)   # [sum, sumsq, rows]: ONE collective
            check(lib().aas_bn_stats(stream(), ptr(x), R, C, ptr(red)), "aas_bn_stats")
            red[2 * C:].fill_(float(R))
            dp.reduce_scalars(red)
            ctx.rows = red[2 * C:]
            check(lib().aas_bn_apply(stream(), ptr(x), ptr(y), R, C, ptr(gamma), ptr(beta), float(eps), float(slope), ptr(stats),
                                     ptr(running_mean), ptr(running_var), float(momentum), ptr(red), ptr(ctx.rows), ptr(nbt)), "aas_bn_apply")
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.slope = slope
        ctx.params = (gamma, beta)      # the nn.Parameters themselves (direct accumulation into their flat-buffer .grad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        dy = _c(dy)
        C = x.shape[-1]
        R = x.numel() // C
        dx = torch.empty_like(x)
        # gamma / beta that live in a flat gradient buffer: the kernel's final reduce adds into .grad itself (no autograd accumulation launch)
        direct = _direct_small(ctx.params) and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
        dgamma = ctx.params[0].grad if direct else torch.empty_like(gamma)
        dbeta = ctx.params[1].grad if direct else torch.empty_like(beta)
        acc = 1 if direct else 0
        if ctx.dp is None:
            wsd = _wsd(x.device, 2 * C)
            check(lib().aas_bn_bwd(stream(), ptr(x), ptr(dy), ptr(dx), R, C, ptr(gamma), ptr(beta), float(ctx.slope),
                                   ptr(stats), ptr(dgamma), ptr(dbeta), acc, ptr(wsd)), "aas_bn_bwd")
        else:
            loc = torch.empty(2 * C, device=x.device, dtype=torch.float64)
            check(lib().aas_bn_bwd_reduce(stream(), ptr(x), ptr(dy), R, C, ptr(gamma), ptr(beta), float(ctx.slope), ptr(stats),
                                          ptr(loc)), "aas_bn_bwd_reduce")
            glob = loc.clone()
            ctx.dp.reduce_scalars(glob)
            check(lib().aas_bn_bwd_apply(stream(), ptr(x), ptr(dy), ptr(dx), R, C, ptr(gamma), ptr(beta), float(ctx.slope),
                                         ptr(stats), ptr(dgamma), ptr(dbeta), acc, ptr(glob), ptr(loc), ptr(ctx.rows)), "aas_bn_bwd_apply")
        if direct:
            return dx, None, None, None, None, None, None, None, None
        return dx, (dgamma if ctx.needs_input_grad[1] else None), (dbeta if ctx.needs_input_grad[2] else None), None, None, None, None, None, None


@_scoped
class _LeakyReLU(torch.autograd.Function):
    """nn.LeakyReLU(negative_slope) on its own (AM_training/model.py:364-367, include_first_BN=False: no BatchNorm to fuse it into)."""

    @staticmethod
    def forward(ctx, x, slope):
        require_cuda(x)
        x = _c(x)
        y = torch.empty_like(x)
        check(lib().aas_leaky_relu_f32(stream(), ptr(y), ptr(x), ptr(x), float(slope), x.numel()), "aas_leaky_relu_f32")
        ctx.save_for_backward(x)
        ctx.slope = float(slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(lib().aas_leaky_relu_f32(stream(), ptr(dx), ptr(dy), ptr(x), ctx.slope, x.numel()), "aas_leaky_relu_f32")
        return dx, None


def leaky_relu(x, slope):
    return _LeakyReLU.apply(x, slope)


def batchnorm_eval(x, gamma, beta, running_mean, running_var, eps=1e-5, slope=1.0):
    """Eval-mode BatchNorm (running statistics) + fused LeakyReLU; inference only (no autograd)."""
    require_cuda(x, gamma)
    x = _c(x.detach())
    C = x.shape[-1]
    y = torch.empty_like(x)
    check(lib().aas_bn_eval(stream(), ptr(x), ptr(y), x.numel() // C, C, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                            float(eps), float(slope)), "aas_bn_eval")
    return y


def softmax_rows(x):
    """softmax over the last dim (C <= 64); inference only (InferenceBatchSoftmax in eval mode, model.py:58-64)."""
    require_cuda(x)
    x = _c(x.detach())
    C = x.shape[-1]
    y = torch.empty_like(x)
    check(lib().aas_softmax_rows(stream(), ptr(x), ptr(y), x.numel() // C, C), "aas_softmax_rows")
    return y


def batchnorm_rows(x, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1, slope=1.0, num_batches_tracked=None):
    """num_batches_tracked: the module's int64 counter on the device - incremented by the apply launch itself."""
    if num_batches_tracked is not None:
        assert num_batches_tracked.dtype == torch.int64 and num_batches_tracked.is_cuda
    return _BatchNormRows.apply(x, gamma, beta, running_mean, running_var, eps, momentum, slope, num_batches_tracked)


# --------------------------------------------------------------------------------------- conv1d (k>1)
def _w_to_kf(W):  # [M,F,KW] -> [M,KW*F]
    M, F, KW = W.shape
    out = torch.empty((M, KW * F), device=W.device, dtype=torch.float32)
    return transpose(W, out, M, F, KW, F * KW, KW, KW * F, F)


def _kf_to_w(W2, F, KW):  # [M,KW*F] -> [M,F,KW]
    M = W2.shape[0]
    out = torch.empty((M, F, KW), device=W2.device, dtype=torch.float32)
    return transpose(W2, out, M, KW, F, KW * F, F, F * KW, KW)


@_scoped
class _Conv1dCL(torch.autograd.Function):
    """Temporal conv (no padding) on channels-last x[N,T,F] with PyTorch-layout weight [M,F,KW]:
    implicit-im2col batched GEMM (a row of the im2col matrix is KW consecutive frames of x)."""

    @staticmethod
    def forward(ctx, x, W, b, stride):
        require_cuda(x, W)
        x = _c(x)
        N, T, F = x.shape
        M, _, KW = W.shape
        T1 = (T - KW) // stride + 1
        if T1 < 1:
            raise RuntimeError("conv1d: input of %d frames is shorter than the kernel (%d)" % (T, KW))
        W2 = _w_to_kf(_c(W))
        y = torch.empty((N, T1, M), device=x.device, dtype=torch.float32)
        gemm(NT, T1, M, KW * F, x, stride * F, W2, KW * F, y, M, bias=_c(b) if b is not None else None,
             batch=N, sA=T * F, sB=0, sC=T1 * M)
        ctx.save_for_backward(x, W2)
        ctx.dims = (N, T, F, M, KW, T1, stride)
        ctx.has_b = b is not None
        ctx.bias_param, ctx.weight_param = b, W
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W2 = ctx.saved_tensors
        N, T, F, M, KW, T1, stride = ctx.dims
        dy = _c(dy)
        dW = db = None
        if ctx.needs_input_grad[1]:
            dW2 = torch.empty((M, KW * F), device=x.device, dtype=torch.float32)
            gemm(TN, M, KW * F, N * T1, dy, M, x, stride * F, dW2, KW * F, kdivB=T1, kouterB=T * F)
            if _direct_small((ctx.weight_param,)) and tuple(ctx.weight_param.grad.shape) == (M, F, KW):
                # [M, KW*F] -> ADDED into the flat-buffer .grad in the module's [M, F, KW] layout (no autograd accumulation launch)
                check(lib().aas_transpose_add_f32(stream(), ptr(dW2), ptr(ctx.weight_param.grad), M, KW, F, KW * F, F, F * KW, KW), "aas_transpose_add_f32")
            else:
                dW = _kf_to_w(dW2, F, KW)
        if ctx.has_b and ctx.needs_input_grad[2]:
            if _direct_small((ctx.bias_param,)):     # straight into the flat-buffer .grad
                check(lib().aas_colsum_f32(stream(), ptr(dy), N * T1, M, M, ptr(ctx.bias_param.grad), 1), "aas_colsum_f32")
            else:
                db = colsum(dy.view(N * T1, M), N * T1, M)
        dx = None
        if ctx.needs_input_grad[0]:
            dcol = torch.empty((N * T1, KW * F), device=x.device, dtype=torch.float32)
            gemm(NN, N * T1, KW * F, M, dy, M, W2, KW * F, dcol, KW * F)
            dx = torch.empty((N, T, F), device=x.device, dtype=torch.float32)
            check(lib().aas_col2im_f32(stream(), ptr(dcol), ptr(dx), N, T, T1, F, KW, stride), "aas_col2im_f32")
        return dx, dW, db, None


def conv1d_cl(x, W, b, stride):
    return _Conv1dCL.apply(x, W, b, stride)


# --------------------------------------------------------------------------------------- losses
@_scoped
class _L1Sum(torch.autograd.Function):
    """sum |a - b| over all elements (fp64 device accumulation), differentiable wrt both."""

    @staticmethod
    def forward(ctx, a, b):
        require_cuda(a, b)
        a, b = _c(a), _c(b)
        acc = torch.zeros((1,), device=a.device, dtype=torch.float64)
        check(lib().aas_l1_fwd(stream(), ptr(a), ptr(b), a.numel(), ptr(acc)), "aas_l1_fwd")
        ctx.save_for_backward(a, b)
        return acc.to(torch.float32).view(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        # the upstream scalar (loss weight / nElement) stays on the device: no host sync in backward
        g = _c(g.reshape(1).to(torch.float32))
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        check(lib().aas_l1_bwd(stream(), ptr(a), ptr(b), a.numel(), 1.0, ptr(g), ptr(ga), ptr(gb), 0), "aas_l1_bwd")
        return ga, gb


def l1_sum(a, b):
    return _L1Sum.apply(a, b)


# ---- device-resident step: loss roots without scaling / slicing / summing launches in between ------------------------------------
_unit_roots = {}


def unit_root(like):
    """A cached all-ones root gradient for `like` (a raw loss root): `torch.autograd.backward([root], [unit_root(root)])` queues no fill
    launch, and the fused loss functions below recognise it and skip the multiplication by it."""
    key = (like.device, like.dtype, tuple(like.shape))
    u = _unit_roots.get(key)
    if u is None:
        u = torch.ones(like.shape, device=like.device, dtype=like.dtype)
        u._aas_unit = True
        _unit_roots[key] = u
    return u


@_scoped
class _L1Pair(torch.autograd.Function):
    """The two masked-L1 sums of the batched discriminator pass (model.py:23-31 twice: trainer_AAS.py:146-147 on the enhanced rows
    with `enhanced` itself as the target, :176-177 on the clean rows), each with its weight / nElement scale folded into the
    backward launch: acc[0] += sum|ae[:N] - leaf|, acc[1] += sum|ae[N:] - clean| (acc: fp64 [2], zeroed by the step prologue and
    returned as the root).  Backward writes d(ae) for both halves into ONE tensor and d(leaf) (the target's gradient)."""

    @staticmethod
    def forward(ctx, ae, leaf, clean, s_ny, s_cl, acc, target_grad=None):
        """target_grad (a list, optional): the gradient wrt `leaf` AS THE TARGET of the first loss is appended to it instead of
        being returned to autograd - the caller adds it where it sums the gradients arriving at `enhanced` anyway (one three-input
        add3 launch instead of an autograd accumulation launch plus a two-input add3)."""
        require_cuda(ae, leaf, clean)
        ctx.target_grad = target_grad
        ae, leaf, clean = _c(ae), _c(leaf), _c(clean)
        assert acc.dtype == torch.float64 and acc.numel() == 2
        # ae [Nn + Nc, F, T]: rows [0, Nn) against leaf [Nn, F, Tn], rows [Nn, ..) against clean [Nc, F, Tc], T = max(Tn, Tc): a class
        # shorter than T takes the row-strided kernels over its own frames (ragged noisy / clean pair, one batched D pass)
        Nn, Fd, Tn = leaf.shape
        Nc, Tc = clean.shape[0], clean.shape[2]
        T = ae.shape[2]
        assert tuple(ae.shape) == (Nn + Nc, Fd, T) and clean.shape[1] == Fd and T == max(Tn, Tc)
        ctx.dims = (Nn, Nc, Fd, Tn, Tc, T)
        _L1Pair._fwd_class(ae, 0, leaf, Nn * Fd, Tn, T, acc, 0)
        _L1Pair._fwd_class(ae, Nn * Fd * T, clean, Nc * Fd, Tc, T, acc, 1)
        ctx.save_for_backward(ae, leaf, clean)
        ctx.scales = tuple((s_ if torch.is_tensor(s_) else float(s_)) for s_ in (s_ny, s_cl))    # python floats or device scalars (data parallel)
        ctx.mark_dirty(acc)
        return acc

    @staticmethod
    def _fwd_class(ae, off, tgt, rows, cols, T, acc, slot):
        a_ptr, acc_ptr = ae.data_ptr() + 4 * off, acc.data_ptr() + 8 * slot
        if cols == T:
            check(lib().aas_l1_fwd(stream(), a_ptr, ptr(tgt), rows * cols, acc_ptr), "aas_l1_fwd")
        else:
            check(lib().aas_l1_fwd2d(stream(), a_ptr, T, ptr(tgt), cols, rows, cols, acc_ptr), "aas_l1_fwd2d")

    @staticmethod
    def backward(ctx, g):
        ae, leaf, clean = ctx.saved_tensors
        Nn, Nc, Fd, Tn, Tc, T = ctx.dims
        gs = None if getattr(g, "_aas_unit", False) else _c(g.to(torch.float32))

        def factors(i):   # -> (host factor, device factor or None) of loss i: its scale times the upstream gradient
            sc, gi = ctx.scales[i], (gs[i:i + 1] if gs is not None else None)
            if not torch.is_tensor(sc):
                return sc, gi
            sc = _c(sc.reshape(1).to(torch.float32))
            return 1.0, (sc if gi is None else sc * gi)
        dae = torch.empty_like(ae)
        dleaf = torch.empty_like(leaf) if ctx.needs_input_grad[1] else None
        for i, (off, tgt, rows, cols, gt) in enumerate(((0, leaf, Nn * Fd, Tn, dleaf), (Nn * Fd * T, clean, Nc * Fd, Tc, None))):
            f_, d_ = factors(i)
            a_ptr, ga_ptr = ae.data_ptr() + 4 * off, dae.data_ptr() + 4 * off
            if cols == T:
                check(lib().aas_l1_bwd(stream(), a_ptr, ptr(tgt), rows * cols, f_, ptr(d_), ga_ptr, ptr(gt), 0), "aas_l1_bwd")
            else:       # the class's own frames; the padding columns of d(ae) are written as zeros
                check(lib().aas_l1_bwd2d(stream(), a_ptr, T, ptr(tgt), cols, rows, cols, f_, ptr(d_), ga_ptr, T, T, ptr(gt), cols), "aas_l1_bwd2d")
        if ctx.target_grad is not None and dleaf is not None:
            ctx.target_grad.append(dleaf)
            dleaf = None
        return dae, dleaf, None, None, None, None, None


def l1_pair(ae, leaf, clean, s_ny, s_cl, acc, target_grad=None):
    return _L1Pair.apply(ae, leaf, clean, s_ny, s_cl, acc, target_grad)


@_scoped
class _L1Scaled(torch.autograd.Function):
    """ONE masked-L1 sum (model.py:23-31) with its 1 / nElement (or weight / nElement) folded into the backward launch:
    acc[0] += sum|a - b| (acc: fp64, zeroed by the step prologue; a plain output buffer, not an autograd tensor); the returned
    root is an UNINITIALISED fp64 scalar that only anchors the backward pass (read the loss from `acc`).  Backward writes
    scale * sign(a - b) for `a` (and its negative for `b` when that needs a gradient) without a scaling launch in between."""

    @staticmethod
    def forward(ctx, a, b, scale, acc, grad_out=None):
        """grad_out (a list, optional): the gradient wrt `a` is appended to it instead of being returned to autograd - the caller
        adds it where it sums the gradients arriving at that tensor anyway (no autograd accumulation launch)."""
        require_cuda(a, b)
        a, b = _c(a), _c(b)
        assert acc.dtype == torch.float64 and acc.numel() >= 1
        check(lib().aas_l1_fwd(stream(), ptr(a), ptr(b), a.numel(), ptr(acc)), "aas_l1_fwd")
        ctx.save_for_backward(a, b)
        ctx.scale, ctx.grad_out = scale, grad_out
        return torch.empty((1,), device=a.device, dtype=torch.float64)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        gs = None if getattr(g, "_aas_unit", False) else _c(g.to(torch.float32))
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        sc = ctx.scale
        if torch.is_tensor(sc):       # a device scalar (data parallel: 1 / global nElement): it rides as the kernel's device factor
            gs = _c(sc.reshape(1).to(torch.float32)) if gs is None else _c((gs * sc).reshape(1).to(torch.float32))
            sc = 1.0
        check(lib().aas_l1_bwd(stream(), ptr(a), ptr(b), a.numel(), float(sc), ptr(gs), ptr(ga), ptr(gb), 0), "aas_l1_bwd")
        if ctx.grad_out is not None and ga is not None:
            ctx.grad_out.append(ga)
            ga = None
        return ga, gb, None, None, None


def l1_scaled(a, b, scale, acc, grad_out=None):
    return _L1Scaled.apply(a, b, scale, acc, grad_out)


class StepResult(dict):
    """What a device-resident training step hands back: device tensors, plus log scalars that are only FORMED when somebody reads
    them (`r["dce"]` = raw device sum x its normaliser) - a step that nobody logs queues no launch for them."""

    def __init__(self, *a, lazy=None, **k):
        super().__init__(*a, **k)
        self._lazy = dict(lazy or {})

    def __missing__(self, key):
        fn = self._lazy.get(key)
        if fn is None:
            raise KeyError(key)
        v = self[key] = fn()
        return v

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy


def ctc_prepare(labels, act_lens, label_lens, device):  # device may be "cpu": the caller uploads `meta` itself
    """Upload the CTC metadata (flat labels, label offsets, label lengths, act lengths) in ONE host->device copy.
    Call it BEFORE queueing the forward pass: a pageable-memory H2D copy blocks the host until the stream has
    drained, so doing it lazily inside the loss would serialise the host behind the whole acoustic model."""
    N = int(label_lens.numel())
    lab_lens_h = label_lens.to("cpu", torch.int32)
    max_l = int(lab_lens_h.max().item()) if N > 0 else 0
    offs_h = torch.zeros(N, dtype=torch.int32)
    if N > 1:
        offs_h[1:] = torch.cumsum(lab_lens_h, 0)[:-1].to(torch.int32)
    nl = int(labels.numel())
    meta = torch.cat([labels.to("cpu", torch.int32).view(-1), offs_h, lab_lens_h, act_lens.to("cpu", torch.int32)])
    meta = meta.to(device, non_blocking=True)
    return dict(meta=meta, nl=nl, N=N, max_l=max_l)


@_scoped
class _CTC(torch.autograd.Function):
    """sum_n -log p(l_n | softmax(acts[:len_n, n])); gradient wrt pre-softmax acts (warp-ctc semantics)."""

    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens, blank, prepared):
        require_cuda(acts)
        acts = _c(acts)
        T, N, C = acts.shape
        dev = acts.device
        pr = prepared if prepared is not None else ctc_prepare(labels, act_lens, label_lens, dev)
        if pr["N"] != N:
            raise ValueError("CTC: %d length entries for %d utterances" % (pr["N"], N))
        meta, nl, max_l = pr["meta"], pr["nl"], pr["max_l"]
        d_lab, d_off, d_ll, d_al = meta[:nl], meta[nl:nl + N], meta[nl + N:nl + 2 * N], meta[nl + 2 * N:]
        smax = 2 * max_l + 1
        ws = torch.empty((N * (T * smax + T),), device=dev, dtype=torch.float64)
        costs = torch.empty((N,), device=dev, dtype=torch.float32)
        grads = torch.empty_like(acts)
        lp = ptr(d_lab) if nl > 0 else ptr(meta)
        with _timed("ctc", "ctc[N=%d,T=%d,L<=%d]" % (N, T, max_l), 0.0, 2 * T):
            check(lib().aas_ctc_loss_async(stream(), ptr(acts), ptr(grads), lp, ptr(d_off), ptr(d_ll), ptr(d_al), C, N, T,
                                           max_l, ptr(costs), ptr(ws), int(blank), 1.0), "aas_ctc_loss_async")
        ctx.save_for_backward(grads)
        ctx.costs = costs
        # reduce the N costs with the colsum kernel (R = N rows, C = 1 column)
        out = torch.empty((1,), device=dev, dtype=torch.float32)
        check(lib().aas_colsum_f32(stream(), ptr(costs), N, 1, 1, ptr(out), 0), "aas_colsum_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        (grads,) = ctx.saved_tensors
        g = _c(g.reshape(-1)[:1].to(torch.float32))
        check(lib().aas_scale_dev_f32(stream(), ptr(grads), ptr(grads), ptr(g), 1.0, grads.numel()), "aas_scale_dev_f32")
        return grads, None, None, None, None, None


def ctc_sum(acts, labels, act_lens, label_lens, blank=0, prepared=None):
    return _CTC.apply(acts, labels, act_lens, label_lens, blank, prepared)


@_scoped
class _CTCScaled(torch.autograd.Function):
    """CTC with the loss weight (w_acoustic / N, trainer_AAS.py:168) folded into the kernel's gradient scale: the root is the vector
    of per-utterance costs [N] (summed and scaled where it is consumed - aas_began_step_raw), backward hands the scaled gradient over
    without a launch."""

    @staticmethod
    def forward(ctx, acts, blank, prepared, scale):
        require_cuda(acts)
        acts = _c(acts)
        T, N, C = acts.shape
        dev = acts.device
        pr = prepared
        if pr["N"] != N:
            raise ValueError("CTC: %d length entries for %d utterances" % (pr["N"], N))
        meta, nl, max_l = pr["meta"], pr["nl"], pr["max_l"]
        d_lab, d_off, d_ll, d_al = meta[:nl], meta[nl:nl + N], meta[nl + N:nl + 2 * N], meta[nl + 2 * N:]
        smax = 2 * max_l + 1
        ws = torch.empty((N * (T * smax + T),), device=dev, dtype=torch.float64)
        costs = torch.empty((N,), device=dev, dtype=torch.float32)
        grads = torch.empty_like(acts)
        lp = ptr(d_lab) if nl > 0 else ptr(meta)
        dscale = scale if torch.is_tensor(scale) else None      # a device scalar (data parallel: w_acoustic / global N)
        with _timed("ctc", "ctc[N=%d,T=%d,L<=%d]" % (N, T, max_l), 0.0, 2 * T):
            check(lib().aas_ctc_loss_async(stream(), ptr(acts), ptr(grads), lp, ptr(d_off), ptr(d_ll), ptr(d_al), C, N, T,
                                           max_l, ptr(costs), ptr(ws), int(blank), 1.0 if dscale is not None else float(scale)), "aas_ctc_loss_async")
        ctx.save_for_backward(grads)
        ctx.dscale = dscale
        return costs

    @staticmethod
    def backward(ctx, g):
        (grads,) = ctx.saved_tensors
        if ctx.dscale is not None:      # the kernel left the gradient unscaled: one launch applies the device scalar
            d = _c(ctx.dscale.reshape(1).to(torch.float32))
            check(lib().aas_scale_dev_f32(stream(), ptr(grads), ptr(grads), ptr(d), 1.0, grads.numel()), "aas_scale_dev_f32")
        if not getattr(g, "_aas_unit", False):     # a general root gradient: per-utterance factors on the [T, N, C] gradient
            grads = grads * g.to(torch.float32).view(1, -1, 1)
        return grads, None, None, None


def ctc_scaled(acts, blank, prepared, scale):
    return _CTCScaled.apply(acts, blank, prepared, scale)


def step_prologue(bufs, rs=None, n_neg=0, n_one=0, kt=None):
    """ONE launch at the head of a device-resident step: zero `bufs` (flat gradient buffers, loss accumulators) and write the
    per-utterance weights rs = [-kt] * n_neg + [1] * n_one of the batched discriminator pass (include/aas_hip.h: aas_step_prologue)."""
    import ctypes
    n = len(bufs)
    vp = (ctypes.c_void_p * max(n, 1))(*[int(b.data_ptr()) for b in bufs])
    sz = (ctypes.c_size_t * max(n, 1))(*[int(b.numel() * b.element_size()) for b in bufs])
    check(lib().aas_step_prologue(stream(), n, vp, sz, ptr(rs), int(n_neg), int(n_one), ptr(kt)), "aas_step_prologue")


def began_step_raw(l1_acc, s_ny, s_cl, costs, s_ctc, d_kt, d_out6, gamma, lambda_k, n_batch):
    check(lib().aas_began_step_raw(stream(), ptr(l1_acc), float(s_ny), float(s_cl), ptr(costs), int(costs.numel()), float(s_ctc), ptr(d_kt),
                                   ptr(d_out6), float(gamma), float(lambda_k), float(n_batch)), "aas_began_step_raw")


def began_step_sums(l1_sums, third, s0, s1, s2, d_kt, d_out6, gamma, lambda_k, n_batch, d_scales3=None, d_n_batch=None):
    """include/aas_hip.h: aas_began_step_sums - the controller from the raw device sums l1_sums[2], third[1] (optionally
    device-resident scales / N)."""
    check(lib().aas_began_step_sums(stream(), ptr(l1_sums), ptr(third), float(s0), float(s1), float(s2), ptr(d_scales3), ptr(d_kt), ptr(d_out6), float(gamma),
                                    float(lambda_k), float(n_batch), ptr(d_n_batch)), "aas_began_step_sums")


def loss_pack(l1_acc, costs, out3):
    check(lib().aas_loss_pack(stream(), ptr(l1_acc), ptr(costs), int(costs.numel()) if costs is not None else 0, ptr(out3)), "aas_loss_pack")


def sums_pack(a2, b1, out3):
    check(lib().aas_sums_pack(stream(), ptr(a2), ptr(b1), ptr(out3)), "aas_sums_pack")


def scales_from_counts(counts, weights, index, out):
    """out[i] = weights[i] / counts[index[i]] on the device (include/aas_hip.h: aas_scales_from_counts)."""
    import ctypes
    n = len(weights)
    w = (ctypes.c_double * n)(*[float(x) for x in weights])
    ix = (ctypes.c_int * n)(*[int(x) for x in index])
    check(lib().aas_scales_from_counts(stream(), ptr(counts), n, w, ix, ptr(out)), "aas_scales_from_counts")


# --------------------------------------------------------------------------------------- reductions / optimiser
def sqsum_into(acc, t):
    check(lib().aas_sqsum_f32(stream(), ptr(t), t.numel(), ptr(acc)), "aas_sqsum_f32")


def adam_step_dev(p, g, m, v, vmax, beta1, beta2, eps, d_hyper, amsgrad=True, grad_scale=1.0):
    """Adam update with step_size / sqrt(bias_correction2) read from the device tensor d_hyper[2]."""
    check(lib().aas_adam_dev_f32(stream(), ptr(p), ptr(g), ptr(m), ptr(v), ptr(vmax), p.numel(), float(beta1), float(beta2),
                                 float(eps), ptr(d_hyper), int(amsgrad), float(grad_scale)), "aas_adam_dev_f32")


def adam_tick(d_step, lr, beta1, beta2, d_hyper):
    check(lib().aas_adam_tick(stream(), ptr(d_step), float(lr), float(beta1), float(beta2), ptr(d_hyper)), "aas_adam_tick")


def began_step(l_ny, l_cl, l_ctc, d_kt, d_out6, gamma, lambda_k, n_batch):
    f = lambda t: _c(t.detach().reshape(1).to(torch.float32))
    a, b, c = f(l_ny), f(l_cl), f(l_ctc)
    check(lib().aas_began_step(stream(), ptr(a), ptr(b), ptr(c), ptr(d_kt), ptr(d_out6), float(gamma), float(lambda_k), float(n_batch)),
          "aas_began_step")


def adam_step(p, g, m, v, vmax, lr, beta1, beta2, eps, step, amsgrad=True, grad_scale=1.0):
    check(lib().aas_adam_f32(stream(), ptr(p), ptr(g), ptr(m), ptr(v), ptr(vmax), p.numel(), float(lr), float(beta1),
                             float(beta2), float(eps), int(step), int(amsgrad), float(grad_scale)), "aas_adam_f32")


def sgd_nesterov_step(p, g, buf, lr, momentum, grad_scale=1.0):
    check(lib().aas_sgd_nesterov_f32(stream(), ptr(p), ptr(g), ptr(buf), p.numel(), float(lr), float(momentum), float(grad_scale)),
          "aas_sgd_nesterov_f32")
