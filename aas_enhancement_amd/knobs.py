"""Schedule / kernel-selection switches of the hot path - the ONE place they live.

The shipped configuration is the table of defaults below.  In normal operation nothing here is read from the environment:
a switch that changes results or timing (`SKIP_WGRAD` drops the weight-gradient products altogether) must not be flippable
by a stray `AAS_*` variable in a product path.  Who may turn them:

  * tests and the A/B tooling, through this API: `knobs.override(NAME=value)` (context manager) / `knobs.set(NAME, value)`;
  * a developer shell, by exporting `AAS_ABLATION=1` TOGETHER with `AAS_<NAME>=...` (tools/ab.sh does) - without
    `AAS_ABLATION=1` such variables are ignored and named once on stderr.

`active()` returns every switch that differs from its default; `bench.py` refuses to print a headline value while it is
non-empty (or while the library's debug flags are non-zero) unless `--allow-ablation` labels the line as an ablation run.
`AAS_PRECISION` (the documented arithmetic-mode selector of a whole run), `AAS_DP_FORCE`, `AAS_NO_PIN` and
`AAS_BENCH_FORCE_SPAWN` are run configuration, not ablation switches, and are not handled here.
"""
import contextlib
import os
import sys

# name -> (default, parser)
_b = lambda s: str(s).strip().lower() in ("1", "true", "yes", "on")
_i = int
_s = str
_opt_i = lambda s: None if s in ("", "None", "auto") else int(s)

# name -> (default, parser, what it does)
_TABLE = {
    # ---- results-changing (timing experiments only; never in a product run)
    "SKIP_WGRAD": (False, _b, "RESULTS-CHANGING timing experiment: drop every weight-gradient product (\"how much of the step do they hold\")"),
    "DEBUG_FLAGS": (0, _i, "`aas_set_debug_flags` bits (`include/aas_hip.h`): A/B kernel selection and ablation bits"),
    # ---- schedule of the AAS step (trainer_AAS.py)
    "OVERLAP_ASR": (True, _b, "acoustic chain on a second stream beside the discriminator chain (0: one chain of persistent launches)"),
    "INTERLEAVE": (True, _b, "the two chains queued layer by layer in alternation"),
    "TWO_LANES": ("auto", _s, "auto / 1 / 0: the two-lane schedule (auto: only for ragged pairs below RAGGED_MIN_RATIO) vs the batched D pass"),
    "RAGGED_BATCHED": (True, _b, "ragged noisy / clean pair: ONE batched D pass with two row classes (0: the two-lane schedule)"),
    "RAGGED_MIN_RATIO": (0.65, float, "... when min(T) / max(T) is at least this (the batched pass runs max(T) steps for every row)"),
    "PAIR_BWD": ("auto", _s, "auto / 0 / 1: one autograd call over both chains' losses (auto: two calls whenever the host queues ahead)"),
    "NEUTRAL_BWD": (True, _b, "paired backward issued from a stream that has nothing queued"),
    "BWD_FROM": ("neutral", _s, "neutral / side: which stream issues the paired backward"),
    "ASR_EXACT": (False, _b, "acoustic model pinned to fp32 in the fast modes"),
    "EARLY_ADAM": (True, _b, "D's (and a trainable A's) Adam step on the weight-gradient stream, beside E's backward"),
    "DEFER_WGRAD": (False, _b, "hold ALL weight-gradient products back until E's backward (measured: no gain)"),
    "DEFER_D_LAYERS": (None, _opt_i, "D's top layers whose products are held back until E's backward (None = 2; 0 in the fp32-equivalent mode)"),
    "DEFER_A_LAYERS": (None, _opt_i, "the same for a trainable A (None = all of its recurrent layers: 30.7 -> 30.3 ms, profiles/r06_trainableA_sweep.txt)"),
    "EBWD_CUS": (128, _i, "CU budget of E's BPTT launches (the rest runs E's weight-gradient products)"),
    "LANE_CUS": (0, _i, "CU budget per lane of the two-lane schedule (0 = half the device)"),
    "FSEGAN_BWD_CUS": (0, _i, "trainer_FSEGAN: CU budget of the BPTT launches (0 = whole device); the rest runs the weight-gradient products"),
    "FSEGAN_DEFER_D": (0, _i, "trainer_FSEGAN: D's top layers whose weight-gradient products are held back until E's backward"),
    "AC_BWD_CUS": (0, _i, "trainer_acoustic: CU budget of the BPTT launches (0 = whole device)"),
    "AM_FWD_CUS": (0, _i, "am_train: CU budget of the forward recurrent launches (0 = whole device)"),
    "AM_BWD_CUS": (None, _opt_i, "am_train: CU budget of the BPTT launches (None = half the device in the fp32-class modes)"),
    # ---- streams
    "CHAIN_LANES": (False, _b, "experiment: the two chains on CU-masked streams (measured slower, DESIGN 4.3)"),
    "CHAIN_PRIO": (False, _b, "the acoustic chain's stream at the highest priority (measured slower)"),
    "WGRAD_LANE": ("", _s, "experiment: weight-gradient stream confined to CU half 0 / 1"),
    "WGRAD_PRIO": (True, _b, "weight-gradient stream at the lowest priority"),
    "WGRAD_EARLY": (True, _b, "weight-gradient products may start right behind the BPTT launch (beside the input-gradient GEMM)"),
    "WGRAD_WGS": (0, _i, "grid cap of the row-major weight-gradient GEMM (0 = none)"),
    # ---- kernel-path selection (ops.py)
    "LINEAR_DIRECT": (True, _b, "pointwise-linear / BatchNorm / conv parameter gradients accumulate straight into the flat buffers"),
    "PLANES_PRE": (True, _b, "fast modes: input projections on the plane GEMM"),
    "PLANES_BWD": (True, _b, "fast modes: input-gradient and weight-gradient products on the plane GEMMs"),
    "PLANES_EMIT": (True, _b, "fast modes: the BPTT kernels write d(gates) as operand planes"),
    "TN_FOLD": (False, _b, "fp32: per-utterance weights folded into the weight-gradient GEMM (measured slower)"),
    "CLASS_WGRAD": (True, _b, "per-utterance weights as one weight-gradient launch per utterance class with a device-scalar alpha"),
    "MULTI_WGRAD": (True, _b, "fp32: a layer's four weight-gradient products as one multi-problem launch (0: four launches)"),
    "TN_WGRAD": (True, _b, "fast modes: weight gradients from row-major planes (0: transposed planes)"),
    "MANAGED_XCHG": (True, _b, "managed exchange buffers: no poison memset launch in front of a persistent launch"),
    "FUSED_XPROJ": (True, _b, "LSTM forward launches form their input projection themselves where the library covers the shape (aas_lstm_fwd_x_ex)"),
    "FUSED_GLUE": (True, _b, "step prologue / raw-sum loss roots / controller launches instead of torch eager glue"),
    "WGRAD_MAXSTEPS": (None, _opt_i, "lifetime cap (k-steps) of the weight-gradient products alone (None: GEMM32_MAXSTEPS)"),
    "GEMM32_MAXSTEPS": (48, _i, "lifetime cap (k-steps) of a GEMM workgroup inside the training step (0 = none)"),
}

_values = {k: v[0] for k, v in _TABLE.items()}
_listeners = []     # callables(name, value): module-level mirrors (ops.PLANES_PRE[0] ...) and library setters
_warned = [False]


def _load_env():
    names = [k for k in _TABLE if ("AAS_" + k) in os.environ]
    if not names:
        return
    if os.environ.get("AAS_ABLATION", "0") != "1":
        if not _warned[0]:
            _warned[0] = True
            sys.stderr.write("[aas] ignoring %s: ablation switches are read from the environment only with AAS_ABLATION=1 "
                             "(aas_enhancement_amd/knobs.py)\n" % ", ".join("AAS_" + n for n in names))
        return
    for n in names:
        _values[n] = _TABLE[n][1](os.environ["AAS_" + n])


_load_env()


def get(name):
    return _values[name]


def set(name, value):   # noqa: A001  (module-level API: knobs.set)
    """Test / A-B API: turn one switch for the rest of the process (or until set back)."""
    if name not in _TABLE:
        raise KeyError("unknown knob %r" % (name,))
    _values[name] = value
    for fn in _listeners:
        fn(name, value)


def on_change(fn):
    _listeners.append(fn)


@contextlib.contextmanager
def override(**kw):
    """`with knobs.override(TWO_LANES="1", EBWD_CUS=96): ...` - previous values back on exit."""
    prev = {k: _values[k] for k in kw}
    try:
        for k, v in kw.items():
            set(k, v)
        yield
    finally:
        for k, v in prev.items():
            set(k, v)


def active():
    """{name: value} of every switch that is not at its shipped default."""
    return {k: v for k, v in _values.items() if v != _TABLE[k][0]}


def defaults():
    return {k: v[0] for k, v in _TABLE.items()}


def markdown_table():
    """The table above as markdown (docs/KNOBS.md is this function's output: `python -m aas_enhancement_amd.knobs > docs/KNOBS.md`)."""
    out = ["# Switches of the hot path (`aas_enhancement_amd/knobs.py`)", "",
           "Generated by `python -m aas_enhancement_amd.knobs`; `tests/test_abi.py` checks that this file matches the table.",
           "Read from the environment (`AAS_<NAME>=value`) ONLY together with `AAS_ABLATION=1`; tests use `knobs.override(...)`;",
           "`bench.py` reports no headline value while any switch is off its default (`--allow-ablation` labels an A/B line).",
           "Run configuration that IS read from the environment: `AAS_PRECISION=0|1|2`, `AAS_DP_FORCE=1`, `AAS_BENCH_FORCE_SPAWN=1`, `AAS_NO_PIN`.", "",
           "| switch | shipped default | what it does |", "|---|---|---|"]
    for k, (d, _, doc) in _TABLE.items():
        out.append("| `%s` | `%r` | %s |" % (k, d, doc))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    sys.stdout.write(markdown_table())
