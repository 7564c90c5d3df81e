"""CPU (-m "not gpu"): the C-ABI library builds/loads here and exports every symbol include/aas_hip.h declares;
the host logic that needs no GPU (collate layouts, decoder, config, error paths)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "aas_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aas_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from aas_enhancement_amd import _lib, build
    build.build(verbose=False)
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n
        assert n in _lib.SIGNATURES, "binding missing for " + n
    assert set(_lib.SIGNATURES) == set(names)
    assert _lib.lib().aas_version() == 1
    assert _lib.lib().aas_rnn_sync_bytes() >= 4096


def test_library_exports_the_warpctc_abi():
    """include/aas_warpctc.h: warp-ctc's four exported functions, under warp-ctc's names."""
    from aas_enhancement_amd import _lib, build
    build.build(verbose=False)
    L = ctypes.CDLL(_lib.LIB_PATH)
    src = open(os.path.join(ROOT, "include", "aas_warpctc.h")).read()
    for n in ("get_warpctc_version", "ctcGetStatusString", "compute_ctc_loss", "get_workspace_size"):
        assert n in src and hasattr(L, n), n
    L.ctcGetStatusString.restype = ctypes.c_char_p
    assert L.get_warpctc_version() == 2 and L.ctcGetStatusString(2) == b"invalid value"


def test_no_cpu_fallback():
    from aas_enhancement_amd import ops
    from aas_enhancement_amd.model import L1Loss_mask, stackedBRNN
    G = stackedBRNN(I=4, H=8, L=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        G(torch.zeros(1, 4, 5))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L1Loss_mask()(torch.zeros(1, 4, 5), torch.zeros(1, 4, 5), torch.zeros(1, 1, 5, dtype=torch.uint8))


def test_argument_validation_without_gpu():
    from aas_enhancement_amd import _lib
    L = _lib.lib()
    assert L.aas_gemm_f32(None, 7, 1, 1, 1, None, 1, None, 1, None, 1, None, None, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0) != 0
    assert b"bad mode" in L.aas_last_error()
    sz = ctypes.c_size_t(0)
    ll, al = np.array([4, 3], np.int32), np.array([15, 12], np.int32)
    assert L.aas_ctc_get_workspace_size(ll.ctypes.data, al.ctypes.data, 29, 2, 15, ctypes.byref(sz)) == 0
    assert sz.value >= 4 * 2 * (15 * 9 + 15)


def test_state_dict_keys_match_reference_goldens():
    import torch.nn as nn
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    from tests.helpers import LABELS, load, sub
    z = load("f1_aas_tiny.npz")
    G = stackedBRNN(I=8, H=16, L=4)
    A = DeepSpeech(nn.GRU, LABELS, 12, 5, True, 11, 2, 8, 2, nFreq=8)
    assert list(G.state_dict().keys()) == list(sub(z, "init.G.").keys())
    assert list(A.state_dict().keys()) == list(sub(z, "init.A.").keys())
    for k, v in A.state_dict().items():
        assert tuple(v.shape) == tuple(z["init.A." + k].shape), k
    pkg = DeepSpeech.serialize(A)
    B = DeepSpeech.load_model_package(pkg)
    assert B.nFreq == 8 and B.rnn_size == 12 and len(B.rnns) == 5


def test_collate_layouts():
    from aas_enhancement_amd.loader_functions import _collate_fn, _collate_fn_paired
    feats = [torch.randn(8, t) for t in (5, 9, 7)]
    out = _collate_fn([(f, [1, 2][: i + 1]) for i, f in enumerate(feats)])
    inputs, targets, pct, tsz, mask = out
    assert inputs.shape == (3, 8, 9) and mask.shape == (3, 1, 9) and mask.dtype == torch.uint8
    assert [int(m.sum()) for m in mask] == [0, 2, 4] and mask.n_valid == 21
    assert torch.allclose(pct, torch.tensor([1.0, 7 / 9.0, 5 / 9.0]))
    assert targets.dtype == torch.int32 and tsz.tolist() == [2, 2, 1]
    p = _collate_fn_paired([(f, [3], f * 2) for f in feats])
    assert len(p) == 6 and torch.equal(p[1][0], feats[1] * 2) and p[2].n_valid == 21


def test_greedy_decoder_and_wer():
    from aas_enhancement_amd.decoder import GreedyDecoder
    from tests.helpers import LABELS
    d = GreedyDecoder(LABELS)
    T, C = 8, len(LABELS)
    path = [2, 2, 0, 2, 28, 3, 3, 0]  # a a _ a ' ' b b _  -> "aa b"
    probs = torch.full((T, 1, C), -5.0)
    for t, k in enumerate(path):
        probs[t, 0, k] = 5.0
    out, _ = d.decode(probs, torch.tensor([T]))
    assert out[0][0] == "aa b"
    assert d.wer("the cat sat", "the cat sit") == 1 and d.cer("abc", "axc") == 1


def test_cli_flags_and_defaults_equal_the_references():
    """F12 (tools/make_goldens.py f12: the reference's own parsers replayed): every flag of AM_training/train.py:24-110 and of
    Speech_enhancement_by_AAS/config.py exists here with the SAME default - a reference command line trains the same model."""
    import json
    from aas_enhancement_amd import am_train
    from aas_enhancement_amd.config import get_config
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "f12_cli_defaults.json")))
    ours = vars(am_train.build_parser().parse_args([]))
    for k, v in ref["AM_training/train.py"].items():
        assert k in ours, "am_train lacks --" + k
        assert ours[k] == v, (k, ours[k], v)
    c, _ = get_config([])
    for k, v in ref["Speech_enhancement_by_AAS/config.py"].items():
        assert hasattr(c, k), "config lacks --" + k
        assert getattr(c, k) == v, (k, getattr(c, k), v)
    # what train.py:128-138 derives before it builds anything
    a = am_train.resolve_args(am_train.build_parser().parse_args(["--process_mel", "true", "--n_mels", "80", "--DB_name", "x", "--expnum", "3", "--gpu", "0"]))
    assert a.nFreq == 80 and a.model_path == "models/x_3_final.pth.tar"
    with pytest.raises(NotImplementedError):
        am_train.resolve_args(am_train.build_parser().parse_args(["--arch_ver", "ResidualDeepSpeech"]))
    with pytest.raises(NotImplementedError):
        am_train.resolve_args(am_train.build_parser().parse_args(["--preprocess", "code"]))


def test_ablation_switches_are_not_read_from_the_environment_by_default():
    """knobs.py: AAS_<NAME> variables take effect only together with AAS_ABLATION=1 (checked in child interpreters)."""
    import subprocess
    import sys
    code = "from aas_enhancement_amd import knobs; print(knobs.get('SKIP_WGRAD'), knobs.get('EBWD_CUS'), sorted(knobs.active()))"
    env = dict(os.environ, AAS_SKIP_WGRAD="1", AAS_EBWD_CUS="96")
    env.pop("AAS_ABLATION", None)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.stdout.split()[:2] == ["False", "128"] and "[]" in r.stdout and "ignoring" in r.stderr, (r.stdout, r.stderr)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(env, AAS_ABLATION="1"), capture_output=True, text=True)
    assert r.stdout.split()[:2] == ["True", "96"] and "SKIP_WGRAD" in r.stdout, (r.stdout, r.stderr)
    from aas_enhancement_amd import knobs
    assert knobs.active() == {}
    with knobs.override(TWO_LANES="1"):
        assert knobs.active() == {"TWO_LANES": "1"}
    assert knobs.active() == {}


def test_bench_launcher_builds_the_torchrun_command_and_relays(monkeypatch, capsys):
    """bench.spawn_ranks (CPU, no GPU touched): fewer devices than ranks -> exit code 3 and a message; enough devices -> one child
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`, rank 0's JSON
    line relayed on stdout, the child's exit code returned."""
    import subprocess
    import sys
    import types
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert bench.spawn_ranks(2) == 3
    assert "only 1 device" in capsys.readouterr().err
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7, stdout='RCCL banner\n{"metric": "m", "value": 1.0, "n_gpus": 4}\n')
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    rc = bench.spawn_ranks(4)
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 7 and out == ['{"metric": "m", "value": 1.0, "n_gpus": 4}']
    c = seen["cmd"]
    assert c[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and c[c.index("--nproc-per-node") + 1] == "4"
    assert c[c.index("--master-addr") + 1] == "127.0.0.1" and c[-4:] == ["--gpus", "4", "--steps", "3"] and c[-5].endswith("bench.py")
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"


def test_knobs_document_is_generated_from_the_table():
    """docs/KNOBS.md is `python -m aas_enhancement_amd.knobs` verbatim: the README's switch list cannot drift from knobs.py."""
    from aas_enhancement_amd import knobs
    path = os.path.join(ROOT, "docs", "KNOBS.md")
    assert open(path).read() == knobs.markdown_table()
    assert all(len(v) == 3 and isinstance(v[2], str) and v[2] for v in knobs._TABLE.values())


def test_aaslaunch_struct_matches_the_header_and_scopes_nest():
    """`ops._CLaunch` (what the host side hands to the `*_ex` entry points) has exactly the fields of `struct aasLaunch` in
    include/aas_hip.h, in order; `aas_launch_scope` installs / returns the previous scope per thread and refuses a struct of another
    size - no GPU needed (the call only records a pointer)."""
    import ctypes
    import re
    import threading
    from aas_enhancement_amd import _lib, ops
    src = open(os.path.join(ROOT, "include", "aas_hip.h")).read()
    body = re.search(r"typedef struct aasLaunch \{(.*?)\} aasLaunch;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\bint\s+(\w+)\s*;", body)
    assert fields == [n for n, _ in ops._CLaunch._fields_] and all(t is ctypes.c_int for _, t in ops._CLaunch._fields_)
    L = _lib.lib()
    a, b = ops._new_claunch(), ops._new_claunch()
    assert a.size == ctypes.sizeof(ops._CLaunch) == 4 * len(fields)
    prev = ctypes.c_void_p()
    assert L.aas_launch_scope(ctypes.byref(a), ctypes.byref(prev)) == 0 and prev.value is None
    assert L.aas_launch_scope(ctypes.byref(b), ctypes.byref(prev)) == 0 and prev.value == ctypes.addressof(a)
    a.debug_flags = 512                                  # scope b is installed: a's fields are not read
    b.debug_flags = 64
    assert L.aas_get_debug_flags() == 64
    seen = []
    t = threading.Thread(target=lambda: seen.append(L.aas_get_debug_flags()))    # another thread: no scope there
    t.start(); t.join()
    assert seen == [0]
    assert L.aas_launch_scope(None, ctypes.byref(prev)) == 0 and prev.value == ctypes.addressof(b)
    assert L.aas_get_debug_flags() == 0
    bad = ops._new_claunch()
    bad.size = 12
    assert L.aas_launch_scope(ctypes.byref(bad), None) != 0 and b"aasLaunch.size" in L.aas_last_error()
    assert L.aas_launch_scope(None, None) == 0
    # the Python wrapper: nested launch_state blocks restore what was installed before them
    s1, s2 = ops.LaunchState(debug_flags=512), ops.LaunchState(precision=1)
    assert ops.state() is ops._PROCESS
    with ops.launch_state(s1):
        assert ops.state() is s1 and L.aas_get_debug_flags() == 512
        with ops.launch_state(s2):
            assert ops.state() is s2 and ops.get_precision() == 1 and L.aas_get_debug_flags() == 0
        assert ops.state() is s1 and L.aas_get_debug_flags() == 512
    assert ops.state() is ops._PROCESS and L.aas_get_debug_flags() == 0
