"""GPU parity tests (-m gpu) added in round 6: the validation passes of the minimize_DCE / FSEGAN / AAS trainers against the
reference's modules + decoder (fixture F13), and the two trainers' `train()` loops over `.pt7` files across two save_iters -
checkpoint files, rotation, best-WER copy, resume - with their validation numbers checked against the CPU oracle."""
import os
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from tests.helpers import LABELS, load, load_sd, rel_err, sub
from tests.test_gpu_round2 import _fill, _write_manifest
from tests.test_gpu_step import cfg

pytestmark = pytest.mark.gpu

REL_OUT, REL_LOSS = 1e-3, 1e-2


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _f13_nets(z):
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    G, Dp, Da = stackedBRNN(I=8, H=16, L=4), stackedBRNN(I=16, O=8, H=16, L=4), stackedBRNN(I=8, H=16, L=4)
    A = DeepSpeech(nn.GRU, LABELS, 12, 3, True, 11, 2, 8, 2, nFreq=8)
    for nm, m in (("G", G), ("Dp", Dp), ("Da", Da), ("A", A)):
        load_sd(m, sub(z, "init.%s." % nm))
    return G, Dp, Da, A


def _f13_batches(z):
    g = lambda k: torch.from_numpy(np.asarray(z[k]))
    return [(g("b%d.inputs" % b), g("b%d.cleans" % b), g("b%d.mask" % b), g("b%d.targets" % b), g("b%d.pct" % b), g("b%d.target_sizes" % b))
            for b in range(2)]


def test_validation_functions_of_the_three_trainers_vs_reference(gpu, precision2):
    """F13: `greedy_decoding` + DCE (trainer_DCE.py:139-153,209-250), `greedy_decoding_and_FSEGAN` (trainer_FSEGAN.py:277-317) and
    `greedy_decoding_and_AAS` (trainer_AAS.py:301-351) of the product trainers against the tuples the reference's own modules and
    decoder produce, batch by batch, and the AverageMeter results of the loops; decoded strings equal."""
    from aas_enhancement_amd.trainer_AAS import Trainer as AAS
    from aas_enhancement_amd.trainer_DCE import Trainer as DCE
    from aas_enhancement_amd.trainer_FSEGAN import Trainer as FSEGAN
    z = load("f13_validation.npz")
    w_adv, w_ac = float(z["w_adversarial"]), float(z["w_acoustic"])
    G, Dp, Da, A = _f13_nets(z)
    dl = types.SimpleNamespace(labels=LABELS)
    c = cfg(w_adversarial=w_adv, w_acoustic=w_ac)
    t_dce, t_fse, t_aas = DCE(c, dl, models=(G, A)), FSEGAN(c, dl, models=(G, Dp, A)), AAS(c, dl, models=(G, Da, A))
    G.eval()
    batches = _f13_batches(z)
    from aas_enhancement_amd.utils import AverageMeter
    m = {k: AverageMeter() for k in ("d.dce", "d.wer", "d.cer", "f.dce", "f.adv", "f.wer", "f.cer", "a.ctc", "a.adv", "a.wer", "a.cer")}
    with torch.no_grad():
        for b, bt in enumerate(batches):
            p = "b%d." % b
            inputs, cleans, mask, targets, pct, tsz = bt
            # --- minimize_DCE
            dce, nEl, wer, cer, nW, nC = t_dce._validate_batch(bt)
            assert float(dce) == pytest.approx(float(z[p + "dce.dce"]), rel=REL_LOSS) and nEl == int(z[p + "dce.nElement"])
            assert (wer, cer, nW, nC) == pytest.approx((float(z[p + "dce.wer"]), float(z[p + "dce.cer"]), int(z[p + "dce.nWord"]), int(z[p + "dce.nChar"])))
            assert t_dce.greedy_decoding(inputs, targets, pct, tsz) == pytest.approx((wer, cer, nW, nC))
            m["d.dce"].update(float(dce), nEl); m["d.wer"].update(wer, nW); m["d.cer"].update(cer, nC)
            # the enhanced features / logits / strings behind those numbers
            enh = t_dce.G(inputs.cuda())
            prob, sizes, *_ = t_dce._greedy_pass(enh, targets, pct, tsz)
            assert rel_err(enh, z[p + "enhanced"]) < REL_OUT and rel_err(prob, z[p + "logits_tnc"]) < REL_OUT
            assert sizes.tolist() == z[p + "sizes"].tolist()
            strings, _ = t_dce.decoder.decode(prob, sizes)
            for i, s_ in enumerate(strings):
                assert s_[0] == bytes(z[p + "decoded%d" % i]).decode("utf8"), (b, i)
            # --- FSEGAN
            t = t_fse.greedy_decoding_and_FSEGAN(inputs, cleans, targets, pct, tsz, mask)
            for k, v in zip(("dce", "l_adv_ny", "nElement", "wer", "cer", "total_word", "total_char"), t):
                assert float(v) == pytest.approx(float(z[p + "fsegan." + k]), rel=REL_LOSS), (b, k)
            m["f.dce"].update(float(t[0]), t[2]); m["f.adv"].update(float(t[1]), t[2]); m["f.wer"].update(t[3], t[5]); m["f.cer"].update(t[4], t[6])
            # --- AAS
            t = t_aas.greedy_decoding_and_AAS(inputs, targets, pct, tsz, mask)
            for k, v in zip(("l_CTC", "l_adv_ny", "nElement", "wer", "cer", "total_word", "total_char"), t):
                assert float(v) == pytest.approx(float(z[p + "aas." + k]), rel=REL_LOSS), (b, k)
            m["a.ctc"].update(float(t[0]), inputs.size(0)); m["a.adv"].update(float(t[1]), t[2]); m["a.wer"].update(t[3], t[5]); m["a.cer"].update(t[4], t[6])
    for ours, ref in (("d.dce", "dce.dce"), ("d.wer", "dce.wer"), ("d.cer", "dce.cer"), ("f.dce", "fsegan.dce"), ("f.adv", "fsegan.adv_ny"),
                      ("f.wer", "fsegan.wer"), ("f.cer", "fsegan.cer"), ("a.ctc", "aas.ctc"), ("a.adv", "aas.adv_ny"), ("a.wer", "aas.wer"), ("a.cer", "aas.cer")):
        assert m[ours].avg == pytest.approx(float(z["avg." + ref]), rel=REL_LOSS), ours


def _loader(tmp, paired=True):
    from aas_enhancement_amd.data_loader import DataLoader
    np.random.seed(11)
    tr = _write_manifest(tmp, [64, 60, 58, 52, 50, 47, 44, 41], paired=paired, seed=10)
    sub_ = _write_manifest(tmp, [61, 55, 43], paired=paired, seed=40)
    val = _write_manifest(tmp, [66, 59, 51, 45, 40], paired=paired, seed=70)
    return DataLoader(batch_size=3, paired=paired, tr_ny_manifest=tr, trsub_manifest=sub_, val_manifest=val, labels=LABELS, num_workers=0, pin_memory=True), sub_, val


def _host_batches(manifest, paired=True, batch_size=3):
    """The evaluation batches the loader hands out for `manifest` (sequential, batch_size rows, the paired collate), on the host."""
    from aas_enhancement_amd import loader_functions as LF
    ds = LF.FeatDataset(manifest, LABELS)
    out = []
    for i in range(0, len(ds), batch_size):
        items = [ds[j] for j in range(i, min(i + batch_size, len(ds)))]
        out.append(LF._collate_fn_paired(items) if paired else LF._collate_fn(items))
    return out


def _oracle_nets(G, A, D=None):
    from oracle import ref_model as RM
    g = RM.RefStackedBRNN(8, 8, 16, 4)
    a = RM.RefDeepSpeech(nn.GRU, LABELS, 12, 3, 11, 2, 8, 2, nFreq=8)
    g.load_state_dict({k: v.cpu() for k, v in G.items()} if isinstance(G, dict) else {k: v.cpu() for k, v in G.state_dict().items()})
    a.load_state_dict({k: v.cpu() for k, v in A.state_dict().items()})
    g.eval()
    if D is None:
        return g, a
    d = RM.RefStackedBRNN(16, 8, 16, 4)
    d.load_state_dict({k: v.cpu() for k, v in D.state_dict().items()})
    return g, a, d


@pytest.mark.parametrize("which", ["minimize_DCE", "FSEGAN"])
def test_train_loop_validates_checkpoints_and_resumes(gpu, tmp_path, which, capsys):
    """`Trainer.train()` of the two trainers over `.pt7` manifests, max_iter 4 / save_iter 2 / log_iter 1 (trainer_DCE.py:111-207,
    trainer_FSEGAN.py:125-275): the log lines of the reference, `G_<iter>.pth` rotation (the previous file removed), `G_valmin_<iter>.pth`
    on the best validation WER, the AverageMeters of the last validation pass equal to the CPU oracle's pass (oracle/ref_step
    dce_validation / fsegan_validation, pinned by F13) over the same files with the saved G, and `--load_path` resume."""
    from oracle import ref_step as RS
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    tmp = str(tmp_path)
    dl, sub_man, val_man = _loader(tmp)
    G = _fill(stackedBRNN(I=8, H=16, L=4), 501)
    A = _fill(DeepSpeech(nn.GRU, LABELS, 12, 3, True, 11, 2, 8, 2, nFreq=8), 503, 0.1)
    with torch.no_grad():
        A.fc[0].module[1].weight.mul_(6.0)
    c = cfg(lr=1e-3, max_iter=4, save_iter=2, log_iter=1, start_iter=0, expnum="x", w_adversarial=0.01, write_log=True)
    if which == "minimize_DCE":
        from aas_enhancement_amd.trainer_DCE import Trainer
        tr = Trainer(c, dl, models=(G, A))
    else:
        from aas_enhancement_amd.trainer_FSEGAN import Trainer
        D = _fill(stackedBRNN(I=16, O=8, H=16, L=4), 502)
        tr = Trainer(c, dl, models=(G, D, A))
    tr.model_dir = os.path.join(tmp, "exp")
    tr._open_log()
    tr.train()
    torch.cuda.synchronize()
    files = sorted(os.listdir(tr.model_dir))
    assert "G_3.pth" in files and "G_1.pth" not in files and "log.txt" in files          # rotation: the previous G_<iter>.pth is removed
    best = [f for f in files if f.startswith("G_valmin_")]
    assert len(best) == 1 and best[0] == "G_valmin_%d.pth" % tr.valmin_iter and tr.valmin_iter in (1, 3)
    assert tr.G.loss_stop <= tr.wer_val.avg + 1e-12
    tr.logFile.close()
    log = open(os.path.join(tr.model_dir, "log.txt")).read().splitlines()
    if which == "minimize_DCE":
        assert sum(l.startswith("[") and "(train) DCE:" in l for l in log) == 4
        for it in (1, 3):
            for name in ("training subset", "validation"):
                assert any(l.startswith("[%d/4] (%s) DCE: " % (it, name)) for l in log)
                assert any(l.startswith("[%d/4] (%s) WER: " % (it, name)) and ", CER: " in l for l in log)
    else:
        assert sum("(train) DCE:" in l and "ADV_cl:" in l and "ADV_ny:" in l for l in log) == 4
        assert sum("(train) conv_measure:" in l and "kt:" in l for l in log) == 4
        for it in (1, 3):
            for name in ("training subset", "validation"):
                assert any(l.startswith("[%d/4] (%s) CTC: " % (it, name)) and "WER: " in l and "CER: " in l for l in log)
    # ---- the last validation pass against the oracle over the same files, with the G that was saved at iteration 3
    sd = torch.load(os.path.join(tr.model_dir, "G_3.pth"))
    for k, v in tr.G.state_dict().items():
        assert torch.equal(v.cpu(), sd[k].cpu()), k
    if which == "minimize_DCE":
        g, a = _oracle_nets(sd, tr.ASR)
        for man, dm, wm, cm in ((sub_man, tr.dce_tr, tr.wer_tr, tr.cer_tr), (val_man, tr.dce_val, tr.wer_val, tr.cer_val)):
            ref = RS.dce_validation(g, a, LABELS, _host_batches(man))
            assert dm.avg == pytest.approx(ref["dce"], rel=REL_LOSS)
            assert wm.avg == pytest.approx(ref["wer"], abs=1e-9) and cm.avg == pytest.approx(ref["cer"], abs=1e-9)
    else:
        g, a, d = _oracle_nets(sd, tr.ASR, tr.D)
        for man, ms in ((sub_man, (tr.dce_tr, tr.adv_ny_tr, tr.wer_tr, tr.cer_tr)), (val_man, (tr.dce_val, tr.adv_ny_val, tr.wer_val, tr.cer_val))):
            ref = RS.fsegan_validation(g, d, a, LABELS, _host_batches(man), c.w_adversarial)
            assert ms[0].avg == pytest.approx(ref["dce"], rel=REL_LOSS) and ms[1].avg == pytest.approx(ref["adv_ny"], rel=REL_LOSS)
            assert ms[2].avg == pytest.approx(ref["wer"], abs=1e-9) and ms[3].avg == pytest.approx(ref["cer"], abs=1e-9)
    # ---- resume (--load_path): G from the newest G_valmin checkpoint, start_iter taken from its name
    c2 = cfg(lr=1e-3, max_iter=4, save_iter=2, log_iter=1, start_iter=-1, load_path=tr.model_dir, w_adversarial=0.01)
    G2 = _fill(stackedBRNN(I=8, H=16, L=4), 777)
    if which == "minimize_DCE":
        tr2 = Trainer(c2, dl, models=(G2, A))
    else:
        tr2 = Trainer(c2, dl, models=(G2, _fill(stackedBRNN(I=16, O=8, H=16, L=4), 778), A))
    assert c2.start_iter == tr.valmin_iter
    sdv = torch.load(os.path.join(tr.model_dir, best[0]))
    for k, v in tr2.G.state_dict().items():
        assert torch.equal(v.cpu(), sdv[k].cpu()), k
    # an explicit start_iter names the checkpoint itself
    c3 = cfg(lr=1e-3, start_iter=tr.valmin_iter, load_path=tr.model_dir)
    G3 = _fill(stackedBRNN(I=8, H=16, L=4), 779)
    tr3 = Trainer(c3, dl, models=(G3, A) if which == "minimize_DCE" else (G3, _fill(stackedBRNN(I=16, O=8, H=16, L=4), 780), A))
    assert c3.start_iter == tr.valmin_iter
    assert torch.equal(tr3.G.state_dict()["first_linear.weight"].cpu(), sdv["first_linear.weight"].cpu())
    with pytest.raises(AssertionError):
        Trainer(cfg(load_path=os.path.join(tmp, "nowhere"), start_iter=-1), dl, models=(G2, A) if which == "minimize_DCE" else (G2, _fill(stackedBRNN(I=16, O=8, H=16, L=4), 781), A))


def test_dce_and_fsegan_validation_need_the_acoustic_model(gpu):
    """The two trainers may be built without A for benchmarks of the step alone; their save_iter block then fails loudly."""
    from aas_enhancement_amd.model import stackedBRNN
    from aas_enhancement_amd.trainer_DCE import Trainer as DCE
    from aas_enhancement_amd.trainer_FSEGAN import Trainer as FSEGAN
    G, D = _fill(stackedBRNN(I=8, H=16, L=4), 1), _fill(stackedBRNN(I=16, O=8, H=16, L=4), 2)
    for tr in (DCE(cfg(), None, models=(G,)), FSEGAN(cfg(), None, models=(G, D))):
        with pytest.raises(RuntimeError, match="acoustic model"):
            tr.validate_and_checkpoint(0)


# ---- launch parameters are arguments / per-thread scopes: two host threads, two trainers ------------------------------------------
def test_two_host_threads_two_trainers_bit_equal_to_each_alone(gpu):
    """Two host threads drive two trainers at the same time - one on ragged noisy / clean pairs (row classes inside D's recurrent
    launches, F1r), one on equal-length pairs in another arithmetic mode and with other kernel-selection bits (F1) - each on a
    stream of its own, through the device-resident step.  Every scalar and every parameter equals what the same trainer produces
    alone (bit for bit in all but one run of ~ 15; never beyond the last bits - see the bound at the end): no launch of one thread runs under the other's row classes, CU budget, tag, mode or flags (they travel
    as the aasLaunch argument / the thread's launch scope), and nothing is left behind in the process settings."""
    import threading
    from aas_enhancement_amd import ops
    from aas_enhancement_amd._lib import lib
    from aas_enhancement_amd.trainer_AAS import Trainer
    from tests.helpers import batch_from
    from tests.test_gpu_step import build_tiny
    zr, ze = load("f1r_aas_tiny_ragged_pair.npz"), load("f1_aas_tiny.npz")
    spec = [dict(z=zr, precision=0, flags=0), dict(z=ze, precision=1, flags=512)]
    STEPS = 6

    def make(i):
        z = spec[i]["z"]
        tr = Trainer(cfg(lr=float(z["cfg_lr"])), None, models=build_tiny(z))
        tr.kt = float(z["kt0"])
        tr.set_precision(spec[i]["precision"])
        tr.launch = ops.LaunchState(debug_flags=spec[i]["flags"])
        return tr

    def drive(i, tr, out, start=None, errs=None):
        try:
            z = spec[i]["z"]
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                if start is not None:
                    start.wait()
                rows = []
                for it in range(STEPS):
                    ny, cl = batch_from(z, "it%d.ny." % (it % 3)), batch_from(z, "it%d.cl." % (it % 3))
                    r = tr.train_step_async(ny, cl, it)
                    if i == 0:
                        assert tr._last_schedule == "batched-ragged"
                    sc = tr.read_scalars()
                    rows.append([sc[k] for k in ("l_adv_ny_G", "l_adv_cl", "l_ctc", "kt")] + [float(r["enhanced"].detach().double().sum()), float(r["prob"].detach().double().sum())])
                st.synchronize()
            out[i] = (np.asarray(rows), {k: v.detach().clone() for m in (tr.G, tr.D, tr.ASR) for k, v in m.state_dict().items()})
        except BaseException as e:  # noqa: BLE001
            if errs is not None:
                errs.append((i, repr(e)))
            raise
    before = (ops.get_precision(), int(lib().aas_get_gemm_max_steps()), int(lib().aas_get_debug_flags()))
    alone = {}
    for i in (0, 1):
        drive(i, make(i), alone)
    both, errs = {}, []
    trs = [make(0), make(1)]
    start = threading.Barrier(2)
    ths = [threading.Thread(target=drive, args=(i, trs[i], both, start, errs)) for i in (0, 1)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    torch.cuda.synchronize()
    assert not errs, errs
    assert not ops.rnn_timeout_flag()
    assert (ops.get_precision(), int(lib().aas_get_gemm_max_steps()), int(lib().aas_get_debug_flags())) == before
    # "Bit for bit" up to the arrival order of the library's few atomic accumulations (the fp64 loss sums, CTC's per-label occupancies,
    # split-K of the thin products): one full-suite run in ~ 15 differs in a last bit there.  A launch that ran under the other thread's
    # row classes, mode or flags differs by 1e-4 ... 1: the bound below is 3e-6 of the largest element.
    for i in (0, 1):
        assert np.allclose(alone[i][0], both[i][0], rtol=3e-6, atol=0.0), (i, alone[i][0] - both[i][0])
        for k, v in alone[i][1].items():
            w = both[i][1][k]
            if not v.dtype.is_floating_point or torch.equal(v, w):
                assert torch.equal(v, w), (i, k)
                continue
            # (Adam turns a last-bit difference in a gradient that is rounding noise into +- lr on that element: a handful of elements
            #  may move by a few lr, everything else stays within the bound)
            far = ((v - w).abs() > 3e-6 * v.abs().max()).float().mean().item()
            assert far < 2e-3 and float((v - w).abs().max()) <= 2.5 * STEPS * float(spec[i]["z"]["cfg_lr"]), (i, k, far)


def test_launch_argument_entry_points_through_ctypes(gpu):
    """aas_lstm_fwd_ex with an aasLaunch built by hand (what a C caller writes): row classes are taken from the struct and consumed,
    the plain entry point right after sees none; a struct of the wrong size is refused; a scope installed with aas_launch_scope
    applies to the plain entry point and is gone after it is removed."""
    import ctypes
    from aas_enhancement_amd import _lib, ops
    L = _lib.lib()
    dev = torch.device("cuda:0")
    H, T, N = 32, 12, 4
    g = torch.Generator().manual_seed(3)
    w_hh, w_hr = [((torch.rand(4 * H, H, generator=g) - 0.5) * 0.3).to(dev) for _ in range(2)]
    pre = torch.randn(T, N, 2, 4 * H, generator=g).to(dev)
    sync = torch.zeros(int(L.aas_rnn_sync_bytes()), dtype=torch.uint8, device=dev)
    xchg = torch.empty(int(L.aas_rnn_xchg_bytes(T, N, H, 4)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run(fn, *extra):
        hout, gact, cst = (torch.empty(2, T, N, H, device=dev), torch.empty(2, T, N, 4 * H, device=dev), torch.empty(2, T, N, H, device=dev))
        rc = fn(st, T, N, H, pre.data_ptr(), w_hh.data_ptr(), w_hr.data_ptr(), hout.data_ptr(), gact.data_ptr(), cst.data_ptr(), sync.data_ptr(),
                xchg.data_ptr(), *extra)
        torch.cuda.synchronize()
        return rc, hout
    la = ops._new_claunch()
    la.cls_n_first, la.cls_T_first, la.cls_T_rest, la.rnn_tag = 2, T, 5, 7
    rc, y1 = run(L.aas_lstm_fwd_ex, ctypes.byref(la))
    assert rc == 0 and la.cls_n_first == -1                       # consumed
    rc, y2 = run(L.aas_lstm_fwd)
    assert rc == 0
    assert float(y1[:, 5:, 2:].abs().max()) == 0.0 and float(y2[:, 5:, 2:].abs().max()) > 0.0 and torch.equal(y1[:, :, :2], y2[:, :, :2])
    rc, y3 = run(L.aas_lstm_fwd_ex, None)                          # NULL = the plain entry point
    assert rc == 0 and torch.equal(y3, y2)
    bad = ops._new_claunch()
    bad.size = 8
    rc, _ = run(L.aas_lstm_fwd_ex, ctypes.byref(bad))
    assert rc != 0 and b"aasLaunch.size" in L.aas_last_error()
    # a thread scope applies to the plain entry points
    la.cls_n_first, la.cls_T_first, la.cls_T_rest = 2, T, 5
    assert L.aas_launch_scope(ctypes.byref(la), None) == 0
    try:
        rc, y4 = run(L.aas_lstm_fwd)
    finally:
        assert L.aas_launch_scope(None, None) == 0
    assert rc == 0 and torch.equal(y4, y1)
    rc, y5 = run(L.aas_lstm_fwd)
    assert rc == 0 and torch.equal(y5, y2)
    # nn.RNN launches refuse pending row classes instead of leaking them into the next lstm / gru launch
    assert L.aas_set_rnn_row_classes(2, T, 5) == 0
    hout, gact = torch.empty(2, T, N, H, device=dev), torch.empty(2, T, N, 4 * H, device=dev)
    pre1 = torch.randn(T, N, 2, H, generator=g).to(dev)
    rc = L.aas_rnn_fwd(st, T, N, H, pre1.data_ptr(), w_hh.data_ptr(), w_hr.data_ptr(), hout.data_ptr(), gact.data_ptr(), sync.data_ptr())
    assert rc != 0 and b"row classes" in L.aas_last_error()
    rc, y6 = run(L.aas_lstm_fwd)
    assert rc == 0 and torch.equal(y6, y2)


@pytest.mark.parametrize("Tn,Tc", [(521, 454)])
def test_long_utterances_step_vs_live_fp64_oracle(gpu, Tn, Tc):
    """5 s utterances (the fixtures stop at T = 200; 731 / 640 and 667 / 667 pass the same way, 50-60 s of CPU each): one AAS step of
    small networks - synchronous and device-resident, ragged pair - against the CPU oracle run here IN FP64 on the same inputs (trainer_AAS.py:131-194).  The exchange tags / ring slots
    of the persistent kernels, the CTC kernel's 64-frame passes (T' = 260) and the row classes of the ragged batched pass all wrap
    or repeat many times at this length.  Why fp64: at T > 1000 the fp32 CPU oracle itself sits 0.9e-3 ... 2.5e-3 from its fp64 run
    in the gradients that pass through CTC (torch's fp32 CPU ctc_loss over hundreds of frames), the product 2e-6 ... 7e-6
    (tools/probe/long_step_err.py) - so the tolerances here are 1e-4, not the fixtures' 1e-3 / 1e-2."""
    import copy
    from aas_enhancement_amd import ops, prng
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    from aas_enhancement_amd.trainer_AAS import Trainer
    from oracle import ref_model as RM
    from oracle import ref_step as RS
    from tests.helpers import NOISE_PARAMS
    F, H, HA, M = 8, 16, 12, 8
    nets_ref = (RM.RefStackedBRNN(F, F, H, 4), RM.RefStackedBRNN(F, F, H, 4), RM.RefDeepSpeech(nn.GRU, LABELS, HA, 3, 11, 2, M, 2, nFreq=F))
    for i, r in enumerate(nets_ref):
        w = prng.fill_state_dict(r.state_dict(), 70 + i, conv_std=0.1 if i == 2 else None)
        sd = r.state_dict()
        for k, v in w.items():
            sd[k].copy_(torch.from_numpy(v))

    def gpu_nets():
        nets = (stackedBRNN(I=F, H=H, L=4), stackedBRNN(I=F, H=H, L=4), DeepSpeech(nn.GRU, LABELS, HA, 3, True, 11, 2, M, 2, nFreq=F))
        for r, g in zip(nets_ref, nets):
            g.load_state_dict(r.state_dict())
        return nets
    nets64 = tuple(copy.deepcopy(m).double() for m in nets_ref)
    N, L = 3, 20
    lens_n, lens_c = [Tn, Tn - 111, Tn - 340], [Tc, Tc - 97, Tc - 255]

    def batch(seed, T, lens, labelled, dt):
        x = prng.uniform(seed, (N, F, T), 0.0, 6.0)
        mask = np.zeros((N, 1, T), dtype=np.uint8)
        for n, l in enumerate(lens):
            x[n, :, l:] = 0.0
            mask[n, 0, l:] = 1
        m = torch.from_numpy(mask)
        if not labelled:
            return (torch.from_numpy(x).to(dt), None, None, None, m)
        return (torch.from_numpy(x).to(dt), torch.from_numpy(prng.randint(seed + 1, (N * L,), 1, 28).astype(np.int32)),
                torch.tensor([l / float(T) for l in lens]), torch.full((N,), L, dtype=torch.int32), m)
    ny, cl = batch(11, Tn, lens_n, True, torch.float32), batch(13, Tc, lens_c, False, torch.float32)
    c = cfg(lr=1e-3, allow_ASR_update_iter=0)
    kt0 = 0.3
    t_sync, t_dev = Trainer(c, None, models=gpu_nets()), Trainer(c, None, models=gpu_nets())
    t_sync.kt = t_dev.kt = kt0
    r = t_sync.train_step(ny, cl, 1, log_norms=True)
    t_dev.train_step_async(ny, cl, 1)
    rd = t_dev.read_scalars()
    torch.cuda.synchronize()
    assert not ops.rnn_timeout_flag()
    rc = RS.StepConfig(lr=1e-3)
    opts = [RS.make_optim(m, rc) for m in nets64]
    _, ref = RS.aas_step(nets64[0], nets64[1], nets64[2], opts[0], opts[1], opts[2], batch(11, Tn, lens_n, True, torch.float64),
                         batch(13, Tc, lens_c, False, torch.float64), rc, kt0, 1)
    tol = 1e-4
    for k in ("l_adv_ny_G", "l_adv_cl", "l_ctc", "kt"):
        assert abs(r[k] - ref[k]) <= tol * abs(ref[k]) + 1e-6, (k, r[k], ref[k])
        assert abs(rd[k] - ref[k]) <= tol * abs(ref[k]) + 1e-6, ("device-resident", k, rd[k], ref[k])
    for k in ("g_adv", "g_ctc_adv"):
        assert abs(r[k] - ref[k]) <= tol * abs(ref[k]), (k, r[k], ref[k])
    assert rel_err(r["prob"], ref["logits"]) < tol
    assert rel_err(r["enhanced"], ref["enhanced"]) < tol
    # every parameter gradient of the three networks, both product paths (the oracle's .grad is what its optimisers stepped on)
    for tr_ in (t_sync, t_dev):
        for nm, net, rn in zip("GDA", (tr_.G, tr_.D, tr_.ASR), nets64):
            ref_g = {k: v.grad for k, v in rn.named_parameters()}
            for k, v in net.named_parameters():
                if nm == "A" and k in NOISE_PARAMS:
                    continue
                assert rel_err(v.grad, ref_g[k]) < tol, (nm, k)


@pytest.mark.parametrize("T,N,H,I", [(200, 30, 500, 500), (33, 30, 500, 500), (61, 7, 96, 96), (40, 30, 128, 64), (17, 5, 100, 200), (5, 2, 12, 8),
                                     (1, 3, 16, 16), (3, 8, 256, 256), (90, 12, 500, 80)])
def test_lstm_forward_with_the_input_projection_inside_vs_fp64_and_vs_gemm_plus_launch(gpu, T, N, H, I):
    """aas_lstm_fwd_x_ex through ctypes - nn.LSTM's forward for one bias-free bidirectional layer in ONE launch (model.py:73-74,83) -
    against an fp64 recurrence on the CPU and against the two-launch form (aas_gemm_f32 + aas_lstm_fwd_ex): h, the saved gate values
    and c, at config-2 size, for I != H, ragged k / row tails, T = 1, and with two row classes (a shorter second class)."""
    import ctypes
    from aas_enhancement_amd import _lib, ops
    L = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(T * 1000 + N * 10 + H)
    x = (torch.randn(T, N, I, generator=g) * 0.5)
    w = [((torch.rand(4 * H, k, generator=g) - 0.5) * (2.0 / k ** 0.5)) for k in (I, H, I, H)]     # w_ih, w_hh, w_ih_rev, w_hh_rev
    xd, wd = x.to(dev), [t.to(dev) for t in w]
    sync = torch.zeros(int(L.aas_rnn_sync_bytes()), dtype=torch.uint8, device=dev)
    xchg = torch.empty(int(L.aas_rnn_xchg_bytes(T, N, H, 4)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def fused(la):
        out = (torch.empty(2, T, N, H, device=dev), torch.empty(2, T, N, 4 * H, device=dev), torch.empty(2, T, N, H, device=dev))
        rc = L.aas_lstm_fwd_x_ex(st, T, N, H, I, xd.data_ptr(), wd[0].data_ptr(), wd[2].data_ptr(), wd[1].data_ptr(), wd[3].data_ptr(),
                                 out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), sync.data_ptr(), xchg.data_ptr(), la)
        torch.cuda.synchronize()
        return rc, out

    def two_launches(la):
        pre = torch.empty(T, N, 2, 4 * H, device=dev)
        for d_ in range(2):
            ops.gemm(ops.NT, T * N, 4 * H, I, xd.view(T * N, I), I, wd[2 * d_], I, pre, 8 * H, c_off=4 * H * d_)
        out = (torch.empty(2, T, N, H, device=dev), torch.empty(2, T, N, 4 * H, device=dev), torch.empty(2, T, N, H, device=dev))
        rc = L.aas_lstm_fwd_ex(st, T, N, H, pre.data_ptr(), wd[1].data_ptr(), wd[3].data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                               out[2].data_ptr(), sync.data_ptr(), xchg.data_ptr(), la)
        torch.cuda.synchronize()
        return rc, out

    def fp64(T_rows):
        """h, (i, f, g, o), c of both directions; row n is live for t < T_rows[n] (dead (t, n): zeros, the state restarts from zero)"""
        xx, ww = x.double(), [t.double() for t in w]
        hout, gact, cst = torch.zeros(2, T, N, H, dtype=torch.float64), torch.zeros(2, T, N, H, 4, dtype=torch.float64), torch.zeros(2, T, N, H, dtype=torch.float64)
        for d_ in range(2):
            h, c = torch.zeros(N, H, dtype=torch.float64), torch.zeros(N, H, dtype=torch.float64)
            for s_ in range(T):
                t = s_ if d_ == 0 else T - 1 - s_
                a = (xx[t] @ ww[2 * d_].t() + h @ ww[2 * d_ + 1].t()).view(N, 4, H)
                i_, f_, g_, o_ = torch.sigmoid(a[:, 0]), torch.sigmoid(a[:, 1]), torch.tanh(a[:, 2]), torch.sigmoid(a[:, 3])
                c = f_ * c + i_ * g_
                h = o_ * torch.tanh(c)
                live = (t < T_rows).double().unsqueeze(1)
                h, c = h * live, c * live
                hout[d_, t], cst[d_, t] = h, c
                gact[d_, t] = torch.stack((i_, f_, g_, o_), -1) * live.unsqueeze(-1)
        return hout, gact.view(2, T, N, 4 * H), cst
    rc, got = fused(None)
    if rc == 3:
        # not covered on this chip for the shape: nothing was launched - the two-launch form is what the library runs then
        pytest.skip("shape not covered by the fused launch on this CU count")
    assert rc == 0, L.aas_last_error()
    assert not ops.rnn_timeout_flag()
    ref = fp64(torch.full((N,), T))
    rc, two = two_launches(None)
    assert rc == 0
    for a, b, r, nm in zip(got, two, ref, ("h", "gates", "c")):
        assert rel_err(a, r) < 2e-6, nm
        assert rel_err(a, b) < 4e-6, nm
    # two row classes: the first n1 rows run T steps, the others T2 (aasLaunch.cls_*), consumed by the fused launch like by the plain one
    if T >= 3 and N >= 2:
        n1, T2 = max(1, N // 3), max(1, T - 2 - T // 4)
        la = ops._new_claunch()
        la.cls_n_first, la.cls_T_first, la.cls_T_rest, la.rnn_tag = n1, T, T2, 9
        rc, gotc = fused(ctypes.byref(la))
        assert rc == 0 and la.cls_n_first == -1
        rows = torch.tensor([T if n < n1 else T2 for n in range(N)])
        refc = fp64(rows)
        for a, r, nm in zip(gotc, refc, ("h", "gates", "c")):
            assert rel_err(a, r) < 2e-6, nm
        rc, after = fused(None)               # nothing is left over for the next launch
        assert rc == 0 and all(torch.equal(a, b) for a, b in zip(after, got))


def test_fused_lstm_forward_refuses_what_it_does_not_cover_and_consumes_nothing(gpu):
    """rc = 3 (the caller takes the GEMM + launch form) for: a CU budget on which the batch needs more than 8 rows per workgroup, the
    split-bf16 mode, I not a multiple of 4, H > 512; the row classes offered with the refused call are still there for the call that
    follows (nothing consumed), and ops.birnn_layer gives the same layer either way."""
    import ctypes
    from aas_enhancement_amd import _lib, ops
    L = _lib.lib()
    dev = torch.device("cuda:0")
    T, N, H = 10, 30, 500
    x = torch.randn(T, N, H, device=dev) * 0.5
    w = [torch.randn(4 * H, H, device=dev) / H ** 0.5 for _ in range(4)]
    sync = torch.zeros(int(L.aas_rnn_sync_bytes()), dtype=torch.uint8, device=dev)
    xchg = torch.empty(int(L.aas_rnn_xchg_bytes(T, N, 1000, 4)), dtype=torch.uint8, device=dev)
    out = (torch.empty(2, T, N, 1000, device=dev), torch.empty(2, T, N, 4000, device=dev), torch.empty(2, T, N, 1000, device=dev))
    st = torch.cuda.current_stream().cuda_stream

    def call(H_, I_, la=None):
        return L.aas_lstm_fwd_x_ex(st, T, N, H_, I_, x.data_ptr(), w[0].data_ptr(), w[2].data_ptr(), w[1].data_ptr(), w[3].data_ptr(),
                                   out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), sync.data_ptr(), xchg.data_ptr(), la)
    assert call(H, H) == 0
    la = ops._new_claunch()
    la.rnn_cu_limit = 128                     # 30 rows on 128 CUs: 32 slices x 2 directions leave two row groups of 15
    la.cls_n_first, la.cls_T_first, la.cls_T_rest = 10, T, 4
    assert call(H, H, ctypes.byref(la)) == 3
    assert la.cls_n_first == 10                # not consumed
    assert call(H, 498) == 3 and call(H, 6) == 3
    assert call(1000, 500) == 3
    with ops.precision(1):
        assert call(H, H) == 3
    torch.cuda.synchronize()
    # the layer op: same result with the fused launch and with the ablation switch that takes the GEMM + launch form
    ys = []
    for flag in (0, 1073741824):
        L.aas_set_debug_flags(flag)
        try:
            ys.append(ops.birnn_layer(x, *[t.clone().requires_grad_(True) for t in (w[0], w[1], w[2], w[3])], kind="lstm", residual=True))
            torch.cuda.synchronize()
        finally:
            L.aas_set_debug_flags(0)
    assert rel_err(ys[0], ys[1]) < 4e-6 and not torch.equal(ys[0], ys[1])


def test_config2_enhancement_network_runs_on_the_fused_forward_launch(gpu):
    """A silent fall-back (aas_lstm_fwd_x_ex returning 3 for the headline's own shape) would cost 0.5 ms per step and nothing would
    fail: E's four layers at config-2 size (N = 30, H = 500, whole chip) must each be ONE recurrent launch with no projection GEMM in
    front, D's N = 60 pass on a 128-CU budget must keep the GEMM + launch form."""
    from aas_enhancement_amd import ops
    from aas_enhancement_amd.model import stackedBRNN
    torch.manual_seed(0)
    E = stackedBRNN(I=80, H=500, L=4).cuda()
    x = torch.randn(30, 80, 200, device="cuda")
    with torch.no_grad():
        E(x)
        torch.cuda.synchronize()
        ops.Profiler.start(("rnn", "gemm"))
        E(x)
        torch.cuda.synchronize()
        prof = ops.Profiler.stop()
    assert prof["lstm_fwdx[N=30,H=500]"]["count"] == 4 and "lstm_fwd[N=30,H=500]" not in prof
    assert prof["gemm_nt"]["count"] == 2                      # first_linear and final_linear only
    with torch.no_grad():
        st = ops.LaunchState()
        st.rnn_cu_limit = 128
        with ops.launch_state(st):
            ops.Profiler.start(("rnn", "gemm"))
            E(torch.randn(60, 80, 200, device="cuda"))
            torch.cuda.synchronize()
            prof = ops.Profiler.stop()
    assert prof["lstm_fwd[N=60,H=500]"]["count"] == 4 and not any(k.startswith("lstm_fwdx") for k in prof)
