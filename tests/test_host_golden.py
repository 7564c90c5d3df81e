"""Host-side contract (SURVEY 8a rows _collate_fn / parse_transcript, 8f rank 1 decoder) against vectors produced by the
reference's OWN loader_functions.py / AM_training/decoder.py (tools/make_goldens.py::f8_host_side)."""
import os

import numpy as np
import torch

from tests.helpers import LABELS, load


def _s(z, k):
    return bytes(z[k]).decode("utf8")


def _samples(z, paired):
    out = []
    for i in range(5):
        f, lab = torch.from_numpy(z["collate.feat%d" % i]), z["collate.label%d" % i].tolist()
        out.append((f, lab, torch.from_numpy(z["collate.paired%d" % i])) if paired else (f, lab))
    return out


def test_collate_matches_reference():
    from aas_enhancement_amd.loader_functions import _collate_fn
    z = load("f8_host_side.npz")
    got = _collate_fn(_samples(z, False))
    for k, v in zip(("inputs", "targets", "pct", "target_sizes", "mask"), got):
        ref = z["collate.out." + k]
        assert v.numpy().dtype == ref.dtype and v.shape == ref.shape, k
        assert np.array_equal(v.numpy(), ref), k
    assert got[4].n_valid == int(got[4].numel() - got[4].sum())


def test_collate_paired_matches_reference():
    from aas_enhancement_amd.loader_functions import _collate_fn_paired
    z = load("f8_host_side.npz")
    got = _collate_fn_paired(_samples(z, True))
    for k, v in zip(("inputs", "outputs", "mask", "targets", "pct", "target_sizes"), got):
        ref = z["collate_paired.out." + k]
        assert v.numpy().dtype == ref.dtype and v.shape == ref.shape, k
        assert np.array_equal(v.numpy(), ref), k


def test_parse_transcript_matches_reference(tmp_path):
    from aas_enhancement_amd.loader_functions import FeatDataset
    z = load("f8_host_side.npz")
    n = int(z["transcript.n"])
    rows = []
    for i in range(n):
        p = tmp_path / ("t%d.txt" % i)
        p.write_text(_s(z, "transcript.text%d" % i), encoding="utf8")
        rows.append("x.pt7,%s" % p)
    man = tmp_path / "m.csv"
    man.write_text("\n".join(rows) + "\n")
    ds = FeatDataset(manifest=str(man), labels=LABELS)
    for i in range(n):
        assert ds.parse_transcript(rows[i].split(",")[1]) == z["transcript.ids%d" % i].tolist(), i


def test_greedy_strings_and_error_counts():
    from aas_enhancement_amd.decoder import GreedyDecoder
    z = load("f8_host_side.npz")
    dec = GreedyDecoder(LABELS)
    paths, sizes = torch.from_numpy(z["decode.paths"]), z["decode.sizes"].tolist()
    strings = dec.convert_to_strings(paths, sizes, remove_repetitions=True)
    for i in range(paths.size(0)):
        assert strings[i][0] == _s(z, "decode.str%d" % i), i
        tgt = dec.convert_to_strings([z["decode.target%d" % i].tolist()])
        assert tgt[0][0] == _s(z, "decode.target_str%d" % i)
    for i in range(int(z["decode.npairs"])):
        a, b = _s(z, "decode.pair%d.a" % i), _s(z, "decode.pair%d.b" % i)
        assert dec.wer(a, b) == int(z["decode.pair%d.wer" % i]), (i, a, b)
        assert dec.cer(a, b) == int(z["decode.pair%d.cer" % i]), (i, a, b)
    # decode() = argmax over classes of [T,N,C] scores, then the same collapse
    T, N = paths.size(1), paths.size(0)
    probs = torch.zeros(T, N, len(LABELS))
    probs.scatter_(2, paths.t().unsqueeze(2), 1.0)
    got, _ = dec.decode(probs, sizes)
    assert [g[0] for g in got] == [s[0] for s in strings]
