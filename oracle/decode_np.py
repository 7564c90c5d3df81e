"""Greedy CTC decoding and WER / CER of the trainers' validation passes (test oracle only).

Restated from the reference's AM_training/decoder.py: `GreedyDecoder.decode` (:186-201: argmax over the classes of every frame
up to the utterance's size, repeated symbols collapsed, blanks dropped - `process_string` :164-184 with remove_repetitions=True),
`convert_to_strings` of a label sequence (:149-162, no collapse), `wer` (:45-63: Levenshtein distance over words) and `cer`
(:65-74: Levenshtein distance over the characters with the spaces removed).  Pinned by tests/golden/f8_host_side.npz (strings and
distances produced by the reference's own decoder.py) and f13_validation.npz.  Pure Python / numpy."""
import numpy as np


def edit_distance(a, b):
    """Levenshtein distance of two sequences (unit costs)."""
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def labels_to_string(seq, labels, blank=0):
    """decoder.py:149-184 without collapse: drop blanks, map the rest."""
    return "".join(labels[int(i)] for i in seq if int(i) != blank)


def greedy_strings(prob_tnc, sizes, labels, blank=0):
    """decoder.py:186-201: per utterance n the argmax path of its first sizes[n] frames, repeats collapsed, blanks dropped."""
    path = np.asarray(prob_tnc).argmax(axis=2)          # [T, N]
    out = []
    for n in range(path.shape[1]):
        s, prev = "", None
        for i in range(int(sizes[n])):
            k = int(path[i, n])
            if k != blank and not (i != 0 and k == prev):
                s += labels[k]
            prev = k
        out.append(s)
    return out


def wer(s1, s2):
    return edit_distance(s1.split(), s2.split())


def cer(s1, s2):
    return edit_distance(s1.replace(" ", ""), s2.replace(" ", ""))
