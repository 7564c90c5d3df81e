"""Greedy CTC decoding + WER/CER with the reference's API (AM_training/decoder.py:45-74,146-201).
Validation pass of the trainers.  The argmax + collapse runs on the device (aas_greedy_decode,
one wavefront per utterance) when the scores are on the GPU; the edit distance is the native host function
aas_edit_distance of libaas_hip.so (the reference uses the python-Levenshtein C extension)."""
import ctypes

import numpy as np
import torch


def _levenshtein(a, b):
    """Edit distance of two strings (native two-row dynamic programme over the code points)."""
    from ._lib import lib
    ia = np.fromiter((ord(c) for c in a), dtype=np.int32, count=len(a))
    ib = np.fromiter((ord(c) for c in b), dtype=np.int32, count=len(b))
    d = lib().aas_edit_distance(ia.ctypes.data_as(ctypes.c_void_p), len(a), ib.ctypes.data_as(ctypes.c_void_p), len(b))
    if d < 0:
        raise RuntimeError("aas_edit_distance failed")
    return int(d)


class Decoder(object):
    def __init__(self, labels, blank_index=0):
        self.labels = labels
        self.int_to_char = dict([(i, c) for (i, c) in enumerate(labels)])
        self.blank_index = blank_index
        space_index = len(labels)
        if " " in labels:
            space_index = labels.index(" ")
        self.space_index = space_index

    def wer(self, s1, s2):
        b = set(s1.split() + s2.split())
        word2char = dict(zip(b, range(len(b))))
        w1 = [chr(word2char[w]) for w in s1.split()]
        w2 = [chr(word2char[w]) for w in s2.split()]
        return _levenshtein("".join(w1), "".join(w2))

    def cer(self, s1, s2):
        s1, s2 = s1.replace(" ", ""), s2.replace(" ", "")
        return _levenshtein(s1, s2)


class GreedyDecoder(Decoder):
    def convert_to_strings(self, sequences, sizes=None, remove_repetitions=False, return_offsets=False):
        strings, offsets = [], []
        for x in range(len(sequences)):
            seq_len = sizes[x] if sizes is not None else len(sequences[x])
            string, string_offsets = self.process_string(sequences[x], seq_len, remove_repetitions)
            strings.append([string])
            offsets.append([string_offsets])
        return (strings, offsets) if return_offsets else strings

    def process_string(self, sequence, size, remove_repetitions=False):
        string, offsets = "", []
        for i in range(int(size)):
            idx = int(sequence[i])
            char = self.int_to_char[idx]
            if char != self.int_to_char[self.blank_index]:
                if remove_repetitions and i != 0 and char == self.int_to_char[int(sequence[i - 1])]:
                    pass
                elif char == self.labels[self.space_index]:
                    string += " "
                    offsets.append(i)
                else:
                    string += char
                    offsets.append(i)
        return string, torch.IntTensor(offsets)

    def decode(self, probs, sizes=None):
        """probs [T,N,C] -> (strings, offsets): argmax path, repeats collapsed, blanks dropped (decoder.py:186-201)."""
        if probs.is_cuda:
            return self._decode_device(probs, sizes)
        _, max_probs = torch.max(probs.transpose(0, 1), 2)
        return self.convert_to_strings(max_probs.view(max_probs.size(0), max_probs.size(1)), sizes,
                                       remove_repetitions=True, return_offsets=True)

    def _decode_device(self, probs, sizes):
        from ._lib import check, lib, ptr, stream
        probs = probs.detach()
        probs = probs if probs.is_contiguous() else probs.contiguous()
        T, N, C = probs.shape
        dev = probs.device
        sz = torch.full((N,), T, dtype=torch.int32) if sizes is None else torch.as_tensor(sizes).to(torch.int32).reshape(-1)
        d_sz = sz.to(dev)
        out = torch.empty((N, T), dtype=torch.int32, device=dev)
        offs = torch.empty((N, T), dtype=torch.int32, device=dev)
        lens = torch.empty((N,), dtype=torch.int32, device=dev)
        check(lib().aas_greedy_decode(stream(), ptr(probs), ptr(d_sz), T, N, C, self.blank_index, ptr(out), ptr(offs), ptr(lens)),
              "aas_greedy_decode")
        out_h, offs_h, lens_h = out.cpu(), offs.cpu(), lens.cpu().tolist()
        strings, offsets = [], []
        for n in range(N):
            k = lens_h[n]
            strings.append(["".join(self.int_to_char[i] for i in out_h[n, :k].tolist())])
            offsets.append([offs_h[n, :k].clone()])
        return strings, offsets
