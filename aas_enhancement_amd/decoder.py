"""Greedy CTC decoding + WER/CER with the reference's API (AM_training/decoder.py:45-74,146-201).
Validation-only.  Pure Python: python-Levenshtein (a C extension the reference
uses) is not a dependency here."""
import torch


def _levenshtein(a, b):
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


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
        """probs [T,N,C] -> argmax path, repeats collapsed, blanks dropped."""
        _, max_probs = torch.max(probs.transpose(0, 1), 2)
        max_probs = max_probs.cpu()
        return self.convert_to_strings(max_probs.view(max_probs.size(0), max_probs.size(1)), sizes,
                                       remove_repetitions=True, return_offsets=True)
