"""Batch format of the hot path: the reference's collate tuple layouts
(Speech_enhancement_by_AAS/loader_functions.py:11-44 FeatDataset, :47-73 _collate_fn,
:75-105 _collate_fn_paired, :118-137 FeatSampler)."""
import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.sampler import Sampler


class FeatDataset(Dataset):
    def __init__(self, manifest, labels):
        with open(manifest) as f:
            self.ids = [x.strip().split(",") for x in f.readlines()]
        self.size = len(self.ids)
        self.labels_map = dict([(labels[i], i) for i in range(len(labels))])

    def __getitem__(self, index):
        sample = self.ids[index]
        feat = torch.load(sample[0])
        transcript = self.parse_transcript(sample[1])
        if len(sample) == 2:
            return feat, transcript
        return feat, transcript, torch.load(sample[2])

    def parse_transcript(self, transcript_path):
        with open(transcript_path, "r", encoding="utf8") as f:
            transcript = f.read().replace("\n", "")
        # unknown characters AND index 0 are dropped (filter(None, ...), loader_functions.py:40)
        return list(filter(None, [self.labels_map.get(x) for x in list(transcript)]))

    def __len__(self):
        return self.size


def _sorted(batch):
    return sorted(batch, key=lambda sample: sample[0].size(1), reverse=True)


def _collate_fn(batch):
    """-> (inputs[N,F,T], targets[sum L] i32, input_percentages[N] f32, target_sizes[N] i32, mask[N,1,T] u8)"""
    batch = _sorted(batch)
    freq_size, max_seqlength, n = batch[0][0].size(0), batch[0][0].size(1), len(batch)
    inputs = torch.zeros(n, freq_size, max_seqlength)
    input_percentages = torch.FloatTensor(n)
    target_sizes = torch.IntTensor(n)
    targets = []
    mask = torch.zeros(n, 1, max_seqlength, dtype=torch.uint8)
    n_valid = 0
    for x in range(n):
        tensor, target = batch[x][0], batch[x][1]
        seq_length = tensor.size(1)
        inputs[x].narrow(1, 0, seq_length).copy_(tensor)
        input_percentages[x] = seq_length / float(max_seqlength)
        target_sizes[x] = len(target)
        targets.extend(target)
        mask[x, :, seq_length:] = 1
        n_valid += seq_length
    mask.n_valid = n_valid
    return inputs, torch.IntTensor(targets), input_percentages, target_sizes, mask


def _collate_fn_paired(batch):
    """-> (inputs, outputs(clean), mask, targets, input_percentages, target_sizes)"""
    batch = _sorted(batch)
    freq_size, max_seqlength, n = batch[0][0].size(0), batch[0][0].size(1), len(batch)
    inputs = torch.zeros(n, freq_size, max_seqlength)
    outputs = torch.zeros(n, freq_size, max_seqlength)
    mask = torch.zeros(n, 1, max_seqlength, dtype=torch.uint8)
    input_percentages = torch.FloatTensor(n)
    target_sizes = torch.IntTensor(n)
    targets = []
    n_valid = 0
    for x in range(n):
        tensor, txt, target = batch[x][0], batch[x][1], batch[x][2]
        seq_length = tensor.size(1)
        inputs[x].narrow(1, 0, seq_length).copy_(tensor)
        outputs[x].narrow(1, 0, seq_length).copy_(target)
        mask[x, :, seq_length:] = 1
        input_percentages[x] = seq_length / float(max_seqlength)
        target_sizes[x] = len(txt)
        targets.extend(txt)
        n_valid += seq_length
    mask.n_valid = n_valid
    return inputs, outputs, mask, torch.IntTensor(targets), input_percentages, target_sizes


class FeatLoader(DataLoader):
    def __init__(self, *args, **kwargs):
        kwargs.setdefault("collate_fn", _collate_fn)
        super().__init__(*args, **kwargs)


class FeatLoader_paired(DataLoader):
    def __init__(self, *args, **kwargs):
        kwargs.setdefault("collate_fn", _collate_fn_paired)
        super().__init__(*args, **kwargs)


class FeatSampler(Sampler):
    """Batches of consecutive (length-sorted) manifest entries; batch order shuffled per epoch."""

    def __init__(self, data_source, batch_size=1):
        self.data_source = data_source
        ids = list(range(0, len(data_source)))
        self.bins = [ids[i:i + batch_size] for i in range(0, len(ids), batch_size)]

    def __iter__(self):
        for ids in self.bins:
            np.random.shuffle(ids)
            yield ids

    def __len__(self):
        return len(self.bins)

    def shuffle(self):
        np.random.shuffle(self.bins)
