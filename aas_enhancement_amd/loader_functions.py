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


def _pad_batch(feats):
    """Zero-pad a list of [F, T_i] features (already in batch order) to [N, F, T_max].
    -> (padded, lengths[N] int64, mask[N,1,T_max] u8 with 1 = padding); mask.n_valid = number of real frames."""
    lengths = torch.tensor([f.size(1) for f in feats], dtype=torch.int64)
    t_max = int(lengths.max())
    padded = torch.zeros(len(feats), feats[0].size(0), t_max)
    for row, f in zip(padded, feats):
        row[:, :f.size(1)] = f
    mask = (torch.arange(t_max)[None, :] >= lengths[:, None]).to(torch.uint8).unsqueeze(1)
    mask.n_valid = int(lengths.sum())
    return padded, lengths, mask


def _order(batch):
    """Stable sort by length, longest first (the order `sorted(..., reverse=True)` gives, loader_functions.py:51)."""
    return sorted(batch, key=lambda sample: sample[0].size(1), reverse=True)


def _flat_targets(transcripts):
    sizes = torch.tensor([len(x) for x in transcripts], dtype=torch.int32)
    flat = torch.tensor([c for x in transcripts for c in x], dtype=torch.int32)
    return flat, sizes


def _collate_fn(batch):
    """-> (inputs[N,F,T], targets[sum L] i32, input_percentages[N] f32, target_sizes[N] i32, mask[N,1,T] u8)
    (loader_functions.py:47-73)"""
    batch = _order(batch)
    inputs, lengths, mask = _pad_batch([s[0] for s in batch])
    targets, target_sizes = _flat_targets([s[1] for s in batch])
    return inputs, targets, (lengths.double() / float(inputs.size(2))).float(), target_sizes, mask


def _collate_fn_paired(batch):
    """-> (inputs, outputs(clean), mask, targets, input_percentages, target_sizes)  (loader_functions.py:75-105)"""
    batch = _order(batch)
    inputs, lengths, mask = _pad_batch([s[0] for s in batch])
    outputs = torch.zeros_like(inputs)
    for row, s in zip(outputs, batch):
        row[:, :s[0].size(1)] = s[2][:, :s[0].size(1)]
    targets, target_sizes = _flat_targets([s[1] for s in batch])
    return inputs, outputs, mask, targets, (lengths.double() / float(inputs.size(2))).float(), target_sizes


def _pad_waves(waves):
    lengths = torch.tensor([w.numel() for w in waves], dtype=torch.int64)
    out = torch.zeros(len(waves), int(lengths.max()))
    for row, w in zip(out, waves):
        row[:w.numel()] = w.reshape(-1)
    return out, lengths


def _collate_wave(batch):
    """`--preprocess code`: samples are (waveform[S], transcript).  -> (waves[N,S_max] zero padded, lengths[N] samples,
    targets[sum L] i32, target_sizes[N] i32), longest first; data_loader turns it into the `_collate_fn` tuple on the device."""
    batch = sorted(batch, key=lambda s: s[0].numel(), reverse=True)
    waves, lengths = _pad_waves([s[0] for s in batch])
    targets, target_sizes = _flat_targets([s[1] for s in batch])
    return waves, lengths, targets, target_sizes


def _collate_wave_paired(batch):
    """paired variant: samples are (noisy waveform, transcript, clean waveform of the same length)."""
    batch = sorted(batch, key=lambda s: s[0].numel(), reverse=True)
    waves, lengths = _pad_waves([s[0] for s in batch])
    cleans, _ = _pad_waves([s[2] for s in batch])
    if cleans.size(1) != waves.size(1):
        cleans = torch.nn.functional.pad(cleans, (0, waves.size(1) - cleans.size(1)))[:, :waves.size(1)]
    targets, target_sizes = _flat_targets([s[1] for s in batch])
    return waves, lengths, targets, target_sizes, cleans


class FeatLoader(DataLoader):
    def __init__(self, *args, **kwargs):
        kwargs.setdefault("collate_fn", _collate_fn)
        super().__init__(*args, **kwargs)


class FeatLoader_paired(DataLoader):
    def __init__(self, *args, **kwargs):
        kwargs.setdefault("collate_fn", _collate_fn_paired)
        super().__init__(*args, **kwargs)


class FeatSampler(Sampler):
    """Batches of consecutive (length-sorted) manifest entries; batch order shuffled per epoch (loader_functions.py:118-137).

    Data parallel (`world` > 1; new in this build): the sampler still walks the GLOBAL bins, in the same order and with the
    same random draws on every rank (same numpy seed), but hands this rank only ITS utterances of each bin - so a rank opens,
    collates, copies and (`--preprocess code`) LMFB-extracts its own shard and nothing else.  The partition: the bin's ids in
    descending manifest order (manifests are sorted by length, so this is the longest-first order `_collate_fn` produces),
    rank r takes positions r, r + W, ...: lengths are balanced across ranks and the union over ranks, re-interleaved, is the
    single-process batch.  Bins with fewer utterances than ranks (the tail bin) are dropped, identically on every rank: every
    rank needs at least one row, and all ranks must issue the same collectives."""

    def __init__(self, data_source, batch_size=1, rank=0, world=1):
        self.data_source = data_source
        self.rank, self.world = int(rank), int(world)
        ids = list(range(0, len(data_source)))
        self.bins = [ids[i:i + batch_size] for i in range(0, len(ids), batch_size)]
        self.log = None       # tests: a list that receives (global ids of the bin, this rank's ids) per batch
        dropped = [b for b in self.bins if not self._usable(b)]
        uneven = sum(1 for b in self.bins if self._usable(b) and len(b) % self.world)
        if self.world > 1 and self.rank == 0 and (dropped or uneven):
            # said once, at construction: what data parallelism changes about the epoch
            print("FeatSampler: %d ranks - %d bin(s) with fewer utterances than ranks (%d utterances) are skipped every epoch; %d bin(s) "
                  "do not divide evenly, so ranks get shards of different sizes there (per-rank BatchNorm statistics then differ: "
                  "--sync_bn makes them global)" % (self.world, len(dropped), sum(len(b) for b in dropped), uneven))

    def shard(self, ids):
        return sorted(ids, reverse=True)[self.rank::self.world]

    def _usable(self, ids):
        return self.world == 1 or len(ids) >= self.world

    def __iter__(self):
        for ids in self.bins:
            np.random.shuffle(ids)       # (every rank draws the same permutation, whether or not it keeps the bin)
            if not self._usable(ids):
                continue
            mine = ids if self.world == 1 else self.shard(ids)
            if self.log is not None:
                self.log.append((list(ids), list(mine)))
            yield mine

    def __len__(self):
        return sum(1 for ids in self.bins if self._usable(ids))

    def shuffle(self):
        np.random.shuffle(self.bins)
