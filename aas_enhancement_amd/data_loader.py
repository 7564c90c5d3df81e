"""DataLoader.next(cl_ny, type) with the reference's restart-and-shuffle semantics
(Speech_enhancement_by_AAS/data_loader.py:8-83), plus what the reference leaves to `.cuda()` inside the trainer:

* `pin_memory=True`: batches are collated into pinned host memory (torch's pin thread) and copied to the device on a
  private copy stream ONE BATCH AHEAD, so the H2D copy of batch i+1 overlaps the training step on batch i; `next()`
  hands out device tensors for the big arrays (inputs / clean targets / mask) and host tensors for the small integer
  metadata the trainer reads on the host (targets, input_percentages, target_sizes) - the `_collate_fn` tuple layouts.
* `preprocess="code"` (AM_training/train.py:59 `--preprocess file|code`): manifests list 16 kHz waveform tensors
  instead of precomputed LMFB `.pt7` files; the log-Mel features are extracted on the device by the LMFB HIP kernel
  (aas_enhancement_amd/lmfb.py), per utterance length, on the copy stream - also one batch ahead.
* `dp=DPContext` (data parallel): the TRAINING sets are sharded BEFORE loading (FeatSampler(rank, world)): a rank opens only its
  own utterances of each global bin.  Its shard is then padded to the GLOBAL maximum length (one host integer, MAX-all-reduced
  over the gloo side group - no device synchronisation) and `input_percentages` re-expressed against it, so the un-masked L1
  sums and A's output lengths equal the single-process ones.  Validation sets are not sharded (rank 0 validates).
"""
import torch

from . import loader_functions as LF
from .loader_functions import FeatDataset, FeatLoader, FeatLoader_paired, FeatSampler


class DataLoader():
    def __init__(self, batch_size, paired=False, tr_cl_manifest="", tr_ny_manifest="", trsub_manifest="",
                 val_manifest="", val2_manifest="", labels=None, num_workers=1, pin_memory=False, preprocess="file",
                 device=None, n_mels=80, dp=None):
        self.batch_size, self.labels, self.num_workers = batch_size, labels, num_workers
        self.paired, self.preprocess = paired, preprocess
        self.dp = dp if (dp is not None and dp.active) else None
        if preprocess not in ("file", "code"):
            raise ValueError("preprocess must be 'file' or 'code', got %r" % (preprocess,))
        self.pin = bool(pin_memory) and torch.cuda.is_available()
        self.device = torch.device(device) if device is not None else (torch.device("cuda", torch.cuda.current_device()) if self.pin else None)
        self.Loader = FeatLoader_paired if paired else FeatLoader
        self._collate = None
        if preprocess == "code":
            self._collate = LF._collate_wave_paired if paired else LF._collate_wave
            if self.device is None or self.device.type != "cuda":
                raise RuntimeError("preprocess='code' extracts LMFB features with the HIP kernel: it needs the GPU (no CPU fallback)")
            from .lmfb import LMFB
            self._lmfb = LMFB(n_mels=n_mels).to(self.device)
        self._ds, self._sp, self._it, self._ahead = {}, {}, {}, {}
        self._copy_stream = None
        if self.device is not None and self.device.type == "cuda":
            from . import ops
            self._copy_stream = ops.refresh_stream(self.device)   # the process's one utility stream: streams share four hardware queues
        for key, manifest, sampled in (("cl/train", tr_cl_manifest, True), ("ny/train", tr_ny_manifest, True),
                                       ("ny/trsub", trsub_manifest, False), ("ny/val", val_manifest, False),
                                       ("ny/val2", val2_manifest, False)):
            if len(manifest) > 0:
                self._ds[key] = FeatDataset(manifest=manifest, labels=labels)
                if sampled:
                    self._sp[key] = FeatSampler(self._ds[key], batch_size=batch_size, **(dict(rank=self.dp.rank, world=self.dp.world) if self.dp else {}))
        # iterators are created on first use: a caller may still reorder a sampler's bins (am_train: --sortagrad / shuffle before
        # the first epoch) - a torch DataLoader iterator with workers prefetches its first bins the moment it exists
        self.reshuffle = True     # reshuffle the training bins whenever a training set wraps around (data_loader.py:42-83)

    def _make(self, key):
        kw = dict(num_workers=self.num_workers, pin_memory=self.pin)
        if self._collate is not None:
            kw["collate_fn"] = self._collate
        if key in self._sp:
            return iter(self.Loader(self._ds[key], batch_sampler=self._sp[key], **kw))
        return iter(self.Loader(self._ds[key], batch_size=self.batch_size, **kw))

    def num_batches(self, type):
        ds = self._ds["ny/" + type]
        return (len(ds) + self.batch_size - 1) // self.batch_size

    # ---- host side: the reference's semantics (data_loader.py:42-83) ---------------------------------------------
    def _next_host(self, key):
        if key not in self._it:
            if key not in self._ds:
                raise KeyError("no manifest was given for %r" % (key,))
            self._it[key] = self._make(key)
        try:
            batch = next(self._it[key])
        except StopIteration:
            if key in self._sp and self.reshuffle:
                self._sp[key].shuffle()       # training sets: reshuffle the batch order, then restart
            self._it[key] = self._make(key)
            batch = next(self._it[key])
        if self.dp is not None and key in self._sp:
            batch = self._pad_to_global(batch)
        return batch

    def _pad_to_global(self, batch):
        """This rank's shard -> padded to the global batch's longest utterance (the padding the single-process collate would
        have produced), `input_percentages` = T_i / T_global.  One host-side MAX all-reduce of one integer per batch."""
        if self.preprocess == "code":
            waves = batch[0]
            s_glob = self.dp.host_max(waves.size(1))
            if s_glob > waves.size(1):
                pad = lambda w: torch.nn.functional.pad(w, (0, s_glob - w.size(1)))
                batch = (pad(waves),) + tuple(batch[1:4]) + ((pad(batch[4]),) if self.paired else ())
            return batch
        big = [i for i, t in enumerate(batch) if torch.is_tensor(t) and t.dim() == 3]       # inputs [, outputs], mask
        pct_i = 4 if self.paired else 2
        t_loc = batch[big[0]].size(2)
        t_glob = self.dp.host_max(t_loc)
        if t_glob == t_loc:
            return batch
        out = list(batch)
        lengths = torch.round(batch[pct_i].double() * t_loc)
        for i in big:
            t = batch[i]
            out[i] = torch.nn.functional.pad(t, (0, t_glob - t_loc), value=1 if t.dtype == torch.uint8 else 0)   # mask: 1 = padding
        out[pct_i] = (lengths / float(t_glob)).float()
        return tuple(out)

    # ---- device side: one batch ahead on the copy stream ---------------------------------------------------------
    def _to_device(self, batch):
        """-> (tuple in the collate layout with the big tensors on the device, event recorded on the copy stream)."""
        dev, cs = self.device, self._copy_stream
        with torch.cuda.stream(cs):
            if self.preprocess == "code":
                batch = self._features_from_waves(batch)
            out = []
            for t in batch:
                if torch.is_tensor(t) and t.dim() == 3:            # inputs / outputs [N,F,T] and mask [N,1,T]
                    if t.dtype == torch.uint8 and not hasattr(t, "n_valid"):
                        t.n_valid = int(t.numel()) - int(t.sum().item()) if not t.is_cuda else None
                    d = t if t.is_cuda else t.to(dev, non_blocking=True)
                    if getattr(t, "n_valid", None) is not None:
                        d.n_valid = t.n_valid
                    out.append(d)
                else:
                    out.append(t)
            ev = torch.cuda.Event()
            ev.record(cs)
        return tuple(out), ev

    def _features_from_waves(self, batch):
        """(waves[N,S], lengths[N], targets, target_sizes[, clean waves]) -> the `_collate_fn*` tuple with LMFB features:
        T_i = 1 + S_i // hop frames per utterance, zero beyond T_i, input_percentages = T_i / T_max, mask = 1 on padding."""
        if self.paired:
            waves, lengths, targets, target_sizes, cleans = batch
        else:
            waves, lengths, targets, target_sizes = batch
        hop = self._lmfb.hop
        d_len = lengths.to(torch.int32).to(self.device, non_blocking=True)
        feats = self._lmfb(waves.to(self.device, non_blocking=True), d_len)
        frames = 1 + lengths.to(torch.int64) // hop
        t_max = feats.size(2)
        mask = (torch.arange(t_max)[None, :] >= frames[:, None]).to(torch.uint8).unsqueeze(1)
        mask.n_valid = int(frames.sum())
        pct = (frames.double() / float(t_max)).float()
        if self.paired:
            clean_feats = self._lmfb(cleans.to(self.device, non_blocking=True), d_len)
            return feats, clean_feats, mask, targets, pct, target_sizes
        return feats, targets, pct, target_sizes, mask

    def next(self, cl_ny="", type=""):
        key = "%s/%s" % (cl_ny, type)
        if self._copy_stream is None:
            return self._next_host(key)
        if key not in self._ahead:
            self._ahead[key] = self._to_device(self._next_host(key))
        batch, ev = self._ahead[key]
        torch.cuda.current_stream().wait_event(ev)
        for t in batch:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(torch.cuda.current_stream())
        self._ahead[key] = self._to_device(self._next_host(key))   # start the next batch's copy / extraction now
        return batch
