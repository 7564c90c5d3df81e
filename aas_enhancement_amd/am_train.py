"""Acoustic-model (A) CTC pre-training on the HIP path - the reference's AM_training/train.py (BASELINE config 5):
the step (:297-349: A(x) -> CTC / N -> plain Adam, inf-loss guard :322-328), the epoch loop with per-epoch greedy-decode
validation (:293-459), the per-epoch / best-WER checkpoint packages (:461-478, DeepSpeech.serialize format, `--continue_from`
:163-185) and the logits dump of AM_training/test.py:138-202 (`--decoder none`).

Data parallel (new in this build): every rank trains its strided shard of the global batch; the loss is normalised by the
GLOBAL batch size and A's flat gradient buffer is SUM-all-reduced over RCCL bucket by bucket as the layers' weight gradients
finish (dist.BucketReducer).  Same batch tuples as the AAS trainer (`_collate_fn` order; the mask entry is ignored).

    python -m aas_enhancement_amd.am_train --train_manifest tr.csv --val_manifest val.csv --labels_path labels.json \
        --nFreq 80 --rnn_size 1000 --rnn_layers 5 --conv_map 128 --batch_size 30 --lr 1e-4 --epochs 10 --gpu 0
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

from . import knobs, ops
from .ctc import CTCLoss
from .decoder import GreedyDecoder
from .dist import BucketReducer, DeviceCounts, DeviceScales, DPContext, FlatBuffers
from .model import DeepSpeech, supported_rnns
from .optim import FlatAdam, FlatSGD
from .utils import AverageMeter, _get_variable_nograd


# CUs the persistent recurrent launches of the step may occupy (0 = the whole device).  Backward default: half of the device in the
# fp32-class modes - with 15 rows per workgroup instead of 8 the MFMA-bound BPTT launch issues the same 16-row tiles, and the
# weight-gradient GEMMs get the other half (config 5: 14.4 -> 13.5 ms); the latency-bound split-bf16 mode keeps the whole device
# (7.2 vs 8.1 ms).
_fwd_cus = lambda: int(knobs.get("AM_FWD_CUS"))


def _bwd_cus():
    if knobs.get("AM_BWD_CUS") is not None:
        return int(knobs.get("AM_BWD_CUS"))
    return 0 if ops._precision[0] == 1 else ops.device_cus() // 2


class AMTrainer(ops.TrainerContext):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), dp=None, labels=None, sync_bn=False, optim="adam", momentum=0.9):
        self._init_context()   # arithmetic mode + launch settings this trainer runs in (ops.TrainerContext)
        self.model = model
        self.criterion = CTCLoss()
        ops.name_layers(model, "A")
        self.flat = FlatBuffers(model)
        if optim == "adam":      # train.py:171-172,246-247
            self.opt = FlatAdam(self.flat, lr=lr, betas=betas, amsgrad=False)
        elif optim == "sgd":     # train.py:173-174,248-249: SGD(momentum, nesterov=True)
            self.opt = FlatSGD(self.flat, lr=lr, momentum=momentum)
        else:
            raise ValueError("optim must be 'adam' or 'sgd', got %r" % (optim,))
        self.dp = dp or DPContext.from_env()
        self._reducer = BucketReducer(self.dp, [self.flat]) if self.dp.active else None
        # data parallel: BatchNorm statistics over the GLOBAL batch (default: local-batch statistics per rank)
        self.launch.sync_bn = self.dp if (self.dp.active and sync_bn) else None
        self.decoder = GreedyDecoder(labels if labels is not None else DeepSpeech.get_labels(model))
        self.losses = AverageMeter()

    # ---- one step (:297-349) -----------------------------------------------------------------------------------
    @ops.with_trainer_precision
    def train_step(self, data_list):
        inputs, targets, input_percentages, target_sizes = data_list[0], data_list[1], data_list[2], data_list[3]
        inputs = _get_variable_nograd(inputs)
        N = inputs.size(0)
        t_out = self.model.output_length(inputs.size(2))
        sizes = input_percentages.clone().mul_(int(t_out)).int()
        meta = self.criterion.prepare(targets, sizes, target_sizes, inputs.device)
        if self.dp.active:   # global batch size as a device scalar (no host sync before the step is queued)
            if getattr(self, "_aux", None) is None:
                self._aux = ops.refresh_stream(inputs.device)   # the one utility stream (few hardware queues)
            counts = DeviceCounts(self.dp, [N], inputs.device, self._aux)
        ops.sync_wgrad()
        self.flat.zero_grad()
        if self._reducer is not None:
            self._reducer.begin()
            self.launch.wgrad_hook = self._reducer.on_wgrad
        try:
            out = self.model(inputs).transpose(0, 1)
            loss = self.criterion(out, targets, sizes, target_sizes, prepared=meta)
            loss = loss * (1.0 / counts.get(0)).float() if self.dp.active else loss / N
            ops.set_rnn_cu_limit(_bwd_cus())
            loss.backward()
            ops.sync_wgrad()
            if self._reducer is not None:
                self._reducer.flush(self.flat)
                self._reducer.wait()
        finally:
            self.launch.wgrad_hook = None
            ops.set_rnn_cu_limit(0)
        v = self.dp.reduce_scalars(loss.detach().reshape(1).clone())
        loss_value = float(v)                                  # host read-back: a synchronisation point
        is_inf = loss_value in (float("inf"), float("-inf"))
        if not is_inf:
            ops.check_rnn_health((loss_value,))
        # inf-loss guard (:322-328): the reference only zeroes the LOGGED value and still steps on the (non-finite)
        # gradients; an infeasible utterance here has an exactly-zero CTC gradient, so the batch's finite part is applied
        self.opt.step()
        return dict(loss=0.0 if is_inf else loss_value, is_inf=is_inf, logits=out)

    # ---- the same step without a host synchronisation -----------------------------------------------------------
    @ops.with_trainer_precision
    def train_step_async(self, data_list):
        """train_step queued WITHOUT reading the loss back: the Adam bias corrections come from a device step counter
        (`FlatAdam.step_dev`) and the (all-reduced) loss goes to a pinned host buffer asynchronously.  `read_loss(handle)`
        waits for that copy only, so a loop that reads the loss of step i-1 after queueing step i (what `fit` does) keeps the
        host one step ahead of the device: the step no longer pays the host's queueing time, and its duration does not depend
        on how busy the host's cores are (7.0-9.1 ms with the per-step read-back on a shared box).  The reference's inf guard
        (:322-328) only concerns the LOGGED value, so nothing in the update depends on the host seeing the loss first."""
        inputs, targets, input_percentages, target_sizes = data_list[0], data_list[1], data_list[2], data_list[3]
        inputs = _get_variable_nograd(inputs)
        N = inputs.size(0)
        t_out = self.model.output_length(inputs.size(2))
        sizes = input_percentages.clone().mul_(int(t_out)).int()
        return self._device_step(inputs, targets, sizes, target_sizes, N)

    def _device_step(self, inputs, targets, sizes, target_sizes, N):
        """The step as library launches only: the CTC metadata goes up from a pinned staging ring, ONE prologue launch zeroes the
        flat gradient buffer, the loss weight 1 / N (:319-320) rides in the CTC kernel's gradient scale (the root is the vector of
        per-utterance costs), and the logged value is their raw sum, scaled where it is read.  Data parallel: 1 / N_global is a
        device scalar formed from the all-reduced batch sizes (dist.DeviceScales; one launch applies it to the CTC gradient), the
        flat gradient buffer is all-reduced bucket by bucket behind the weight-gradient products, and the logged loss is the
        all-reduced cost sum times that scalar (one controller launch on the utility stream) - no host synchronisation."""
        dev, dp = inputs.device, self.dp
        meta = ops.ctc_prepare(targets, sizes, target_sizes, "cpu")
        meta = dict(meta, meta=self._upload_small(meta["meta"], dev))
        ops.sync_wgrad()
        if dp.active:
            if getattr(self, "_aux", None) is None:
                self._aux = ops.refresh_stream(dev)   # the one utility stream (few hardware queues)
            cnt = self._upload_small(torch.tensor([float(N)], dtype=torch.float64), dev)
            scales = DeviceScales(dp, cnt, [1.0, 1.0, 1.0], [0, 0, 0], self._aux)
            scale = scales[0]
        else:
            scale = 1.0 / N
        ops.step_prologue([self.flat.flat_g])
        if dp.active:
            self._reducer.begin()
            self.launch.wgrad_hook = self._reducer.on_wgrad
        try:
            ops.set_rnn_cu_limit(_fwd_cus())
            out = self.model(inputs).transpose(0, 1)
            costs = ops.ctc_scaled(out, self.criterion.blank, meta, scale)
            ops.set_rnn_cu_limit(_bwd_cus())
            torch.autograd.backward([costs], [ops.unit_root(costs)])
            ops.sync_wgrad()
            if dp.active:
                self._reducer.flush(self.flat)
                self._reducer.wait()
        finally:
            self.launch.wgrad_hook = None
            ops.set_rnn_cu_limit(0)
        self.opt.step_dev()
        if not dp.active:
            # the N per-utterance costs go to the host as they are (summed and scaled where the loss is read: no reduction launch)
            return dict(costs_dev=costs.detach(), loss_scale=1.0 / N, handle=self._loss_to_host(costs.detach(), 1.0 / N), logits=out)
        if getattr(self, "_dp_acc", None) is None:
            self._dp_acc = (torch.zeros(1, device=dev, dtype=torch.float64), torch.zeros(6, device=dev, dtype=torch.float64))
        main = torch.cuda.current_stream()
        self._aux.wait_stream(main)
        with torch.cuda.stream(self._aux):
            out3 = torch.empty(3, device=dev, dtype=torch.float64)
            ops.loss_pack(None, costs.detach(), out3)            # [0, 0, sum of this rank's costs]
            dp.reduce_scalars(out3)
            ops.began_step_sums(out3, out3[2:], 0.0, 0.0, 1.0, self._dp_acc[0], self._dp_acc[1], 0.0, 0.0, 0.0, d_scales3=scales.all, d_n_batch=scales.cnt)
            handle = self._loss_to_host(self._dp_acc[1][2:3], 1.0)     # slot 2 = the global loss: sum over all ranks' costs / N_global
        for t_ in (costs, scales.all, scales.cnt):
            t_.record_stream(self._aux)
        return dict(handle=handle, logits=out)

    def _loss_to_host(self, v, scale):
        """Asynchronous copy of a device loss scalar into a ring of four pinned slots -> the handle `read_loss` waits for."""
        ring = getattr(self, "_loss_ring", None)
        if ring is None:
            ring = self._loss_ring = dict(i=0, slots=[[torch.zeros(1, dtype=torch.float32).pin_memory(), None, 1.0] for _ in range(4)])
        slot = ring["slots"][ring["i"] % 4]
        ring["i"] += 1
        if slot[1] is not None:
            slot[1].synchronize()          # (the copy of four steps ago)
        if slot[0].numel() != v.numel() or slot[0].dtype != v.dtype:
            slot[0] = torch.zeros(v.numel(), dtype=v.dtype).pin_memory()
        slot[0].copy_(v, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        slot[1], slot[2] = ev, float(scale)
        return slot

    def read_loss(self, handle):
        """-> (logged loss, is_inf) of a train_step_async; waits for that step's loss copy only."""
        handle[1].synchronize()
        loss_value = float(handle[0].sum(dtype=handle[0].dtype)) * (handle[2] if len(handle) > 2 else 1.0)
        is_inf = loss_value in (float("inf"), float("-inf"))
        if not is_inf:
            ops.check_rnn_health((loss_value,))
        return (0.0 if is_inf else loss_value), is_inf

    # ---- validation (:357-399) ---------------------------------------------------------------------------------
    @torch.no_grad()
    @ops.with_trainer_precision
    def validate(self, batches, transcript_prob=0.0):
        """Greedy-decode WER / CER in percent, averaged over utterances (each utterance's edit distance divided by its own
        reference length, as the reference does); the model runs in eval mode (running-statistics BatchNorm, softmax)."""
        self.model.eval()
        total_wer = total_cer = 0.0
        n_utt = 0
        try:
            for data in batches:
                inputs, targets, input_percentages, target_sizes = data[0], data[1], data[2], data[3]
                split_targets, offset = [], 0
                for size in target_sizes.tolist():
                    split_targets.append(targets[offset:offset + int(size)])
                    offset += int(size)
                out = self.model(_get_variable_nograd(inputs)).transpose(0, 1)
                sizes = input_percentages.clone().mul_(int(out.size(0))).int()
                decoded, _ = self.decoder.decode(out, sizes)
                refs = self.decoder.convert_to_strings(split_targets)
                for x in range(len(refs)):
                    hyp, ref = decoded[x][0], refs[x][0]
                    wer_i = self.decoder.wer(hyp, ref) / float(max(len(ref.split()), 1))
                    cer_i = self.decoder.cer(hyp, ref) / float(max(len(ref), 1))
                    total_wer += wer_i
                    total_cer += cer_i
                    if random.uniform(0, 1) < transcript_prob:
                        print("reference = " + ref)
                        print("decoding = " + hyp)
                        print("wer = " + str(wer_i) + ", cer = " + str(cer_i))
                n_utt += len(refs)
        finally:
            self.model.train()
        n_utt = max(n_utt, 1)
        return 100.0 * total_wer / n_utt, 100.0 * total_cer / n_utt

    # ---- epoch loop (:293-486) ---------------------------------------------------------------------------------
    def fit(self, train_batches, val_batches, epochs, save_path=None, best_path=None, start_epoch=0, print_every=100,
            on_epoch_end=None, history=None, presharded=False, transcript_prob=0.0):
        """`train_batches(epoch)` / `val_batches()` return iterables of batch tuples.  Writes the running package to
        `save_path` after every epoch and the best-validation-WER package to `best_path`.  Returns the history dict.
        Data parallel: `presharded` says the batches are already this rank's shard (DataLoader(dp=...)); otherwise every rank
        is handed the global batch and keeps its strided share."""
        hist = history or dict(loss_results=[], wer_results=[], cer_results=[])
        best_wer = min(hist["wer_results"]) if hist["wer_results"] else None
        rank0 = self.dp.rank == 0
        for epoch in range(start_epoch, epochs):
            self.model.train()
            self.losses.reset()
            avg_loss, n_batches, end = 0.0, 0, time.time()
            prev = None
            for i, data in enumerate(train_batches(epoch)):
                if self.dp.active and not presharded:
                    data = self.dp.shard_collated(tuple(data[:4]) + ((data[4],) if len(data) > 4 else (torch.zeros(data[0].size(0), 1, data[0].size(2), dtype=torch.uint8),)))
                # the loss of step i is read after step i+1 has been queued: the host stays one step ahead of the device
                cur = (self.train_step_async(data), data[0].size(0), i)
                if prev is not None:
                    avg_loss, end = self._log_step(prev, epoch, avg_loss, end, rank0, print_every)
                    n_batches += 1
                prev = cur
            if prev is not None:
                avg_loss, end = self._log_step(prev, epoch, avg_loss, end, rank0, print_every)
                n_batches += 1
                prev = None
            avg_loss /= max(n_batches, 1)
            if rank0:
                print("Training Summary Epoch: [{0}]\tAverage Loss {loss:.3f}\t".format(epoch + 1, loss=avg_loss))
            wer, cer = self.validate(val_batches(), transcript_prob=transcript_prob) if val_batches is not None else (float("nan"), float("nan"))
            hist["loss_results"].append(avg_loss); hist["wer_results"].append(wer); hist["cer_results"].append(cer)
            if rank0:
                print("Validation Summary Epoch: [{0}]\tAverage WER {wer:.3f}\tAverage CER {cer:.3f}\t".format(epoch + 1, wer=wer, cer=cer))
                pkg = lambda: DeepSpeech.serialize(self.model, optimizer=self.opt, epoch=epoch, loss_results=list(hist["loss_results"]),
                                                   wer_results=list(hist["wer_results"]), cer_results=list(hist["cer_results"]))
                if save_path:
                    torch.save(pkg(), save_path)
                if best_path and (best_wer is None or best_wer > wer):
                    print("Found better validated model, saving to %s" % best_path)
                    torch.save(pkg(), best_path)
            if best_wer is None or best_wer > wer:
                best_wer = wer
            if on_epoch_end is not None:
                on_epoch_end(epoch, avg_loss, wer, cer)
        return hist

    def _log_step(self, rec, epoch, avg_loss, end, rank0, print_every):
        r, n, i = rec
        loss_value, is_inf = self.read_loss(r["handle"])
        if is_inf:
            print("WARNING: received an inf loss, setting loss value to 0")
        self.losses.update(loss_value, n)
        if rank0 and print_every and i % print_every == 0:
            print("Epoch: [{0}][{1}]\tTime {2:.3f}\tLoss {loss.val:.4f} ({loss.avg:.4f})".format(epoch + 1, i + 1, time.time() - end, loss=self.losses))
        return avg_loss + loss_value, time.time()

    @classmethod
    def resume(cls, path, lr=1e-4, gpu=0, dp=None, labels=None, sync_bn=False, optim="adam", momentum=0.9):
        """`--continue_from` (:163-185): model + optimiser state + history from a package; -> (trainer, start_epoch, history)."""
        package = torch.load(path, map_location=lambda storage, loc: storage)
        model = DeepSpeech.load_model_package(package, gpu=gpu)
        tr = cls(model, lr=lr, dp=dp, labels=labels, sync_bn=sync_bn, optim=optim, momentum=momentum)
        if package.get("optim_dict") is not None:
            tr.opt.load_state_dict(package["optim_dict"])
        start_epoch = int(package.get("epoch", 1)) - 1
        if package.get("iteration", None) is None:
            start_epoch += 1          # saved after the epoch finished: continue with the next one
        hist = dict(loss_results=list(package.get("loss_results", []) or []), wer_results=list(package.get("wer_results", []) or []),
                    cer_results=list(package.get("cer_results", []) or []))
        return tr, start_epoch, hist


@torch.no_grad()
def dump_logits(model, batches, output_path=None):
    """AM_training/test.py:138-202 with `--decoder none`: a list of (logits [T',N,C] numpy, sizes [N] numpy) per batch,
    saved with np.save when `output_path` is given (the input of the offline beam-search / LM tuning tools)."""
    was_training = model.training
    model.eval()
    out = []
    try:
        for data in batches:
            inputs, input_percentages = data[0], data[2]
            o = model(_get_variable_nograd(inputs)).transpose(0, 1)
            sizes = input_percentages.clone().mul_(int(o.size(0))).int()
            out.append((o.cpu().numpy(), sizes.numpy()))
    finally:
        model.train(was_training)
    if output_path:
        arr = np.empty(len(out), dtype=object)
        for i, item in enumerate(out):
            arr[i] = item
        np.save(output_path, arr, allow_pickle=True)
    return out


def weights_init(model, seed=None):
    """AM_training/utils.py:119-131 as applied by train.py:242 (`model.apply(weights_init)`): Conv weight ~ N(0, 0.1) and
    bias 0; BatchNorm weight ~ N(1, 0.01) and bias 0; everything else (GRU, fc) keeps its constructor initialisation."""
    from .model import _BNParams, _ConvK
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    for m in model.modules():
        if isinstance(m, _ConvK):
            m.weight.data.copy_(torch.randn(m.weight.shape, generator=g) * 0.1)
            m.bias.data.zero_()
        elif isinstance(m, _BNParams):
            m.weight.data.copy_(1.0 + torch.randn(m.weight.shape, generator=g) * 0.01)
            m.bias.data.zero_()


def str2bool(v):
    return str(v).lower() in ("true", "1")


def build_parser():
    """The reference's parser, flag for flag and default for default (AM_training/train.py:24-110), plus this build's extras at the
    end.  A reference command line therefore trains the model the reference would: --batch_size 20 --rnn_size 500 --rnn_layers 2
    --conv_map 256 --lr 1e-5 --epochs 300 unless told otherwise."""
    ap = argparse.ArgumentParser(description="DeepSpeech training")
    ap.add_argument("--DB_name", type=str, default="librispeech")
    ap.add_argument("--expnum", type=int, default=0)
    ap.add_argument("--train_manifest", metavar="DIR", help="path to train manifest csv", default="data/librispeech_logMel_train_manifest.csv")
    ap.add_argument("--val_manifest", metavar="DIR", help="path to validation manifest csv", default="data/librispeech_logMel_val_manifest.csv")
    ap.add_argument("--batch_size", default=20, type=int, help="Batch size for training")
    ap.add_argument("--num_workers", default=1, type=int, help="Number of workers used in data-loading")
    ap.add_argument("--labels_path", default="labels.json", help="Contains all characters for transcription")
    # used only with --preprocess code (the reference's SpectrogramDataset); this build's LMFB kernel fixes 16 kHz / 20 ms hamming / 10 ms
    ap.add_argument("--sample_rate", default=16000, type=int, help="Sample rate")
    ap.add_argument("--window_size", default=.02, type=float, help="Window size for spectrogram in seconds")
    ap.add_argument("--window_stride", default=.01, type=float, help="Window stride for spectrogram in seconds")
    ap.add_argument("--window", default="hamming", help="Window type for spectrogram generation")
    ap.add_argument("--rnn_size", default=500, type=int, help="Hidden size of RNNs")
    ap.add_argument("--rnn_layers", default=2, type=int, help="Number of RNN layers")
    ap.add_argument("--rnn_type", default="gru", help="Type of the RNN. rnn|gru|lstm are supported")
    ap.add_argument("--conv_layers", default=2, type=int)
    ap.add_argument("--cnn_residual_blocks", default=1, type=int)   # (ResidualDeepSpeech only: out of scope, accepted and unused)
    ap.add_argument("--conv_map", default=256, type=int)
    ap.add_argument("--conv_kernel", default=11, type=int)
    ap.add_argument("--conv_stride", default=2, type=int)
    ap.add_argument("--nFreq", default=40, type=int)
    ap.add_argument("--n_mels", default=40, type=int)
    ap.add_argument("--preprocess", default="file", type=str, help="file: LMFB .pt7 tensors | code: waveforms + the LMFB HIP kernel")
    ap.add_argument("--process_mel", default=False, type=str2bool)
    ap.add_argument("--normalize", default=False, type=str2bool)
    ap.add_argument("--arch_ver", default="ken", type=str, help="ken (1D CNN, lReLU): the only architecture on the hot path")
    ap.add_argument("--nDownsample", type=int, default=1)
    ap.add_argument("--print_every", type=int, default=100)
    ap.add_argument("--epochs", default=300, type=int, help="Number of training epochs")
    ap.add_argument("--gpu", default=-1, type=int)
    ap.add_argument("--lr", "--learning-rate", default=1e-5, type=float, help="initial learning rate")
    ap.add_argument("--momentum", default=0.9, type=float, help="momentum")
    ap.add_argument("--optim", default="adam", help="adam|sgd")
    ap.add_argument("--one_sample_DEBUG", default=False, type=str2bool)
    ap.add_argument("--include_first_BN", default=True, type=str2bool)
    ap.add_argument("--log_params", dest="log_params", action="store_true", help="Log parameter values and gradients")
    ap.add_argument("--save_folder", default="models/", help="Location to save epoch models")
    ap.add_argument("--model_path", default="", help="Location to save best validation model (set from --DB_name / --expnum, train.py:138)")
    ap.add_argument("--continue_from", default="", help="Continue from checkpoint model")
    ap.add_argument("--augment", type=str2bool, default=False, help="Use random tempo and gain perturbations.")
    ap.add_argument("--transcript_prob", type=float, default=0.002)
    ap.add_argument("--sortagrad", default=False, type=str2bool, help="load minibatch with order of increasing length from shorter to longer")
    # ---- this build's extras (absent from the reference)
    ap.add_argument("--precision", default=None, choices=("fp32", "fp32eq", "bf16x3"),
                    help="absent: AAS_PRECISION from the environment, else fp32 | fp32 (the reference's arithmetic) | fp32eq (fp32-equivalent six-product GEMMs) | bf16x3 (split-bf16 fast mode)")
    ap.add_argument("--seed", default=123456, type=int, help="torch.manual_seed (train.py:112 hard-codes 123456)")
    ap.add_argument("--dist_backend", default="nccl")
    ap.add_argument("--sync_bn", action="store_true", help="data parallel: all-reduce the BatchNorm statistics (global-batch BN)")
    return ap


def resolve_args(a):
    """What train.py:128-138,201-214 derives from the flags before it builds anything; and what this build cannot honour, said loudly."""
    if a.arch_ver != "ken":
        raise NotImplementedError("--arch_ver %s: only 'ken' (DeepSpeech_ken, 1-D convolutions + LeakyReLU) is on the MI355X hot path; "
                                  "ResidualDeepSpeech / ResidualCNN4block are out of scope (SURVEY.md 2)" % a.arch_ver)
    if a.process_mel:
        a.nFreq = a.n_mels                                      # train.py:131-132
    a.model_path = "models/" + a.DB_name + "_" + str(a.expnum) + "_final.pth.tar"     # train.py:138 (always overwritten)
    notes = []
    if a.preprocess == "code":
        if not a.process_mel:
            raise NotImplementedError("--preprocess code without --process_mel true asks for the 161-bin log-spectrogram + CMVN features of the "
                                      "reference's SpectrogramDataset; this build extracts log-Mel filterbank features on the GPU (LMFB kernel): "
                                      "pass --process_mel true --n_mels N")
        if (a.sample_rate, a.window_size, a.window_stride, a.window) != (16000, .02, .01, "hamming"):
            raise NotImplementedError("the LMFB kernel is built for 16 kHz / 20 ms hamming window / 10 ms hop (train.py:39-42 defaults)")
        for name in ("normalize", "augment"):
            if getattr(a, name):
                notes.append("--%s true is not implemented by the GPU feature extraction: ignored" % name)
    else:
        for name, dflt in (("sample_rate", 16000), ("window_size", .02), ("window_stride", .01), ("window", "hamming"), ("normalize", False), ("augment", False)):
            if getattr(a, name) != dflt:
                notes.append("--%s is used only with --preprocess code (train.py:38): ignored in file mode" % name)
    if a.gpu < 0:
        notes.append("--gpu -1 selects the reference's CPU path; this build has none (HIP only): running on cuda:0")
        a.gpu = 0
    for name in ("one_sample_DEBUG", "log_params"):
        if getattr(a, name):
            notes.append("--%s: accepted for command-line compatibility, no effect" % name)
    for n_ in notes:
        print("[am_train] " + n_, file=sys.stderr)
    return a


def main(argv=None):
    ap = build_parser()
    a, unparsed = ap.parse_known_args(argv)
    if len(unparsed) > 0:                                       # train.py:121-125
        print(unparsed)
        assert len(unparsed) == 0, "length of unparsed option should be 0"
    a = resolve_args(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        a.gpu = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(a.gpu)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.dist_backend, **(dict(device_id=torch.device("cuda", a.gpu)) if a.dist_backend == "nccl" else {}))
    torch.manual_seed(a.seed); np.random.seed(a.seed); random.seed(a.seed)
    torch.cuda.set_device(a.gpu)
    if a.precision is not None:
        ops.set_precision({"fp32": 0, "bf16x3": 1, "fp32eq": 2}[a.precision])
    from .data_loader import DataLoader
    with open(a.labels_path) as f:
        labels = str("".join(json.load(f)))
    dp = DPContext.from_env()
    dl = DataLoader(batch_size=a.batch_size, tr_ny_manifest=a.train_manifest, val_manifest=a.val_manifest, labels=labels,
                    num_workers=a.num_workers, pin_memory=True, preprocess=a.preprocess, n_mels=a.nFreq, dp=dp)
    os.makedirs(a.save_folder, exist_ok=True)
    if a.continue_from:
        tr, start_epoch, hist = AMTrainer.resume(a.continue_from, lr=a.lr, gpu=a.gpu, dp=dp, labels=labels, sync_bn=a.sync_bn, optim=a.optim,
                                                 momentum=a.momentum)
    else:
        rnn_type = a.rnn_type.lower()
        assert rnn_type in supported_rnns, "rnn_type should be either lstm, rnn or gru"
        audio_conf = (dict(sample_rate=a.sample_rate, window_size=a.window_size, window_stride=a.window_stride, window=a.window, n_mels=a.n_mels,
                           process_mel=a.process_mel) if a.preprocess == "code" else None)
        model = DeepSpeech(rnn_hidden_size=a.rnn_size, rnn_layers=a.rnn_layers, rnn_type=supported_rnns[rnn_type], labels=labels, audio_conf=audio_conf,
                           kernel_sz=a.conv_kernel, stride=a.conv_stride, map=a.conv_map, cnn_layers=a.conv_layers, nFreq=a.nFreq,
                           nDownsample=a.nDownsample, include_first_BN=a.include_first_BN)     # train.py:228-239
        weights_init(model)
        tr, start_epoch, hist = AMTrainer(model.cuda(), lr=a.lr, dp=dp, labels=labels, sync_bn=a.sync_bn, optim=a.optim, momentum=a.momentum), 0, None
    # batch order (train.py:270-272,484-486): sortagrad keeps the manifest (increasing length) order for the first epoch and never
    # reshuffles; otherwise the bins are shuffled before the first epoch and after every epoch.  The loader builds its iterators
    # lazily, so this shuffle happens before any batch has been drawn.
    dl.reshuffle = not a.sortagrad
    if not (a.sortagrad and start_epoch == 0):
        dl._sp["ny/train"].shuffle()
    n_train = len(dl._sp["ny/train"])
    train_batches = lambda epoch: (dl.next("ny", "train") for _ in range(n_train))
    val_batches = lambda: (dl.next("ny", "val") for _ in range(dl.num_batches("val")))
    file_path = "%s/%s_%d.pth.tar" % (a.save_folder, a.DB_name, a.expnum)      # always overwritten by the most recent epoch's model
    if dp.rank == 0:
        print("Number of parameters: %d" % DeepSpeech.get_param_size(tr.model))
        os.makedirs(os.path.dirname(a.model_path) or ".", exist_ok=True)
        torch.save(DeepSpeech.serialize(tr.model, optimizer=tr.opt, epoch=0), file_path)     # train.py:288-291 ("save model file for error check")
    tr.fit(train_batches, val_batches, a.epochs, save_path=file_path, best_path=a.model_path, start_epoch=start_epoch, print_every=a.print_every,
           history=hist, presharded=dp.active, transcript_prob=a.transcript_prob)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
