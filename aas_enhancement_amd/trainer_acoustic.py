"""Acoustic-supervision trainer on the MI355X HIP path - same API as the reference's
Speech_enhancement_by_AAS/trainer_acoustic.py (Trainer.train, hot loop :120-142): the enhancer E is trained through the
acoustic model A alone, loss = CTC(A(E(x))) / N.  There is NO discriminator in this trainer: none is built, none runs
(`main.py --trainer acoustic_supervision`; it used to be mapped onto the AAS trainer with w_adversarial = 0, which still ran
D's two forward and two backward passes and multiplied them by zero).

The step is one serial chain - E forward, A forward, CTC, A backward, E backward - so every persistent recurrent launch
takes the whole chip; the weight-gradient products run on the side stream as in the AAS trainer.  `train_step_async` keeps
the loss and the running CTC average of the log line on the device (no host synchronisation per step).
"""

import torch

from . import knobs, ops
from .trainer_AAS import Trainer as _AASTrainer


class Trainer(_AASTrainer):
    def build_model(self):
        from .model import stackedBRNN, supported_rnns
        c = self.config
        print("initialize enhancement model")
        self.G = stackedBRNN(I=c.nFeat, H=c.rnn_size, L=c.rnn_layers, rnn_type=supported_rnns[c.rnn_type])
        self.D = _NoNet()
        self.ASR = self.load_asr_package()

    def __init__(self, config, data_loader=None, models=None):
        if models is not None and len(models) == 2:   # (E, A): the shared plumbing iterates G, D, ASR - D is an empty stand-in
            models = (models[0], _NoNet(), models[1])
        super().__init__(config, data_loader, models)

    def make_optimizers(self):
        """Adam(amsgrad) on E and A (:117-118), flat buffers as in the AAS trainer."""
        from .dist import BucketReducer, DPContext, FlatBuffers
        from .optim import FlatAdam
        c = self.config
        self.dp = getattr(self, "dp", None) or DPContext.from_env()
        for name, m in (("G", self.G), ("ASR", self.ASR)):
            ops.name_layers(m, name)
        self._frozen_asr = self.asr_frozen()
        if self._frozen_asr:
            for p in self.ASR.parameters():
                p.requires_grad_(False)
        self._flat = {"G": FlatBuffers(self.G)}
        if not self._frozen_asr:
            self._flat["A"] = FlatBuffers(self.ASR)
        mk = lambda f: FlatAdam(f, lr=c.lr, betas=(self.beta1, self.beta2), amsgrad=True)
        self._opts = (mk(self._flat["G"]), mk(self._flat["A"]) if "A" in self._flat else None, None)
        self._reducer = BucketReducer(self.dp, self._flat.values()) if self.dp.active else None
        self.launch.sync_bn = self.dp if (self.dp.active and getattr(c, "sync_bn", False)) else None
        return self._opts

    # ---- one iteration of :120-142 -------------------------------------------------------------
    def _core(self, inputs, ctc_meta, scale, it, targets=None, sizes=None, target_sizes=None, n_glob=None):
        """E forward -> A forward -> CTC -> backward through A and E -> Adam.  `scale` = 1 / N (python float, or a device
        scalar when data parallel); returns (enhanced, prob, l_CTC) as device tensors."""
        c = self.config
        optimizer_g, optimizer_asr, _ = self._opts
        asr_steps = optimizer_asr is not None and it > c.allow_ASR_update_iter
        ops.sync_wgrad()
        for f in self._flat.values():
            f.flat_g.zero_()
        if self._reducer is not None:
            self._reducer.begin()
            self.launch.wgrad_hook = self._reducer.on_wgrad
        try:
            ops.set_rnn_cu_limit(0)
            enhanced = self.G(inputs)
            prob = self.ASR(enhanced).transpose(0, 1)
            if targets is None:
                l_CTC = ops.ctc_sum(prob, None, None, None, self.CTCLoss.blank, ctc_meta) * scale
            else:
                l_CTC = self.CTCLoss(prob, targets, sizes, target_sizes, prepared=ctc_meta) / n_glob
            # (AAS_AC_BWD_CUS: CU cap of the BPTT launches, leaving CUs to the weight-gradient products beside them; 0 = whole chip)
            ops.set_rnn_cu_limit(knobs.get("AC_BWD_CUS"))
            l_CTC.backward()
            ops.set_rnn_cu_limit(0)
            ops.sync_wgrad()
            if self._reducer is not None:
                for f in self._flat.values():
                    self._reducer.flush(f)
                self._reducer.wait()
        finally:
            self.launch.wgrad_hook = None
        return enhanced, prob, l_CTC, asr_steps

    @ops.with_trainer_precision
    def train_step(self, data_list, iter):
        """Synchronous form: returns the host scalars the reference logs (:139-147)."""
        if self._opts is None:
            self.make_optimizers()
        if getattr(self, "_acc_live", False):
            self.read_scalars()
        inputs, targets, input_percentages, target_sizes, _ = self._prep(data_list)
        N = inputs.size(0)
        t_out = self.ASR.output_length(inputs.size(2))
        sizes = input_percentages.clone().mul_(int(t_out)).int()
        ctc_meta = self.CTCLoss.prepare(targets, sizes, target_sizes, inputs.device)
        n_glob = self.dp.global_counts([N])[0] if self.dp.active else N
        enhanced, prob, l_CTC, asr_steps = self._core(inputs, ctc_meta, None, iter, targets, sizes, target_sizes, n_glob)
        self._opts[0].step()
        if asr_steps:
            self._opts[1].step()
            ops.refresh_weight_planes(self.ASR)
        ops.refresh_weight_planes(self.G)
        l_ctc = float(self.dp.reduce_scalars(l_CTC.detach().reshape(1).double()).item())
        ops.check_rnn_health((l_ctc,))
        self.ctc_tr_local.update(l_ctc, n_glob)
        self._last_sync_l_ctc = l_ctc
        return dict(l_ctc=l_ctc, enhanced=enhanced, prob=prob)

    @ops.with_trainer_precision
    def train_step_async(self, data_list, iter):
        """The same iteration without a host synchronisation: the loss and the running CTC average stay on the device;
        `read_scalars()` fetches them when a log line needs them."""
        if self._opts is None:
            self.make_optimizers()
        inputs, targets, input_percentages, target_sizes, _ = self._prep(data_list)
        N, dev = inputs.size(0), inputs.device
        t_out = self.ASR.output_length(inputs.size(2))
        sizes = input_percentages.clone().mul_(int(t_out)).int()
        meta = ops.ctc_prepare(targets, sizes, target_sizes, "cpu")
        meta = dict(meta, meta=self._upload_small(meta["meta"], dev))
        if getattr(self, "_acc", None) is None:
            self._acc = torch.zeros(6, device=dev, dtype=torch.float64)      # aas_began_step* layout: [-, -, last loss, -, sum(loss * N), sum(N)]
            self._no_l1 = torch.zeros(2, device=dev, dtype=torch.float64)
            self._no_kt = torch.zeros(1, device=dev, dtype=torch.float64)
        # library launches only: one prologue launch zeroes the flat gradient buffers, the loss weight 1 / N rides in the CTC kernel's
        # gradient scale, and the log accumulators are advanced from the raw per-utterance costs by the controller launch (its
        # adversarial inputs are zero here: this trainer has no discriminator).  Data parallel: 1 / N_global as a device scalar from
        # the all-reduced batch sizes, bucketed gradient all-reduce, the cost sum all-reduced before the controller.
        c, dp = self.config, self.dp
        optimizer_g, optimizer_asr, _ = self._opts
        asr_steps = optimizer_asr is not None and iter > c.allow_ASR_update_iter
        ops.sync_wgrad()
        if dp.active:
            from .dist import DeviceScales
            aux = ops.refresh_stream(dev)
            cnt = self._upload_small(torch.tensor([float(N)], dtype=torch.float64), dev)
            scales = DeviceScales(dp, cnt, [1.0, 1.0, 1.0], [0, 0, 0], aux)
            scale = scales[0]
        else:
            scale = 1.0 / N
        ops.step_prologue([f.flat_g for f in self._flat.values()])
        if dp.active:
            self._reducer.begin()
            self.launch.wgrad_hook = self._reducer.on_wgrad
        try:
            ops.set_rnn_cu_limit(0)
            enhanced = self.G(inputs)
            prob = self.ASR(enhanced).transpose(0, 1)
            costs = ops.ctc_scaled(prob, self.CTCLoss.blank, meta, scale)
            ops.set_rnn_cu_limit(knobs.get("AC_BWD_CUS"))
            torch.autograd.backward([costs], [ops.unit_root(costs)])
            ops.set_rnn_cu_limit(0)
            ops.sync_wgrad()
            if dp.active:
                for f in self._flat.values():
                    self._reducer.flush(f)
                self._reducer.wait()
        finally:
            self.launch.wgrad_hook = None
            ops.set_rnn_cu_limit(0)
        optimizer_g.step_dev()
        if asr_steps:
            optimizer_asr.step_dev()
            ops.refresh_weight_planes(self.ASR)
        ops.refresh_weight_planes(self.G)
        if not dp.active:
            ops.began_step_raw(self._no_l1, 0.0, 0.0, costs.detach(), 1.0 / N, self._no_kt, self._acc, 0.0, 0.0, float(N))
        else:
            main = torch.cuda.current_stream()
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                out3 = torch.empty(3, device=dev, dtype=torch.float64)
                ops.loss_pack(None, costs.detach(), out3)
                dp.reduce_scalars(out3)
                ops.began_step_sums(out3, out3[2:], 0.0, 0.0, 1.0, self._no_kt, self._acc, 0.0, 0.0, 0.0, d_scales3=scales.all, d_n_batch=scales.cnt)
                self._acc_ev = torch.cuda.Event()
                self._acc_ev.record(aux)
            for t_ in (costs, scales.all, scales.cnt):
                t_.record_stream(aux)
        self._acc_live = True
        return dict(enhanced=enhanced, prob=prob, scalars=self._acc)

    def read_scalars(self):
        """One D2H copy: the last queued step's loss; feeds the running CTC average of the log line; a synchronisation
        point - raises if a persistent kernel timed out or the run diverged."""
        if getattr(self, "_acc", None) is None:      # only synchronous train_step calls so far: nothing queued to read back
            return dict(l_ctc=getattr(self, "_last_sync_l_ctc", None))
        if getattr(self, "_acc_ev", None) is not None:     # (data parallel: the controller ran on the auxiliary stream)
            torch.cuda.current_stream().wait_event(self._acc_ev)
        _, _, l_ctc, _, s, n = self._acc.tolist()
        self._acc[4:6].zero_()
        self._acc_live = False
        ops.check_rnn_health((l_ctc,))
        if n > 0:
            self.ctc_tr_local.update(s / n, n)
        return dict(l_ctc=l_ctc)

    def zero_grad_all(self):
        ops.sync_wgrad()
        for f in (self._flat or {}).values():
            f.zero_grad()

    def train(self):
        """:113-200: iterations that print nothing are queued without a read-back; the log line reads the running average."""
        from tqdm import trange
        c = self.config
        self.make_optimizers()
        rank0 = self.dp.rank == 0
        for iter in trange(c.start_iter, c.max_iter, disable=not rank0):
            data_list = self.data_loader.next(cl_ny="ny", type="train")
            if self.dp.active and getattr(self.data_loader, "dp", None) is None:
                data_list = self.dp.shard_collated(data_list)
            self.train_step_async(data_list, iter)
            if (iter + 1) % c.log_iter == 0:
                self.read_scalars()
                s = "[{}/{}] (train) CTC: {:.7f}".format(iter, c.max_iter, self.ctc_tr_local.avg)
                if rank0:
                    print(s)
                if self.logFile:
                    self.logFile.write(s + "\n")
                    self.logFile.flush()
                self.ctc_tr_local.reset()
            if (iter + 1) % c.save_iter == 0:
                if getattr(self, "_acc_live", False):
                    self.read_scalars()
                self._save_iter_block(iter)      # (rank 0 validates; rank 0's BatchNorm buffers of A to every rank afterwards)

    @ops.with_trainer_precision
    def greedy_decoding_and_AAS(self, inputs, targets, input_percentages, target_sizes, mask, transcript_prob=0.001):
        """Validation pass of trainer_acoustic.py:203-245: CTC and WER / CER only (the adversarial entries read 0)."""
        from .utils import _get_variable_volatile
        inputs = _get_variable_volatile(inputs)
        N = inputs.size(0)
        enhanced = self.G(inputs)
        prob, sizes, wer, cer, total_word, total_char = self._greedy_pass(enhanced, targets, input_percentages, target_sizes, transcript_prob)
        l_CTC = self.CTCLoss(prob, targets, sizes, target_sizes) / N
        return l_CTC, 0.0, 1, wer, cer, total_word, total_char


class _NoNet(torch.nn.Module):
    """Stand-in for the discriminator this trainer does not have (the shared checkpoint / device plumbing iterates G, D, ASR)."""

    def forward(self, *a, **k):
        raise RuntimeError("the acoustic_supervision trainer has no discriminator")
