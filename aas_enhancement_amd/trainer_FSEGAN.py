"""FSEGAN trainer on the MI355X HIP path - same API as the reference's Speech_enhancement_by_AAS/trainer_FSEGAN.py
(`Trainer.__init__ / build_model / load_model / train / zero_grad_all / greedy_decoding_and_FSEGAN`; hot loop :128-182, the save_iter
block with validation through the pre-trained acoustic model and the `G_<iter>.pth` / `G_valmin_<iter>.pth` lifecycle :199-275,
`greedy_decoding_and_FSEGAN` :277-317).

The reference file is unrunnable as written.  This builds the INTENDED step:
nFeat_in = nFeat_out = nFeat, D = stackedBRNN(I=2*nFeat, O=nFeat) fed through ``forward_paired``
everywhere (the reference's ``self.D(cleans, mixture)`` at :167 has the wrong arity), and the DCE
term IS back-propagated (G loss = dce + w_adversarial * adv).  ``config.fsegan_as_written=True``
keeps the reference's behaviour of only logging the DCE term (:161-163).
"""
import torch

from . import knobs, ops
from .decoder import GreedyDecoder
from .model import L1Loss_mask, stackedBRNN, supported_rnns
from .utils import _get_variable_nograd, _get_variable_volatile, attach_n_valid
from .validation import ValidationMixin


class Trainer(ops.TrainerContext, ValidationMixin):
    def __init__(self, config, data_loader=None, models=None):
        """models: (G, D) or (G, D, ASR) - networks built by the caller (tests / bench); None -> build_model() (:85-94)."""
        self._init_context()   # arithmetic mode + launch settings this trainer runs in (ops.TrainerContext)
        self.config, self.data_loader = config, data_loader
        self.lr, self.beta1, self.beta2 = config.lr, config.beta1, config.beta2
        self.optimizer = getattr(config, "optimizer", "adam")
        self.batch_size = config.batch_size
        self.diffLoss = L1Loss_mask()
        self.model_dir = "logs/" + str(config.expnum)
        self.kt, self.lb, self.gamma = 0, config.lambda_k, config.gamma
        self.conv_measure = 0
        self._init_validation_state(("dce_tr", "dce_tr_local", "dce_val", "adv_ny_tr", "adv_ny_val", "wer_tr", "wer_val", "cer_tr", "cer_val"))
        self.decoder = GreedyDecoder(data_loader.labels) if data_loader is not None and getattr(data_loader, "labels", None) else None
        self.as_written = getattr(config, "fsegan_as_written", False)
        self.ASR = None
        if models is not None:
            self.G, self.D = models[0], models[1]
            if len(models) > 2:
                self.ASR = models[2]
        else:
            self.build_model()
        self.G.loss_stop = 100000
        if config.gpu >= 0:
            self.G.cuda(); self.D.cuda()
            if self.ASR is not None:
                self.ASR.cuda()
        if len(getattr(config, "load_path", "")) > 0:
            self.load_model()
        self._open_log()
        self._opts = None
        self._flat = None
        self.dp = None

    def zero_grad_all(self):
        self.G.zero_grad(); self.D.zero_grad()
        if self.ASR is not None:
            self.ASR.zero_grad()

    def build_model(self):
        """:85-94 with nFeat_in = nFeat_out = nFeat and nFeat_D = 2 * nFeat (config.py defines none of the three: SURVEY 0.13)."""
        c = self.config
        rt = supported_rnns[c.rnn_type]
        print("initialize enhancement & discriminator model")
        self.G = stackedBRNN(I=c.nFeat, O=c.nFeat, H=c.rnn_size, L=c.rnn_layers, rnn_type=rt)
        self.D = stackedBRNN(I=2 * c.nFeat, O=c.nFeat, H=c.rnn_size, L=c.rnn_layers, rnn_type=rt)
        self.ASR = self.load_asr_package()

    def make_optimizers(self):
        """Adam(amsgrad) per network on flat parameter / gradient buffers (one fused launch each); the recurrent layers'
        weight gradients accumulate into them on the side stream (ops.sync_wgrad joins before they are read)."""
        from .dist import BucketReducer, DPContext, FlatBuffers
        from .optim import FlatAdam
        c = self.config
        self.dp = getattr(self, "dp", None) or DPContext.from_env()
        for name, m in (("G", self.G), ("D", self.D)):
            ops.name_layers(m, name)
        if self.ASR is not None:
            ops.name_layers(self.ASR, "ASR")
            for p in self.ASR.parameters():      # A only decodes in this trainer: no gradient ever reaches it
                p.requires_grad_(False)
        self._flat = {"G": FlatBuffers(self.G), "D": FlatBuffers(self.D)}
        mk = lambda f: FlatAdam(f, lr=c.lr, betas=(self.beta1, self.beta2), amsgrad=True)
        self._opts = (mk(self._flat["G"]), mk(self._flat["D"]))
        self._reducer = BucketReducer(self.dp, self._flat.values()) if self.dp.active else None
        return self._opts

    def get_gradient_norm(self, model):
        ops.sync_wgrad()
        acc = torch.zeros((1,), device=next(model.parameters()).device, dtype=torch.float64)
        flat = (self._flat or {}).get("G" if model is self.G else "D")
        if flat is not None:
            ops.sqsum_into(acc, flat.flat_g)
        else:
            for p in model.parameters():
                if p.grad is not None:
                    ops.sqsum_into(acc, p.grad)
        return acc.sqrt().to(torch.float32)

    def _batch(self, data_list):
        mask = data_list[2]
        if not mask.is_cuda:
            attach_n_valid(mask)
        elif getattr(mask, "n_valid", None) is None:
            # a loader that hands out device masks without the host-side count: one read-back here, so that both step forms
            # (and read_scalars() behind train_step_async) see the same state - the synchronous fallback read it back anyway
            mask.n_valid = int(mask.numel()) - int(mask.sum().item())
        return _get_variable_nograd(data_list[0]), _get_variable_nograd(data_list[1]), _get_variable_nograd(mask)

    def _device_step(self, mixture, cleans, nElement):
        """The step as library launches only (no torch glue): ONE prologue launch zeroes the flat gradient buffers and the three
        loss sums and writes the discriminator's per-utterance weights [-kt] * N + [1] * N; the batched D input is laid down
        time-major directly; the three L1 losses are raw device sums whose weights ride in their backward launches; the
        controller consumes the raw sums (aas_began_step_sums).  Data parallel: the normalisers are device scalars formed from
        the all-reduced nElement (dist.DeviceScales), the gradient buffers are all-reduced bucket by bucket behind the
        weight-gradient products.  -> (enhanced, scales): scales = python floats, or the DeviceScales when data parallel."""
        c, dp = self.config, self.dp
        N, dev = mixture.size(0), mixture.device
        if getattr(self, "_sums", None) is None or self._rs_pair.numel() != 2 * N:
            self._sums = torch.zeros(2, device=dev, dtype=torch.float64)       # raw L1 sums [adv_ny, adv_cl]
            self._sum3 = torch.zeros(2, device=dev, dtype=torch.float64)       # [dce, -] (a 16-byte block of its own)
            self._rs_pair = torch.empty(2 * N, device=dev, dtype=torch.float32)
        ops.sync_wgrad()
        if dp.active:
            from .dist import DeviceScales
            cnt = self._upload_small(torch.tensor([float(nElement)], dtype=torch.float64), dev)
            scales = DeviceScales(dp, cnt, [c.w_adversarial, c.w_adversarial, 1.0], [0, 0, 0], ops.refresh_stream(dev))
            if self._kt_ev is not None:     # kt was advanced on the auxiliary stream behind the previous step's scalar all-reduce
                torch.cuda.current_stream().wait_event(self._kt_ev)
        else:
            scales = (c.w_adversarial / nElement, c.w_adversarial / nElement, 1.0 / nElement)
        ops.step_prologue([f.flat_g for f in self._flat.values()] + [self._sums, self._sum3], self._rs_pair, N, N, self._kt_dev)
        w = self._rs_pair.detach()
        rs = ops.RowWeights(w, classes=[(0, N, w[0:1]), (N, N, None)])   # the two utterance classes and their weights (device scalars)
        if dp.active:
            self._reducer.begin()
            self.launch.wgrad_hook = self._reducer.on_wgrad
        try:
            enhanced = self.G(mixture)
            leaf = enhanced.detach().requires_grad_(True)
            ae = self.D(None, wgrad_row_scale=rs, tnc=ops.layout_paired_cat(leaf, mixture, cleans))
            tgrad = []
            l_pair = ops.l1_pair(ae, leaf, cleans, scales[0], scales[1], self._sums.detach(), tgrad)
            l_dce = ops.l1_scaled(leaf, cleans, scales[2], self._sum3, tgrad)
            roots = [l_pair] if self.as_written else [l_pair, l_dce]     # (:161-163: the reference only logs the DCE term)
            # (knobs.FSEGAN_BWD_CUS / FSEGAN_DEFER_D: CU budget of the BPTT launches and held-back discriminator layers, as in the AAS
            #  step; measured flat for this single-chain step - profiles/r06_fsegan_sweep.txt - so both default to "off")
            ops.set_rnn_cu_limit(knobs.get("FSEGAN_BWD_CUS"))
            ndef = int(knobs.get("FSEGAN_DEFER_D"))
            if ndef > 0:
                lids = [m._aas_layer_id for m in self.D.modules() if getattr(m, "_aas_layer_id", None) is not None]
                self.launch.defer_lids.update(lids[-ndef:])
            try:
                torch.autograd.backward(roots, [ops.unit_root(r_) for r_ in roots])
            finally:
                self.launch.defer_lids.clear()
            ops.flush_deferred_wgrad()
            if dp.active:   # D's small parameters; its layer buckets are in flight: overlaps E's backward
                self._reducer.flush(self._flat["D"])
            # the gradients arriving at `enhanced`: through D's input, as the target of the adversarial L1, from the DCE term
            parts = [leaf.grad] + tgrad
            gsum = ops.add3(parts[0], parts[1], parts[2] if len(parts) > 2 else None)
            enhanced.backward(gsum)
            ops.set_rnn_cu_limit(0)
            ops.sync_wgrad()
            if dp.active:
                self._reducer.flush(self._flat["G"])
                self._reducer.wait()
        finally:
            self.launch.wgrad_hook = None
            ops.set_rnn_cu_limit(0)
        return enhanced, scales

    def _controller(self, scales, nElement):
        """The BEGAN controller (:175-179) and the log scalars from the raw sums in one launch; slot 2 carries the DCE term.  Data
        parallel: on the auxiliary stream, behind the all-reduce of the three sums; the main stream only waits for its event where
        kt is next read (the step prologue)."""
        if not self.dp.active:
            ops.began_step_sums(self._sums, self._sum3, scales[0], scales[1], scales[2], self._kt_dev, self._g_out, self.gamma, self.lb, float(nElement))
            return
        main, aux = torch.cuda.current_stream(), ops.refresh_stream(self._sums.device)
        aux.wait_stream(main)
        with torch.cuda.stream(aux):
            out3 = torch.empty(3, device=self._sums.device, dtype=torch.float64)
            ops.sums_pack(self._sums, self._sum3, out3)
            self.dp.reduce_scalars(out3)
            ops.began_step_sums(out3, out3[2:], 1.0, 1.0, 1.0, self._kt_dev, self._g_out, self.gamma, self.lb, 0.0, d_scales3=scales.all, d_n_batch=scales.cnt)
            self._kt_ev = torch.cuda.Event()
            self._kt_ev.record(aux)
        for t_ in (scales.all, scales.cnt):
            t_.record_stream(aux)

    def _ensure_dev_state(self, dev):
        if getattr(self, "_kt_dev", None) is None:
            self._kt_dev = torch.zeros(1, device=dev, dtype=torch.float64)
            self._g_out = torch.zeros(6, device=dev, dtype=torch.float64)
            self._kt_ev = None
            self._kt_dev_live = False
        if not self._kt_dev_live:
            self._kt_dev.fill_(float(self.kt))     # (the host copy is the current one: a synchronous step / a checkpoint advanced it)
        self._kt_dev_live = True

    @ops.with_trainer_precision
    def train_step(self, data_list, iter=0):
        """:128-182 (intended semantics), host-synchronous: the same launches as `train_step_async` plus the gradient
        norm of G before the update and ONE read-back; returns the scalars of the log lines.  Data parallel: `data_list` is this
        rank's shard; the losses are normalised by the GLOBAL nElement, so the SUM all-reduce of the flat gradient buffers gives
        the single-process gradients."""
        if self._opts is None:
            self.make_optimizers()
        optimizer_g, optimizer_d = self._opts
        mixture, cleans, mask = self._batch(data_list)
        nElement = mask.n_valid
        self._ensure_dev_state(mixture.device)
        enhanced, scales = self._device_step(mixture, cleans, nElement)
        g_norm = self.get_gradient_norm(self.G)
        optimizer_g.step_dev(); optimizer_d.step_dev()
        ops.refresh_weight_planes(self.G); ops.refresh_weight_planes(self.D)
        self._controller(scales, nElement)
        r = self.read_scalars()
        r.update(g_norm=float(g_norm), enhanced=enhanced)
        return r

    # ---- the same step without a host synchronisation (what train() queues on iterations that print nothing) -----------
    @ops.with_trainer_precision
    def train_step_async(self, data_list, iter=0):
        """train_step queued WITHOUT reading anything back: kt, the Adam bias corrections, the loss sums and the running DCE
        average of the log line stay on the device (`aas_began_step_sums`, `FlatAdam.step_dev`), so the host queues step i+1
        while the GPU runs step i.  `read_scalars()` (one D2H copy) returns the last step's losses and updates the host-side kt.
        Library launches only (`_device_step`).  Data parallel: the global nElement is all-reduced on the utility stream and kept
        on the device, the gradient buffers are all-reduced bucket by bucket behind the weight-gradient products
        (dist.BucketReducer), and kt is advanced from the all-reduced loss sums - still no host synchronisation."""
        if self._opts is None:
            self.make_optimizers()
        optimizer_g, optimizer_d = self._opts
        mixture, cleans, mask = self._batch(data_list)
        nElement = mask.n_valid
        self._ensure_dev_state(mixture.device)
        enhanced, scales = self._device_step(mixture, cleans, nElement)
        optimizer_g.step_dev(); optimizer_d.step_dev()
        ops.refresh_weight_planes(self.G); ops.refresh_weight_planes(self.D)
        self._controller(scales, nElement)
        return dict(enhanced=enhanced, scalars=self._g_out)

    def read_scalars(self):
        """One D2H copy of (l_adv_ny_G, l_adv_cl, dce, kt) of the last train_step_async; updates the host-side kt and the running
        DCE average; a synchronisation point: raises if a persistent kernel timed out or the run diverged."""
        if getattr(self, "_kt_ev", None) is not None:
            torch.cuda.current_stream().wait_event(self._kt_ev)
        l_adv_ny_G, l_adv_cl, dce, kt, dce_sum, dce_n = self._g_out.tolist()
        self._g_out[4:6].zero_()
        ops.check_rnn_health((l_adv_ny_G, l_adv_cl, dce))
        self.kt = kt
        self._kt_dev_live = False
        if dce_n > 0:
            self.dce_tr_local.update(dce_sum / dce_n, dce_n)
        bal = self.gamma * l_adv_cl - l_adv_ny_G
        return dict(l_adv_ny_G=l_adv_ny_G, l_adv_cl=l_adv_cl, dce=dce, kt=kt, conv_measure=l_adv_cl + abs(bal))

    def train(self):
        """:125-182.  Iterations that print nothing are queued with train_step_async; logging iterations read the scalars back."""
        from tqdm import trange
        c = self.config
        self.make_optimizers()
        rank0 = self.dp.rank == 0
        presharded = getattr(self.data_loader, "dp", None) is not None
        for iter in trange(c.start_iter, c.max_iter, disable=not rank0):
            data = self.data_loader.next(cl_ny="ny", type="train")
            if self.dp.active and not presharded:
                data = _shard_paired(self.dp, data)
            self.train_step_async(data, iter)
            if (iter + 1) % c.log_iter == 0:
                r = self.read_scalars()
                for s in ("[{}/{}] (train) DCE: {:.7f}, ADV_cl: {:.7f}, ADV_ny: {:.7f}".format(iter, c.max_iter, self.dce_tr_local.avg, r["l_adv_cl"], r["l_adv_ny_G"]),
                          "[{}/{}] (train) conv_measure: {:.4f}, kt: {:.4f} ".format(iter, c.max_iter, r["conv_measure"], self.kt)):
                    if rank0:
                        print(s)
                    if self.logFile:
                        self.logFile.write(s + "\n")
                if self.logFile:
                    self.logFile.flush()
                self.dce_tr_local.reset()
            if (iter + 1) % c.save_iter == 0:
                if getattr(self, "_kt_dev_live", False):
                    self.read_scalars()
                self._save_iter_block(iter)

    # ---- validation + checkpoint lifecycle (:199-275) -------------------------------------------------------------------------
    @ops.with_trainer_precision
    def validate_and_checkpoint(self, iter):
        c = self.config
        if self.ASR is None or self.decoder is None:
            raise RuntimeError("FSEGAN validation decodes through the pre-trained acoustic model: --ASR_path (build_model) or "
                               "models=(G, D, ASR), and a data loader with `labels`, are needed")
        self.G.eval()
        for (name, dl), (dce_m, adv_m, wer_m, cer_m) in zip(self._validation_sets(), ((self.dce_tr, self.adv_ny_tr, self.wer_tr, self.cer_tr),
                                                                                      (self.dce_val, self.adv_ny_val, self.wer_val, self.cer_val))):
            for m in (dce_m, adv_m, wer_m, cer_m):
                m.reset()
            for _ in range(self.data_loader.num_batches(dl)):
                d = self.data_loader.next(cl_ny="ny", type=dl)
                with torch.no_grad():
                    dce, adv_ny, nElement, wer, cer, nWord, nChar = self.greedy_decoding_and_FSEGAN(d[0], d[1], d[3], d[4], d[5], d[2])
                dce_m.update(float(dce), nElement); adv_m.update(float(adv_ny), nElement)
                wer_m.update(wer, nWord); cer_m.update(cer, nChar)
            # (sic: the reference labels the DCE average "CTC" in these two lines, :225,:252)
            self._log("[{}/{}] ({}) CTC: {:.7f}, WER: {:.7f}, CER: {:.7f}".format(iter, c.max_iter, name, dce_m.avg, wer_m.avg * 100, cer_m.avg * 100))
        if self.logFile:
            self.logFile.flush()
        self.G.train()   # end of validation
        self._save_rotating("G", self.G, iter)
        self._keep_if_best(iter, self.wer_val.avg, ("G",))

    @ops.with_trainer_precision
    def greedy_decoding_and_FSEGAN(self, mixture, cleans, targets, input_percentages, target_sizes, mask, transcript_prob=0.001):
        """:277-317 -> (dce, l_adv_ny, nElement, wer, cer, total_word, total_char); D through forward_paired (its :297 too)."""
        mixture, cleans = _get_variable_volatile(mixture), _get_variable_volatile(cleans)
        attach_n_valid(mask) if not mask.is_cuda else None
        mask = _get_variable_volatile(mask)
        enhanced = self.G(mixture)
        _, _, wer, cer, total_word, total_char = self._greedy_pass(enhanced, targets, input_percentages, target_sizes, transcript_prob)
        ae_ny = self.D.forward_paired(enhanced, mixture)
        l_adv_ny, nElement = self.diffLoss(ae_ny, enhanced, mask)
        l_adv_ny = l_adv_ny * self.config.w_adversarial
        dce, nElement_ = self.diffLoss(enhanced, cleans, mask)
        assert (nElement == nElement_)
        return dce, l_adv_ny, nElement, wer, cer, total_word, total_char


def _shard_paired(dp, data):
    """Strided shard of a `_collate_fn_paired` tuple (inputs, cleans, mask, targets, pct, target_sizes) handed out by a loader
    that is not data-parallel aware."""
    inputs, cleans, mask, targets, pct, tsz = data
    ny = dp.shard_collated((inputs, targets, pct, tsz, mask))
    idx = torch.tensor(dp.shard_rows(inputs.size(0)), dtype=torch.long)
    return (ny[0], cleans.index_select(0, idx.to(cleans.device)), ny[4], ny[1], ny[2], ny[3])
