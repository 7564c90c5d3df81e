"""AAS trainer on the MI355X HIP path - same API as the reference's
Speech_enhancement_by_AAS/trainer_AAS.py (Trainer.__init__/build_model/train/zero_grad_all/
get_gradient_norm/greedy_decoding_and_AAS; hot loop :131-194).

``train_step`` is the loop body of ``train()`` (:131-194) factored out so that bench.py and the
parity tests can drive exactly the code ``main.py --trainer AAS`` runs.

Schedules (results identical, SURVEY.md 3.1 / 8a "verified algebraic identities"):
  * ``schedule='as_executed'``: the reference's literal order - D forward x3, E backward x2.
  * ``schedule='fused'`` (default): D(enhanced) is evaluated once; the D-step parameter gradients
    are (-kt) x the G-step's D-parameter gradients that the reference computes and then discards at
    :152, so no second D forward/backward is run.
"""
import torch

from . import knobs, ops
from .ctc import CTCLoss
from .decoder import GreedyDecoder
from .model import L1Loss_mask, stackedBRNN, supported_rnns
from .utils import _get_variable_nograd, _get_variable_volatile, attach_n_valid
from .validation import ValidationMixin


class Trainer(ops.TrainerContext, ValidationMixin):
    def __init__(self, config, data_loader=None, models=None):
        self.config = config
        self.data_loader = data_loader
        self.lr, self.beta1, self.beta2 = config.lr, config.beta1, config.beta2
        self.optimizer = getattr(config, "optimizer", "adam")
        self.batch_size = config.batch_size
        self.schedule = getattr(config, "schedule", "fused")
        self.diffLoss = L1Loss_mask()
        self.model_dir = "logs/" + str(config.expnum)
        self.kt = 0  # BEGAN proportional controller state (:47)
        self.lb = config.lambda_k
        self.gamma = config.gamma
        self.conv_measure = 0
        self._init_validation_state(("ctc_tr", "ctc_tr_local", "ctc_val", "adv_ny_tr", "adv_ny_val", "wer_tr", "wer_val", "cer_tr", "cer_val"))
        self.CTCLoss = CTCLoss()
        self.decoder = GreedyDecoder(data_loader.labels) if data_loader is not None and getattr(data_loader, "labels", None) else None
        if models is not None:
            self.G, self.D, self.ASR = models
        else:
            self.build_model()
        self.G.loss_stop = 100000
        if config.gpu >= 0:
            self.G.cuda(); self.D.cuda(); self.ASR.cuda()
        if len(getattr(config, "load_path", "")) > 0:
            self.load_model()
        self._open_log()
        self._opts = None
        self._flat = None
        self.dp = None
        # the arithmetic this trainer runs in: the library's setting when it is built (config.precision / AAS_PRECISION / fp32);
        # every step and validation entry point runs under ops.precision(self.precision); set_precision() switches it
        self._init_context()
        # a parameter of D is used from two streams when the step switches schedules: autograd synchronises the accumulation
        # correctly and says so once per process; the notice is not actionable here
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)

    def build_model(self):
        print("initialize enhancement & discriminator model")
        c = self.config
        self.G = stackedBRNN(I=c.nFeat, H=c.rnn_size, L=c.rnn_layers, rnn_type=supported_rnns[c.rnn_type])
        self.D = stackedBRNN(I=c.nFeat, H=c.rnn_size, L=c.rnn_layers, rnn_type=supported_rnns[c.rnn_type])
        self.ASR = self.load_asr_package()

    # load_model (:98-123): ValidationMixin.load_model - G from `<load_path>/G_valmin_<iter>.pth`, newest when start_iter <= 0

    # ------------------------------------------------------------------------------------------
    def asr_frozen(self):
        """north_star's "frozen A": optimizer_asr never steps when allow_ASR_update_iter >= max_iter (:187)."""
        c = self.config
        return c.allow_ASR_update_iter >= getattr(c, "max_iter", 30000000) - 1

    def make_optimizers(self):
        """Adam(amsgrad) per network (:127-129) on flat parameter/gradient buffers: one fused HIP launch,
        one grad-norm launch and (data parallel) one RCCL all-reduce per network."""
        from .dist import BucketReducer, DPContext, FlatBuffers
        from .optim import FlatAdam
        c = self.config
        self.dp = getattr(self, "dp", None) or DPContext.from_env()
        for name, m in (("G", self.G), ("D", self.D), ("ASR", self.ASR)):
            ops.name_layers(m, name)
        self._frozen_asr = self.asr_frozen()
        if self._frozen_asr:
            for p in self.ASR.parameters():
                p.requires_grad_(False)  # its gradients are never applied: skip the wgrad GEMMs
        self._flat = {"G": FlatBuffers(self.G), "D": FlatBuffers(self.D)}
        if not self._frozen_asr:
            self._flat["A"] = FlatBuffers(self.ASR)
        mk = lambda f: FlatAdam(f, lr=c.lr, betas=(self.beta1, self.beta2), amsgrad=True)
        self._opts = (mk(self._flat["G"]), mk(self._flat["A"]) if "A" in self._flat else None, mk(self._flat["D"]))
        # (recurrent-layer weight gradients of FlatBuffers-owned parameters accumulate into the flat buffers on a side stream)
        self._reducer = BucketReducer(self.dp, self._flat.values()) if self.dp.active else None
        # data parallel: A's BatchNorm statistics over the GLOBAL batch (--sync_bn) instead of "8 replicas with local-batch BN"
        self.launch.sync_bn = self.dp if (self.dp.active and getattr(c, "sync_bn", False)) else None
        return self._opts

    def zero_grad_all(self):
        if getattr(self, "_flat", None):
            ops.sync_wgrad()
            for f in self._flat.values():
                f.zero_grad()
        else:
            self.G.zero_grad(); self.D.zero_grad(); self.ASR.zero_grad()

    def get_gradient_norm(self, model):
        """sqrt(sum_p sum grad^2) (:353-361): fp64 device accumulator, one launch on the flat buffer."""
        dev = next(model.parameters()).device
        ops.sync_wgrad()
        acc = torch.zeros((1,), device=dev, dtype=torch.float64)
        flat = getattr(self, "_flat", None)
        key = "G" if model is self.G else ("D" if model is self.D else "A")
        if flat and key in flat:
            ops.sqsum_into(acc, flat[key].flat_g)
        else:
            for p in model.parameters():
                if p.grad is not None:
                    ops.sqsum_into(acc, p.grad if p.grad.is_contiguous() else p.grad.contiguous())
        return acc.sqrt().to(torch.float32)

    def _prep(self, data_list):
        inputs, targets, pct, target_sizes, mask = data_list[0], data_list[1], data_list[2], data_list[3], data_list[4]
        if not mask.is_cuda:
            attach_n_valid(mask)
        return (_get_variable_nograd(inputs), targets, pct, target_sizes, _get_variable_nograd(mask))

    @ops.with_trainer_precision
    def train_step(self, data_list, data_list_cl, iter, log_norms=True):
        """One iteration of :131-194.  Returns the host scalars the reference logs.
        Data parallel (world > 1): `data_list*` are this rank's shards; losses are normalised by the
        global N / nElement so that summed gradients equal the single-process ones."""
        if self._opts is None:
            self.make_optimizers()
        optimizer_g, optimizer_asr, optimizer_d = self._opts
        c, dp = self.config, self.dp
        if getattr(self, "_kt_dev_live", False):
            self.read_scalars()   # queued train_step_async steps advanced kt on the device: fetch it (one sync)
        self.zero_grad_all()
        inputs, targets, input_percentages, target_sizes, mask = self._prep(data_list)
        cl_inputs, cl_mask = data_list_cl[0], data_list_cl[4]
        if not cl_mask.is_cuda:
            attach_n_valid(cl_mask)
        cl_inputs, cl_mask = _get_variable_nograd(cl_inputs), _get_variable_nograd(cl_mask)
        N = inputs.size(0)
        # CTC metadata goes to the device before any kernel is queued (a pageable H2D copy blocks the host until the
        # stream drains; done lazily inside the loss it would stall the host behind the whole acoustic model)
        t_out = self.ASR.output_length(inputs.size(2))
        sizes = input_percentages.clone().mul_(int(t_out)).int()
        ctc_meta = self.CTCLoss.prepare(targets, sizes, target_sizes, inputs.device)
        if dp.active:
            nv = lambda m: getattr(m, "n_valid", None) or (int(m.numel()) - int(m.sum().item()))
            N_glob, nv_ny, nv_cl = dp.global_counts([N, nv(mask), nv(cl_mask)])
            mask.n_valid, cl_mask.n_valid = nv_ny, nv_cl
        else:
            N_glob = N
        enhanced = self.G(inputs)
        g_adv = g_ctc_adv = None
        handle_d = None
        if self.schedule == "as_executed":
            enhanced_D = enhanced.detach()
            ae_ny_G = self.D(enhanced)
            l_adv_ny_G, _ = self.diffLoss(ae_ny_G, enhanced, mask)
            l_adv_ny_G = l_adv_ny_G * c.w_adversarial
            l_adv_ny_G.backward(retain_graph=True)
            if log_norms:
                g_adv = self.get_gradient_norm(self.G)
            ops.sync_wgrad()
            self._flat["D"].zero_grad()   # == self.D.zero_grad() (:152)
            ae_ny_D = self.D(enhanced_D)
            l_adv_ny_D, _ = self.diffLoss(ae_ny_D, enhanced_D, mask)
            l_adv_ny_D = l_adv_ny_D * (-self.kt) * c.w_adversarial
            l_adv_ny_D.backward()
            del l_adv_ny_D
            prob = self.ASR(enhanced).transpose(0, 1)
            assert prob.size(0) == t_out
            l_CTC = c.w_acoustic * self.CTCLoss(prob, targets, sizes, target_sizes, prepared=ctc_meta) / N_glob
            l_CTC.backward()
            if log_norms:
                g_ctc_adv = self.get_gradient_norm(self.G)
            ae_cl = self.D(cl_inputs)
            l_adv_cl, _ = self.diffLoss(ae_cl, cl_inputs, cl_mask)
            l_adv_cl = c.w_adversarial * l_adv_cl
            l_adv_cl.backward()
        else:
            # E is back-propagated ONCE: both losses are taken on a detached leaf of `enhanced`, their
            # gradients wrt it add up (linear), then a single backward runs through E.  On logging
            # iterations g_adv needs E's adversarial-only gradients, so E is back-propagated per loss.
            leaf = enhanced.detach().requires_grad_(True)
            Nn = leaf.size(0)
            overlap = self._overlap_asr()
            acoustic = None
            same_len = tuple(cl_inputs.shape[1:]) == tuple(leaf.shape[1:])
            interleave = same_len and not log_norms and self._interleave_ok()
            if overlap:  # two chains of persistent launches side by side, half the chip each
                ops.set_rnn_cu_limit(ops.device_cus() // 2)
                if not interleave:
                    acoustic = self._acoustic_branch(enhanced, targets, sizes, target_sizes, N_glob, ctc_meta)
            if interleave:
                w = torch.empty(Nn + cl_inputs.size(0), device=leaf.device, dtype=torch.float32)
                w[:Nn] = -float(self.kt)
                w[Nn:] = 1.0
                rs = ops.RowWeights(w, classes=[(0, Nn, w[0:1]), (Nn, cl_inputs.size(0), None)])
                l_adv_ny_G, l_adv_cl, prob, l_CTC, leaf_a = self._interleaved_DA(enhanced, leaf, cl_inputs, rs, N_glob, ctc_meta, targets, sizes,
                                                                                target_sizes, mask=mask, cl_mask=cl_mask)
                acoustic = (prob, l_CTC, leaf_a)
            elif same_len:
                # D(enhanced) and D(clean) share ONE batched pass (rows are independent: D has no batch
                # statistics).  The D-step gradients of the enhanced half are (-kt) x its G-step parameter
                # gradients (:152-160), applied as per-utterance weights on the weight-gradient products only,
                # so the gradient flowing back to `enhanced` stays the G-step one.
                w = torch.empty(Nn + cl_inputs.size(0), device=leaf.device, dtype=torch.float32)
                w[:Nn] = -float(self.kt)
                w[Nn:] = 1.0
                rs = ops.RowWeights(w, classes=[(0, Nn, w[0:1]), (Nn, cl_inputs.size(0), None)])
                ae = self.D(torch.cat([leaf, cl_inputs], 0), wgrad_row_scale=rs)
                l_adv_ny_G, _ = self.diffLoss(ae[:Nn], leaf, mask)
                l_adv_cl, _ = self.diffLoss(ae[Nn:], cl_inputs, cl_mask)
                l_adv_ny_G = l_adv_ny_G * c.w_adversarial
                l_adv_cl = c.w_adversarial * l_adv_cl
                (l_adv_ny_G + l_adv_cl).backward()
            else:  # different padded lengths: two D passes, D-step = (-kt) x G-step parameter gradients
                ae_ny_G = self.D(leaf)
                l_adv_ny_G, _ = self.diffLoss(ae_ny_G, leaf, mask)
                l_adv_ny_G = l_adv_ny_G * c.w_adversarial
                l_adv_ny_G.backward()
                ops.sync_wgrad()
                ops.axpby_(self._flat["D"].flat_g, self._flat["D"].flat_g, -float(self.kt), 0.0)
                ae_cl = self.D(cl_inputs)
                l_adv_cl, _ = self.diffLoss(ae_cl, cl_inputs, cl_mask)
                l_adv_cl = c.w_adversarial * l_adv_cl
                l_adv_cl.backward()
            if log_norms:
                enhanced.backward(leaf.grad, retain_graph=True)
                g_adv = self.get_gradient_norm(self.G)
                leaf.grad = None
            if dp.active:  # D's all-reduce overlaps the acoustic branch and E's backward
                ops.sync_wgrad()
                handle_d = dp.allreduce_sum_(self._flat["D"].flat_g, async_op=True)
            # CTC loss (:163-172).  The acoustic branch A(enhanced) -> CTC -> backward is independent of the D branch
            # above: with AAS_OVERLAP_ASR=1 it was queued on a second stream before it, else it follows it here.
            if acoustic is None:
                acoustic = self._acoustic_branch(enhanced, targets, sizes, target_sizes, N_glob, ctc_meta)
            prob, l_CTC, leaf_a = acoustic
            torch.cuda.current_stream().wait_stream(self._side)
            if overlap:
                ops.set_rnn_cu_limit(0)
            leaf_a.grad.record_stream(torch.cuda.current_stream())
            # (on logging iterations the adversarial part was already back-propagated for g_adv)
            gsum = ops.add3(leaf.grad, leaf_a.grad) if leaf.grad is not None else leaf_a.grad
            ops.set_rnn_cu_limit(knobs.get("EBWD_CUS"))  # leave CUs to E's weight-gradient GEMMs
            enhanced.backward(gsum)
            ops.set_rnn_cu_limit(0)
            if log_norms:
                g_ctc_adv = self.get_gradient_norm(self.G)
        ops.sync_wgrad()  # join the side-stream weight-gradient products before the gradients are consumed
        # data parallel: SUM all-reduce of the flat gradient buffers (RCCL over xGMI)
        if dp.active:
            if handle_d is None:
                handle_d = dp.allreduce_sum_(self._flat["D"].flat_g, async_op=True)
            hs = [handle_d, dp.allreduce_sum_(self._flat["G"].flat_g, async_op=True)]
            if "A" in self._flat and iter > c.allow_ASR_update_iter:
                hs.append(dp.allreduce_sum_(self._flat["A"].flat_g, async_op=True))
            for h in hs:
                h.wait()
        # update (:184-188)
        optimizer_g.step()
        optimizer_d.step()
        if optimizer_asr is not None and iter > c.allow_ASR_update_iter:
            optimizer_asr.step()
            ops.refresh_weight_planes(self.ASR)
        ops.refresh_weight_planes(self.G)
        ops.refresh_weight_planes(self.D)
        # one packed device->host read for the three scalars the controller / log need
        packed = torch.stack([l_adv_ny_G.detach().reshape(()), l_adv_cl.detach().reshape(()), l_CTC.detach().reshape(())])
        l_adv_ny_G_data, l_adv_cl_data, l_ctc_data = dp.reduce_scalars(packed).tolist()
        ops.check_rnn_health((l_adv_ny_G_data, l_adv_cl_data, l_ctc_data))   # (the read-back above synchronised)
        self.ctc_tr_local.update(l_ctc_data, N_glob)
        # Proportional Control Theory (:190-194)
        g_d_balance = self.gamma * l_adv_cl_data - l_adv_ny_G_data
        self.kt += self.lb * g_d_balance
        self.kt = max(min(1, self.kt), 0)
        self._kt_dev_live = False   # the device-resident copy (train_step_async / graph) is stale now
        conv_measure = l_adv_cl_data + abs(g_d_balance)
        return dict(l_adv_ny_G=l_adv_ny_G_data, l_adv_cl=l_adv_cl_data, l_ctc=l_ctc_data, kt=self.kt,
                    conv_measure=conv_measure, g_adv=g_adv, g_ctc_adv=g_ctc_adv, enhanced=enhanced, prob=prob)

    # ---- device-resident step ---------------------------------------------------------------------------
    # The whole fused iteration (zero-grad .. Adam .. kt update) as ONE device-only launch sequence: kt, the Adam bias corrections
    # and the loss scalars stay on the device, so the host queues step i+1 while the GPU runs step i.  Same kernels, same results
    # as train_step(schedule='fused', log_norms=False).  (Rounds 1-5 also replayed this sequence from a captured hipGraph,
    # `train_step_graph`: the graph executor ran the two interleaved chains one after the other - 26.4 vs 21.7 ms in the fast
    # mode - and capture forced the exchange buffers and split-K workspaces onto their slow paths; removed in round 6)
    def _device_core(self, inputs, cl_inputs, nv_ny, nv_cl, ctc_meta, it=None):
        """The fused iteration as a device-only launch sequence.  Data parallel: the loss normalisers are the all-reduced
        GLOBAL counts (kept on the device), the gradient buffers are all-reduced bucket by bucket as the weight-gradient
        products finish (dist.BucketReducer), and kt is advanced from the all-reduced loss scalars - still no host sync."""
        c, dp = self.config, self.dp
        optimizer_g, optimizer_asr, optimizer_d = self._opts
        asr_steps = optimizer_asr is not None and (it is None or it > c.allow_ASR_update_iter)
        ops.sync_wgrad()
        N = inputs.size(0)
        dev = inputs.device
        # device-resident glue as library launches (knobs.FUSED_GLUE; equal padded lengths): ONE prologue launch zeroes the flat
        # gradient buffers and the loss accumulators and writes the discriminator's per-utterance weights [-kt]*N + [1]*N; the
        # losses stay raw device sums until the controller launch consumes them (no scaling / slicing / summing launches).  Data
        # parallel: the same, with the normalisers as device scalars formed from the all-reduced counts and the three raw loss sums
        # all-reduced before the controller (aas_loss_pack -> all-reduce -> aas_began_step_sums)
        # (a ragged noisy / clean pair on the batched schedule - two row classes inside D's recurrent launches - takes the same path:
        #  its L1 sums run on the row-strided kernels over each class's own frames)
        batched = tuple(cl_inputs.shape) == tuple(inputs.shape) or self._ragged_batched_ok(inputs, cl_inputs)
        fused = knobs.get("FUSED_GLUE") and batched and not self._lanes_ok(True) and self._interleave_ok()
        self._fused = None
        if dp.active:
            cnt = self._upload_small(torch.tensor([float(N), float(nv_ny), float(nv_cl)], dtype=torch.float64), dev)
            # global N, nElement(noisy), nElement(clean): all-reduced asynchronously and kept on the device; a loss only waits for
            # them where it is formed (after the forward passes), so the collective's latency hides behind E
            if getattr(self, "_aux_stream", None) is None:
                self._aux_stream = ops.refresh_stream(dev)
            from .dist import DeviceScales
            scales = DeviceScales(dp, cnt, [c.w_adversarial, c.w_adversarial, c.w_acoustic], [1, 2, 0], self._aux_stream)
            n_glob = scales
        else:
            scales = (c.w_adversarial / nv_ny, c.w_adversarial / nv_cl, c.w_acoustic / N)
            n_glob = float(N)
        if fused:
            Nc = cl_inputs.size(0)
            if getattr(self, "_l1_acc", None) is None or self._rs_pair.numel() != N + Nc:
                self._l1_acc = torch.zeros(2, device=dev, dtype=torch.float64)
                self._rs_pair = torch.empty(N + Nc, device=dev, dtype=torch.float32)
            self._wait_kt()      # (data parallel: kt was advanced on the auxiliary stream behind the previous step's scalar all-reduce)
            ops.step_prologue([f.flat_g for f in self._flat.values()] + [self._l1_acc], self._rs_pair, N, Nc, self._kt_dev)
            self._fused = {}
        elif knobs.get("FUSED_GLUE") and not dp.active:
            # the other single-process device paths (ragged pair on the two-lane schedule, ...): the zero fills and the weights
            # [-kt] * N of the enhanced pass in one prologue launch; their losses keep the autograd scaling
            if getattr(self, "_rs_lane", None) is None or self._rs_lane.numel() != N:
                self._rs_lane = torch.empty(N, device=dev, dtype=torch.float32)
            ops.step_prologue([f.flat_g for f in self._flat.values()], self._rs_lane, N, 0, self._kt_dev)
        else:
            self._rs_lane = None
            ops.step_prologue([f.flat_g for f in self._flat.values()])
        if dp.active:
            self._reducer.begin()
            if not self._reducer.deferred:     # E's first recurrent layer is back-propagated last
                first = next((m for m in self.G.modules() if getattr(m, "_aas_layer_id", None) is not None), None)
                if first is not None:
                    self._reducer.defer_layer([p_.grad for p_ in (first.weight_ih_l0, first.weight_hh_l0, first.weight_ih_l0_reverse,
                                                                  first.weight_hh_l0_reverse) if p_.grad is not None])
            self.launch.wgrad_hook = self._reducer.on_wgrad
        try:
            # two chains of half-chip persistent launches run side by side: the weight-gradient GEMMs never take more than the
            # other half of the CUs, so a recurrent launch always finds its CUs (AAS_WGRAD_WGS overrides)
            lanes = self._lanes_ok(tuple(cl_inputs.shape) == tuple(inputs.shape) or self._ragged_batched_ok(inputs, cl_inputs))
            # (ops.set_wgrad_cap / AAS_WGRAD_WGS: a grid cap on the weight-gradient GEMMs, so that they never hold more than half
            #  of the CUs while a recurrent launch waits to become resident, bought 0.4 ms before the XCD-aware recurrent launches;
            #  since then it is worth +-0.05 ms with a frozen A and costs 1.5 ms with a trainable one: off by default)
            self._last_schedule = "lanes" if lanes else ("batched" if cl_inputs.size(2) == inputs.size(2) else "batched-ragged")
            if lanes:
                enhanced, prob, l_adv_ny_G, l_adv_cl, l_CTC = self._two_lane_core(inputs, cl_inputs, scales, ctc_meta, asr_steps)
            else:
                enhanced, prob, l_adv_ny_G, l_adv_cl, l_CTC = self._batched_D_core(inputs, cl_inputs, scales, ctc_meta, asr_steps)
            ops.set_rnn_cu_limit(0)
            ops.sync_wgrad()
            if dp.active:
                self._reducer.flush(self._flat["G"])
                self._reducer.wait()
        except BaseException:
            self._early_adam = False
            raise
        finally:
            self.launch.wgrad_hook = None
        with ops._timed("mark", "optimizer_g.step_dev", 0.0):      # (ops.Profiler class "mark": where the step's first consumer of the all-reduced gradients is queued)
            optimizer_g.step_dev()
        if not getattr(self, "_early_adam", False):
            optimizer_d.step_dev()
            if asr_steps:
                optimizer_asr.step_dev()
        self._early_adam = False
        # the updated weights' operand planes for the next step, off the critical path (weight-gradient stream)
        for net, on in ((self.G, True), (self.D, True), (self.ASR, asr_steps)):
            if on:
                ops.refresh_weight_planes(net)
        if not dp.active:   # controller + log scalars in one tiny launch
            if self._fused:     # from the raw sums: L = scale x sum inside the launch
                ops.began_step_raw(self._fused["l1"], scales[0], scales[1], self._fused["costs"], scales[2], self._kt_dev, self._g_out,
                                   self.gamma, self.lb, n_glob)
                self._fused = None
            else:
                ops.began_step(l_adv_ny_G, l_adv_cl, l_CTC, self._kt_dev, self._g_out, self.gamma, self.lb, n_glob)
            return enhanced, prob
        # data parallel: the three loss scalars are all-reduced and kt advanced on the auxiliary stream - the main stream goes
        # straight on to the next step and only waits for this event where kt is read (the step prologue)
        main, aux = torch.cuda.current_stream(), self._aux_stream
        if self._fused:
            # raw sums -> [sum L1 noisy, sum L1 clean, sum CTC costs] (one launch) -> SUM over the ranks -> the controller with the
            # GLOBAL normalisers and batch size as device scalars (one launch): Proportional Control Theory (:190-194) on the device
            l1, costs = self._fused["l1"], self._fused["costs"]
            self._fused = None
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                out3 = torch.empty(3, device=l1.device, dtype=torch.float64)
                ops.loss_pack(l1, costs, out3)
                dp.reduce_scalars(out3)
                ops.began_step_sums(out3, out3[2:], 1.0, 1.0, 1.0, self._kt_dev, self._g_out, self.gamma, self.lb, 0.0,
                                    d_scales3=scales.all, d_n_batch=scales.cnt)
                self._kt_ev = torch.cuda.Event()
                self._kt_ev.record(aux)
            for t_ in (l1, costs, scales.all, scales.cnt):
                t_.record_stream(aux)
            return enhanced, prob
        packed = torch.stack([l_adv_ny_G.detach().reshape(()), l_adv_cl.detach().reshape(()), l_CTC.detach().reshape(())]).double()
        aux.wait_stream(main)
        with torch.cuda.stream(aux):
            dp.reduce_scalars(packed)   # every loss is already divided by its GLOBAL normaliser: the sum over ranks is the loss
            # Proportional Control Theory (:190-194) on the device
            bal = self.gamma * packed[1] - packed[0]
            self._kt_dev.copy_(torch.clamp(self._kt_dev + self.lb * bal, 0.0, 1.0))
            self._g_out[:3].copy_(packed)
            self._g_out[3:4].copy_(self._kt_dev)
            # running CTC average of the log line (ctc_tr_local.update(l_ctc, N) every iteration, :169-170) kept on the device
            self._g_out[4:5].add_(packed[2] * n_glob.count(0))
            self._g_out[5:6].add_(n_glob.count(0))
            self._kt_ev = torch.cuda.Event()
            self._kt_ev.record(aux)
        packed.record_stream(aux)
        return enhanced, prob

    def _batched_D_core(self, inputs, cl_inputs, scales, ctc_meta, asr_steps):
        """D(enhanced) and D(clean) as ONE batched pass of 2N rows beside the acoustic chain - the default schedule: for equal
        padded lengths, and for a ragged noisy / clean pair with two row classes of different length inside D's recurrent
        launches (`_ragged_batched_ok`; pinned to the reference by fixtures F1r / F3r)."""
        dp = self.dp
        N, dev = inputs.size(0), inputs.device
        enhanced = self.G(inputs)
        leaf = enhanced.detach().requires_grad_(True)
        overlap = self._overlap_asr()
        acoustic = None
        self._wait_kt()
        if getattr(self, "_fused", None) is not None:
            w = self._rs_pair.detach()        # written by the step prologue: [-kt] * N + [1] * N
        else:
            w = torch.empty(N + cl_inputs.size(0), device=dev, dtype=torch.float32)
            w[:N].copy_((-self._kt_dev).to(torch.float32).expand(N))
            w[N:] = 1.0
        ragged = cl_inputs.size(2) != inputs.size(2)
        if ragged:
            # noisy / clean batches of different padded length in ONE pass of max(T) frames: the recurrent launches treat the two
            # row classes as sequences of their own length (ops.RowWeights.row_len -> aasLaunch.cls_*), the losses read each class's
            # own frames
            assert self._ragged_batched_ok(inputs, cl_inputs)
        # the two utterance classes of the batched pass and their weights (device scalars), for the row-major weight-gradient GEMM
        rs = ops.RowWeights(w, classes=[(0, N, w[0:1]), (N, cl_inputs.size(0), None)],
                            row_len=(N, inputs.size(2), cl_inputs.size(2)) if ragged else None)
        if overlap:  # two chains of persistent launches side by side, half the chip each
            ops.set_rnn_cu_limit(ops.device_cus() // 2)
        if self._interleave_ok():
            # the discriminator's weight-gradient products are held back while the two BPTT chains run (they slow the chains'
            # cross-CU exchange) and released into E's backward phase, where half of the chip has little else to do
            self.launch.defer_wgrad = knobs.get("DEFER_WGRAD")   # (measured: no gain, 19.0 vs 19.2 ms - kept as a switch)
            # The weight-gradient products of the layers that are back-propagated FIRST (D's - and a trainable A's - top layers)
            # are held back until E's backward: beside the two BPTT chains every CU is taken and the products only slow the chains
            # down, beside E's backward half of the chip is free.  Not all of them: E's backward phase has room for about two D
            # layers on top of E's own (16.7 -> 16.1 ms with two, 16.2 with three, 16.5 with one).
            self.launch.defer_lids.clear()
            # (fp32 mode, same-box runs: 0 / 1 / 2 / 3 / 4 held-back layers = 30.6-31.0 / 30.1 / 29.9-30.4 / 30.2-30.3 / 30.1-30.2 ms
            #  with the eight-wave fp32 GEMM; with the four-wave one the weight-gradient stream was saturated and 0 was best)
            # (fp32-equivalent mode, with the six-product BPTT: 0 / 1 / 2 = 23.9-24.2 / 24.1-24.2 / 24.5-24.6 ms - its weight-gradient
            #  products are cheap enough to run beside the chains)
            # (a trainable A - the reference's default - round 6, same box, 30 steps each: 0 / 1 / 2 / 3 / 4 / all 5 of A's layers held back =
            #  30.7-30.9 / 30.7 / 30.8 / 30.8 / 30.7 / 30.3-30.5 ms: beside the two chains A's 306 GFLOP of weight-gradient products slow
            #  the chain they belong to; profiles/r06_trainableA_sweep.txt.  A frozen A has no such products: the setting is moot there.)
            for net, knob, dflt in ((self.D, "DEFER_D_LAYERS", 0 if ops._precision[0] == 2 else 2), (self.ASR, "DEFER_A_LAYERS", 99)):
                ndef = knobs.get(knob)
                ndef = dflt if ndef is None else int(ndef)
                if ndef > 0:
                    lids = [m._aas_layer_id for m in net.modules() if getattr(m, "_aas_layer_id", None) is not None]
                    self.launch.defer_lids.update(lids[-ndef:])
            try:
                l_adv_ny_G, l_adv_cl, prob, l_CTC, leaf_a = self._interleaved_DA(enhanced, leaf, cl_inputs, rs, None, ctc_meta, None, None, None,
                                                                                scales=scales)
            finally:
                self.launch.defer_wgrad = False
                self.launch.defer_lids.clear()
        else:
            if overlap:
                acoustic = self._acoustic_branch(enhanced, None, None, None, None, ctc_meta, scale=scales[2])
            ae = self.D(torch.cat([leaf, cl_inputs], 0), wgrad_row_scale=rs)
            l_adv_ny_G = ops.l1_sum(ae[:N], leaf) * scales[0]
            l_adv_cl = ops.l1_sum(ae[N:], cl_inputs) * scales[1]
            (l_adv_ny_G + l_adv_cl).backward()
            if acoustic is None:
                acoustic = self._acoustic_branch(enhanced, None, None, None, None, ctc_meta, scale=scales[2])
            prob, l_CTC, leaf_a = acoustic
        torch.cuda.current_stream().wait_stream(self._side)
        # E's BPTT launches are capped too: the CUs they leave free run E's weight-gradient GEMMs (side stream), which would
        # otherwise wait for each fully-resident 512-thread launch to retire
        ops.set_rnn_cu_limit(knobs.get("EBWD_CUS"))
        leaf_a.grad.record_stream(torch.cuda.current_stream())
        # the gradients arriving at `enhanced`: through D's input, as the L1 target (:146-147: `enhanced` is both), through A
        tg = (self._fused or {}).get("tgrad") or [None]
        gsum = ops.add3(leaf.grad, leaf_a.grad, tg[0])
        if getattr(self, "keep_enh_grads", False):   # parity gates / tests: the two gradients arriving at `enhanced` (:148, :170)
            self._enh_grads = (ops.add3(leaf.grad, tg[0]) if tg[0] is not None else leaf.grad, leaf_a.grad)
        ops.flush_deferred_wgrad()
        if dp.active:   # D's (and a trainable A's) small parameters; their layer buckets are in flight: overlaps E's backward
            self._reducer.flush(self._flat["D"])
            if asr_steps:
                self._reducer.flush(self._flat["A"])
        elif knobs.get("EARLY_ADAM") and ops.DIRECT_WGRAD[0] and ops.LINEAR_DIRECT[0]:
            # D's (and a trainable A's) gradients are complete once the products queued on the weight-gradient stream have run:
            # their Adam steps and weight-plane refreshes go onto that stream now and overlap E's backward instead of
            # following it (every parameter gradient of D is produced on that stream; A's BatchNorm / conv / fc ones on `side`)
            wg = ops.wgrad_stream(leaf.device)
            wg.wait_stream(torch.cuda.current_stream())   # (cheap insurance: any gradient autograd accumulated on this stream)
            if asr_steps:
                wg.wait_stream(self._side)
            with torch.cuda.stream(wg):
                self._opts[2].step_dev()
                if asr_steps:
                    self._opts[1].step_dev()
            self._early_adam = True
        enhanced.backward(gsum)
        return enhanced, prob, l_adv_ny_G, l_adv_cl, l_CTC

    def _ragged_batched_ok(self, inputs, cl_inputs):
        """A noisy / clean pair that differs only in its padded length T can take the batched-D schedule (knobs.RAGGED_BATCHED): D's
        recurrent launches carry two row classes.  Needs the interleaved device path and lstm / gru layers in D.  The batched pass
        runs max(T) steps for every row, the two-lane schedule each batch's own: config 2 with noisy T = 200 measured 27.0-27.3 ms at
        every clean T, against 28.8 / 27.9 / 27.5 / 27.3 / 26.9 ms on two lanes at clean T = 184 / 160 / 150 / 130 / 120
        (tools/ragged_bench.py), hence the length-ratio threshold."""
        if not knobs.get("RAGGED_BATCHED") or str(knobs.get("TWO_LANES")) != "auto" or not self._interleave_ok():
            return False
        a, b = tuple(inputs.shape), tuple(cl_inputs.shape)
        if len(a) != 3 or len(b) != 3 or a[1] != b[1] or a[2] == b[2]:
            return False
        if min(a[2], b[2]) < float(knobs.get("RAGGED_MIN_RATIO")) * max(a[2], b[2]):
            return False
        return all(m.kind in ("lstm", "gru") for m in self.D.modules() if getattr(m, "_aas_layer_id", None) is not None)

    def _lanes_ok(self, same_shape=True):
        """AAS_TWO_LANES = auto (default): the two-lane schedule when the noisy and clean batches have different padded
        lengths (real loaders), the batched-D schedule when they are equal (0.1-0.3 ms / step faster at config 2); 1 / 0 force."""
        mode = str(knobs.get("TWO_LANES"))
        return self._interleave_ok() and (mode == "1" or (mode == "auto" and not same_shape))

    def _two_lane_core(self, inputs, cl_inputs, scales, ctc_meta, asr_steps):
        """The step as two lanes of half-chip persistent launches that are busy from the first kernel to the last:

            main stream:  E forward      -> D(enhanced) forward -> D(enhanced) backward -> E backward
            side stream:  D(clean) fwd   -> A forward + CTC     -> A backward           -> D(clean) backward

        D(clean) needs nothing from E, so its forward fills the half of the chip that idles during E's forward and its
        backward the half that idles during E's backward; D then runs N rows per launch instead of 2N (a persistent launch
        on a fixed CU budget slows down with the rows per workgroup: BPTT 3.7 -> 6.2 us / step from N=30 to N=60), and
        the noisy / clean batches may have different padded lengths.  The D-step parameter gradients of the enhanced pass
        are (-kt) x its G-step ones (per-utterance weights on the weight-gradient products, as before); both passes
        accumulate into D's flat gradient buffer on the weight-gradient stream.  Every pair of neighbouring chains is
        QUEUED layer by layer in alternation (forward: generators; backward: one autograd call per pair, whose nodes pop
        in reverse creation order), because the device only overlaps what the host has already queued."""
        c, dp = self.config, self.dp
        N, dev = inputs.size(0), inputs.device
        main = torch.cuda.current_stream()
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = ops.chain_stream()
        side = self._side = self._side_stream
        side.wait_stream(main)
        ops.set_rnn_cu_limit(knobs.get("LANE_CUS") or ops.device_cus() // 2)
        if dp.active:
            self._reducer.expect(self._flat["D"], 2)      # every D layer is back-propagated twice: reduce after the second
        self._wait_kt()
        if getattr(self, "_rs_lane", None) is not None and self._rs_lane.numel() == N and not dp.active:
            w = self._rs_lane.detach()          # written by the step prologue
        else:
            w = (-self._kt_dev).to(torch.float32).expand(N).contiguous()
        rs = ops.RowWeights(w, classes=[(0, N, w[0:1])])

        def alternate(gen_main, gen_side):
            a = b = None
            while a is None or b is None:
                if a is None:
                    a = next(gen_main)
                if b is None:
                    with torch.cuda.stream(side):
                        b = next(gen_side)
            return a, b
        # ---- E forward beside D(clean) forward
        enhanced, ae_cl = alternate(self.G.forward_stages(inputs), self.D.forward_stages(cl_inputs))
        with torch.cuda.stream(side):
            l_adv_cl = ops.l1_sum(ae_cl, cl_inputs) * scales[1]
        # ---- D(enhanced) forward beside A forward (+ CTC)
        leaf = enhanced.detach().requires_grad_(True)
        leaf_a = enhanced.detach().requires_grad_(True)
        side.wait_stream(main)
        enhanced.record_stream(side)
        ae_ny, out_a = alternate(self.D.forward_stages(leaf, wgrad_row_scale=rs), self.ASR.forward_stages(leaf_a))
        l_adv_ny_G = ops.l1_sum(ae_ny, leaf) * scales[0]
        with torch.cuda.stream(side):
            prob = out_a.transpose(0, 1)
            l_CTC = ops.ctc_sum(prob, None, None, None, self.CTCLoss.blank, ctc_meta) * scales[2]
        # ---- their backward passes, alternating (nodes pop in reverse creation order, each on its forward's stream)
        self._backward_pair(l_adv_ny_G, l_CTC, main, side)
        if dp.active and asr_steps:
            with torch.cuda.stream(side):
                self._reducer.flush(self._flat["A"])
        # ---- E backward beside D(clean) backward
        main.wait_stream(side)          # A's gradient wrt enhanced (everything queued on the side stream so far)
        leaf_a.grad.record_stream(main)
        gsum = ops.add3(leaf.grad, leaf_a.grad)
        if getattr(self, "keep_enh_grads", False):
            self._enh_grads = (leaf.grad, leaf_a.grad)
        torch.autograd.backward([enhanced, l_adv_cl], [gsum, None])
        main.wait_stream(side)
        if dp.active:
            self._reducer.flush(self._flat["D"])
        return enhanced, prob, l_adv_ny_G, l_adv_cl, l_CTC

    def _wait_kt(self):
        """The device-resident kt (and the scalars of the last step) may still be in flight on the auxiliary stream (data parallel)."""
        ev = getattr(self, "_kt_ev", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _ensure_dev_state(self, dev):
        if getattr(self, "_kt_dev", None) is None:
            self._kt_dev = torch.zeros(1, device=dev, dtype=torch.float64)
            self._g_out = torch.zeros(6, device=dev, dtype=torch.float64)
            self._kt_dev.fill_(float(self.kt))
            self._kt_dev_live = False   # the host copy is the current one until a device-resident step has run

    @ops.with_trainer_precision
    def train_step_async(self, data_list, data_list_cl, iter):
        """The fused iteration queued WITHOUT any host synchronisation: kt, the Adam bias corrections and the loss scalars
        stay on the device (`_device_core`), the CTC metadata goes up
        from pinned memory, and nothing is read back - so the host queues step i+1 while the GPU is still running step i
        and the ~10 ms of Python launch overhead per step never leaves a stream dry at a step boundary.  Returns device
        tensors; `read_scalars()` (one D2H copy) updates `self.kt` and returns the last step's losses - call it when a log
        line needs them (the reference logs every `log_iter` iterations, trainer_AAS.py:196-215).  Works data parallel
        (global normalisers, bucketed gradient all-reduce, all-reduced kt inputs: all on the device), with a trainable A, and
        on noisy / clean pairs of different padded length (batched pass with two row classes, or the two-lane schedule below a
        length ratio of knobs.RAGGED_MIN_RATIO).  Falls back to train_step for the as-executed schedule and for masks without
        a host-side frame count."""
        if self._opts is None:
            self.make_optimizers()
        if self.schedule != "fused":
            return self.train_step(data_list, data_list_cl, iter, log_norms=False)
        inputs, targets, input_percentages, target_sizes, mask = self._prep(data_list)
        cl_inputs, cl_mask = data_list_cl[0], data_list_cl[4]
        if not cl_mask.is_cuda:
            attach_n_valid(cl_mask)
        nv = lambda m: getattr(m, "n_valid", None)
        nv_ny, nv_cl = nv(mask), nv(cl_mask)
        if nv_ny is None or nv_cl is None or (tuple(cl_inputs.shape) != tuple(inputs.shape) and not self._lanes_ok(False)
                                              and not self._ragged_batched_ok(inputs, cl_inputs)):
            return self.train_step(data_list, data_list_cl, iter, log_norms=False)
        dev = next(self.G.parameters()).device
        cl_inputs = _get_variable_nograd(cl_inputs)
        t_out = self.ASR.output_length(inputs.size(2))
        sizes = input_percentages.clone().mul_(int(t_out)).int()
        meta = ops.ctc_prepare(targets, sizes, target_sizes, "cpu")
        meta = dict(meta, meta=self._upload_small(meta["meta"], dev))
        self._ensure_dev_state(dev)
        if not getattr(self, "_kt_dev_live", False):
            self._kt_dev.fill_(float(self.kt))   # another path advanced the host copy since
        self._kt_dev_live = True
        self._host_ahead = True
        enhanced, prob = self._device_core(inputs, cl_inputs, nv_ny, nv_cl, meta, it=iter)
        self._host_ahead = False
        return dict(enhanced=enhanced, prob=prob, scalars=self._g_out)

    def read_scalars(self):
        """One D2H copy of (l_adv_ny_G, l_adv_cl, l_ctc, kt) of the last train_step_async; updates the host-side kt and the
        running CTC average, and (it is a synchronisation point) raises if a persistent kernel timed out or the run diverged."""
        self._wait_kt()
        l_adv_ny_G, l_adv_cl, l_ctc, kt, ctc_sum, ctc_n = self._g_out.tolist()
        self._g_out[4:6].zero_()
        ops.check_rnn_health((l_adv_ny_G, l_adv_cl, l_ctc))
        self.kt = kt
        self._kt_dev_live = False
        if ctc_n > 0:
            self.ctc_tr_local.update(ctc_sum / ctc_n, ctc_n)
        bal = self.gamma * l_adv_cl - l_adv_ny_G
        return dict(l_adv_ny_G=l_adv_ny_G, l_adv_cl=l_adv_cl, l_ctc=l_ctc, kt=kt, conv_measure=l_adv_cl + abs(bal))

    def _interleaved_DA(self, enhanced, leaf, cl_inputs, rs, N_glob, ctc_meta, targets, sizes, target_sizes, scales=None,
                        mask=None, cl_mask=None):
        """Discriminator pass and acoustic pass QUEUED layer by layer in alternation on two streams, one combined backward.

        The two chains are independent, but the device only overlaps what the host has queued: queueing one whole chain
        (forward and backward, ~150 launches) before the other - in Python or as consecutive hipGraph nodes, which are
        enqueued in creation order at ~30 us apiece - delays the second chain by 5-6 ms (rocprofv3 timeline).  So both
        forwards are advanced one layer at a time (`forward_stages`), and ONE `torch.autograd.backward` call over both
        losses lets the engine pop backward nodes in reverse creation order, i.e. alternating between the chains, each
        on the stream its forward ran on.  Returns (l_adv_ny_G, l_adv_cl, prob, l_CTC, leaf_a)."""
        c = self.config
        caller = torch.cuda.current_stream()
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = ops.chain_stream()
        side = self._side = self._side_stream
        side.wait_stream(caller)
        # CU-masked lanes (ops.lane_stream): the discriminator chain on the low CU half, the acoustic chain (side) on the high half
        main = ops.lane_stream(0, leaf.device) if ops.CHAIN_LANES[0] else caller
        if main is not caller:
            main.wait_stream(caller)
        Nn = leaf.size(0)
        leaf_a = enhanced.detach().requires_grad_(True)
        enhanced.record_stream(side)
        with torch.cuda.stream(main):
            if main is not caller:
                for t_ in (enhanced, leaf, cl_inputs, rs.w):
                    t_.record_stream(main)
            fused = getattr(self, "_fused", None) is not None and scales is not None and mask is None
            Tn, Tc = leaf.size(2), cl_inputs.size(2)
            # (different padded lengths: only with the row classes the caller put into rs - _batched_D_core)
            assert Tn == Tc or rs.row_len == (leaf.size(0), Tn, Tc)
            if fused or Tn != Tc:
                gD = self.D.forward_stages(None, wgrad_row_scale=rs, pair=(leaf, cl_inputs))
            else:
                gD = self.D.forward_stages(torch.cat([leaf, cl_inputs], 0), wgrad_row_scale=rs)
        gA = self.ASR.forward_stages(leaf_a)
        ae = out_a = None
        while ae is None or out_a is None:
            if ae is None:
                with torch.cuda.stream(main):
                    ae = next(gD)
            if out_a is None:
                with torch.cuda.stream(side):
                    out_a = next(gA)
        if fused:
            # raw roots: the L1 sums [2] and the CTC costs [N]; their weights ride in the backward launches (ops.l1_pair / ctc_scaled)
            with torch.cuda.stream(main):
                self._fused["tgrad"] = []
                l_pair = ops.l1_pair(ae, leaf, cl_inputs, scales[0], scales[1], self._l1_acc.detach(), self._fused["tgrad"])
            with torch.cuda.stream(side):
                prob = out_a.transpose(0, 1)
                l_CTC = ops.ctc_scaled(prob, self.CTCLoss.blank, ctc_meta, scales[2])
            self._fused.update(l1=l_pair.detach(), costs=l_CTC.detach())
            with torch.cuda.stream(main):
                self._backward_pair(l_pair, l_CTC, main, side)
            if main is not caller:
                caller.wait_stream(main)
                for t_ in [leaf.grad, ae] + list(self._fused["tgrad"]):     # (the target gradient is read by add3 on the caller's stream too)
                    if t_ is not None:
                        t_.record_stream(caller)
            return None, None, prob, None, leaf_a
        with torch.cuda.stream(main):
            if mask is not None:   # eager path: masked-L1 modules (host-side n_valid)
                l_adv_ny_G, _ = self.diffLoss(ae[:Nn], leaf, mask)
                l_adv_cl, _ = self.diffLoss(ae[Nn:], cl_inputs, cl_mask)
                l_adv_ny_G = l_adv_ny_G * c.w_adversarial
                l_adv_cl = c.w_adversarial * l_adv_cl
            else:   # device path: (weight / normaliser) per loss as python floats or device scalars (data parallel)
                l_adv_ny_G = ops.l1_sum(ae[:Nn] if Tn >= Tc else ae[:Nn, :, :Tn], leaf) * scales[0]
                l_adv_cl = ops.l1_sum(ae[Nn:] if Tc >= Tn else ae[Nn:, :, :Tc], cl_inputs) * scales[1]
            l_pair = l_adv_ny_G + l_adv_cl
        with torch.cuda.stream(side):
            prob = out_a.transpose(0, 1)
            if targets is None:
                l_CTC = ops.ctc_sum(prob, None, None, None, self.CTCLoss.blank, ctc_meta) * scales[2]
            else:
                l_CTC = c.w_acoustic * self.CTCLoss(prob, targets, sizes, target_sizes, prepared=ctc_meta) / N_glob
        with torch.cuda.stream(main):
            self._backward_pair(l_pair, l_CTC, main, side)
        if main is not caller:
            caller.wait_stream(main)
            for t_ in (leaf.grad, l_adv_ny_G, l_adv_cl, ae):
                if t_ is not None:
                    t_.record_stream(caller)
        return l_adv_ny_G, l_adv_cl, prob, l_CTC, leaf_a

    def _backward_pair(self, loss_main, loss_side, main, side):
        """ONE autograd call over two losses whose graphs live on two streams, without tying the streams together.
        The engine makes every root's consumer wait for the stream that is current when backward() is called (it assumes the
        root gradients were produced there): called from the main stream, the side chain's backward would wait for everything
        queued on main - i.e. the acoustic backward for the tail of D's forward.  So the root gradients are created on the
        streams of their losses and the call is issued from a stream that has nothing queued."""
        mode = str(knobs.get("PAIR_BWD"))
        if mode == "0" or (mode == "auto" and (getattr(self, "_host_ahead", False) or self.dp.active)):
            # Two independent backward calls, each issued from (and confined to) the stream of its chain: nothing ties the
            # chains together.  Right whenever the host queues ahead of the device (train_step_async), where the order in
            # which the host queues the two chains does not matter; and data parallel, where the utility stream that the
            # paired call is issued from also carries the collectives (the acoustic backward then waited for the end of D's
            # forward: 19.3 vs 18.4 ms / step with one-rank RCCL).
            unit = (lambda l: ops.unit_root(l)) if getattr(self, "_fused", None) is not None else (lambda l: None)
            with torch.cuda.stream(side):
                torch.autograd.backward([loss_side], [unit(loss_side)])
            torch.autograd.backward([loss_main], [unit(loss_main)])
            return
        if not knobs.get("NEUTRAL_BWD"):
            side.wait_stream(main)
            torch.autograd.backward([loss_main, loss_side])
            return
        if getattr(self, "_neutral_stream", None) is None:
            self._neutral_stream = ops.refresh_stream(torch.cuda.current_device())   # (one utility stream: the device has few hardware queues)
        g_main = torch.ones_like(loss_main)
        with torch.cuda.stream(side):
            g_side = torch.ones_like(loss_side)
        caller = side if knobs.get("BWD_FROM") == "side" else self._neutral_stream
        with torch.cuda.stream(caller):
            torch.autograd.backward([loss_main, loss_side], [g_main, g_side])

    def _interleave_ok(self):
        c = self.config
        exact_a = getattr(c, "asr_exact_fp32", None)
        if exact_a is None:
            exact_a = knobs.get("ASR_EXACT")
        return self._overlap_asr() and not exact_a and knobs.get("INTERLEAVE")

    @staticmethod
    def _overlap_asr():
        return knobs.get("OVERLAP_ASR")

    def _acoustic_branch(self, enhanced, targets, sizes, target_sizes, N_glob, ctc_meta, scale=None):
        """A(enhanced) -> CTC/N -> backward down to a private leaf (optionally on a second stream)."""
        c = self.config
        # Default (AAS_OVERLAP_ASR=1): the acoustic branch runs on a second stream, queued BEFORE the discriminator pass, and
        # every persistent recurrent launch of the two chains is capped at half the CUs (ops.set_rnn_cu_limit), so a launch of
        # each chain is always fully resident next to one of the other.  The recurrent kernels are latency- not throughput-
        # bound, so two chains side by side finish sooner than one after the other (27.2 -> 24.0 ms / step); without the cap
        # (each launch sized for the whole chip) the overlap measured nothing.  AAS_OVERLAP_ASR=0: one chain on the main stream.
        if self._overlap_asr():
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = ops.chain_stream()
            self._side = self._side_stream
        else:
            self._side = torch.cuda.current_stream()
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        exact_a = getattr(c, "asr_exact_fp32", None)
        if exact_a is None:
            exact_a = knobs.get("ASR_EXACT")
        prev = ops.get_precision()
        with torch.cuda.stream(self._side):
            if exact_a:
                ops.set_precision(0)
            try:
                leaf_a = enhanced.detach().requires_grad_(True)
                enhanced.record_stream(self._side)
                prob = self.ASR(leaf_a).transpose(0, 1)
                if targets is None:  # device path: labels live only in the pre-uploaded device metadata
                    l_CTC = ops.ctc_sum(prob, None, None, None, self.CTCLoss.blank, ctc_meta) * scale
                else:
                    l_CTC = c.w_acoustic * self.CTCLoss(prob, targets, sizes, target_sizes, prepared=ctc_meta) / N_glob
                l_CTC.backward()
            finally:
                ops.set_precision(prev)
        return prob, l_CTC, leaf_a

    def _next_pair(self):
        """Next (noisy, clean) training batches; data parallel: the loader shards BEFORE loading (DataLoader(dp=...): every rank
        walks the same global bins with the same seed and opens only its strided share of the length-sorted utterances)."""
        data_list = self.data_loader.next(cl_ny="ny", type="train")
        data_list_cl = self.data_loader.next(cl_ny="cl", type="train")
        if self.dp.active and getattr(self.data_loader, "dp", None) is None:
            # a loader that is not data-parallel aware hands out the global batch: shard it here (every rank loaded all of it)
            data_list, data_list_cl = self.dp.shard_collated(data_list), self.dp.shard_collated(data_list_cl)
        return data_list, data_list_cl

    def train(self):
        """:125-297.  Iterations that print nothing are queued with train_step_async (no host read-back); logging iterations
        run train_step, which also returns the two gradient norms of the log line."""
        from tqdm import trange
        c = self.config
        self.make_optimizers()
        rank0 = self.dp.rank == 0
        for iter in trange(c.start_iter, c.max_iter, disable=not rank0):
            logging = (iter + 1) % c.log_iter == 0
            data_list, data_list_cl = self._next_pair()
            if not logging:
                self.train_step_async(data_list, data_list_cl, iter)
            else:
                r = self.train_step(data_list, data_list_cl, iter, log_norms=True)
                lines = [
                    "[{}/{}] (train) CTC: {:.7f}, ADV_cl: {:.7f}, ADV_ny: {:.7f}".format(iter, c.max_iter, self.ctc_tr_local.avg, r["l_adv_cl"], r["l_adv_ny_G"]),
                    "[{}/{}] (train) conv_measure: {:.4f}, kt: {:.4f} ".format(iter, c.max_iter, r["conv_measure"], self.kt),
                    "[{}/{}] (train) gradient norm, adv: {:.4f}, adv + ctc : {:.4f}".format(iter, c.max_iter, float(r["g_adv"]), float(r["g_ctc_adv"])),
                ]
                for s in lines:
                    if rank0:
                        print(s)
                    if self.logFile:
                        self.logFile.write(s + "\n")
                if self.logFile:
                    self.logFile.flush()
                self.ctc_tr_local.reset()
            if (iter + 1) % c.save_iter == 0:
                if getattr(self, "_kt_dev_live", False):
                    self.read_scalars()
                # rank 0 validates and writes the checkpoints; A stays in train mode during validation (as in the reference), so
                # its BatchNorm running statistics move on rank 0 only: every rank then takes rank 0's (ValidationMixin)
                self._save_iter_block(iter)

    # ---- validation + checkpoint lifecycle (:215-297) -----------------------------------------
    @ops.with_trainer_precision
    def validate_and_checkpoint(self, iter):
        c = self.config
        self.G.eval()
        for (name, dl), (ctc_m, adv_m, wer_m, cer_m) in zip(self._validation_sets(), ((self.ctc_tr, self.adv_ny_tr, self.wer_tr, self.cer_tr),
                                                                                      (self.ctc_val, self.adv_ny_val, self.wer_val, self.cer_val))):
            for m in (ctc_m, adv_m, wer_m, cer_m):
                m.reset()
            for _ in range(self.data_loader.num_batches(dl)):
                d = self.data_loader.next(cl_ny="ny", type=dl)
                with torch.no_grad():
                    ctc, adv_ny, nElement, wer, cer, nWord, nChar = self.greedy_decoding_and_AAS(d[0], d[1], d[2], d[3], d[4])
                ctc_m.update(float(ctc), d[0].size(0)); adv_m.update(float(adv_ny), nElement)
                wer_m.update(wer, nWord); cer_m.update(cer, nChar)
            self._log("[{}/{}] ({}) CTC: {:.7f}, WER: {:.7f}, CER: {:.7f}".format(iter, c.max_iter, name, ctc_m.avg, wer_m.avg * 100, cer_m.avg * 100), flush=True)
        self.G.train()
        self._save_rotating("G", self.G, iter)
        self._save_rotating("ASR", self.ASR, iter)
        self._keep_if_best(iter, self.wer_val.avg, ("G", "ASR"))

    @ops.with_trainer_precision
    def greedy_decoding_and_AAS(self, inputs, targets, input_percentages, target_sizes, mask, transcript_prob=0.001):
        """:301-351 (keeps the reference's ``cer = ce/total_word`` quirk at :338)."""
        inputs = _get_variable_volatile(inputs)
        attach_n_valid(mask) if not mask.is_cuda else None
        mask = _get_variable_volatile(mask)
        N = inputs.size(0)
        enhanced = self.G(inputs)
        prob, sizes, wer, cer, total_word, total_char = self._greedy_pass(enhanced, targets, input_percentages, target_sizes, transcript_prob)
        ae_ny = self.D(enhanced)
        l_adv_ny, nElement = self.diffLoss(ae_ny, enhanced, mask)
        l_adv_ny = l_adv_ny * self.config.w_adversarial
        l_CTC = self.config.w_acoustic * self.CTCLoss(prob, targets, sizes, target_sizes) / N
        return l_CTC, l_adv_ny, nElement, wer, cer, total_word, total_char
