"""FSEGAN trainer (reference Speech_enhancement_by_AAS/trainer_FSEGAN.py; hot loop :128-182).

The reference file is unrunnable as written.  This builds the INTENDED step:
nFeat_in = nFeat_out = nFeat, D = stackedBRNN(I=2*nFeat, O=nFeat) fed through ``forward_paired``
everywhere (the reference's ``self.D(cleans, mixture)`` at :167 has the wrong arity), and the DCE
term IS back-propagated (G loss = dce + w_adversarial * adv).  ``config.fsegan_as_written=True``
keeps the reference's behaviour of only logging the DCE term (:161-163).
"""
import os

import torch

from . import ops
from .model import L1Loss_mask, stackedBRNN, supported_rnns
from .utils import AverageMeter, _get_variable_nograd, attach_n_valid


class Trainer(ops.TrainerContext):
    def __init__(self, config, data_loader=None, models=None):
        self._init_context()   # arithmetic mode + launch settings this trainer runs in (ops.TrainerContext)
        self.config, self.data_loader = config, data_loader
        self.lr, self.beta1, self.beta2 = config.lr, config.beta1, config.beta2
        self.diffLoss = L1Loss_mask()
        self.model_dir = "logs/" + str(config.expnum)
        self.kt, self.lb, self.gamma = 0, config.lambda_k, config.gamma
        self.dce_tr_local = AverageMeter()
        self.as_written = getattr(config, "fsegan_as_written", False)
        if models is not None:
            self.G, self.D = models
        else:
            self.build_model()
        if config.gpu >= 0:
            self.G.cuda(); self.D.cuda()
        self.logFile = None
        if config.mode == "train" and getattr(config, "write_log", True) and int(os.environ.get("RANK", "0")) == 0:
            os.makedirs(self.model_dir, exist_ok=True)
            self.logFile = open(self.model_dir + "/log.txt", "w")
        self._opts = None
        self._flat = None
        self.dp = None

    def zero_grad_all(self):
        self.G.zero_grad(); self.D.zero_grad()

    def build_model(self):
        c = self.config
        rt = supported_rnns[c.rnn_type]
        self.G = stackedBRNN(I=c.nFeat, O=c.nFeat, H=c.rnn_size, L=c.rnn_layers, rnn_type=rt)
        self.D = stackedBRNN(I=2 * c.nFeat, O=c.nFeat, H=c.rnn_size, L=c.rnn_layers, rnn_type=rt)

    def make_optimizers(self):
        """Adam(amsgrad) per network on flat parameter / gradient buffers (one fused launch each); the recurrent layers'
        weight gradients accumulate into them on the side stream (ops.sync_wgrad joins before they are read)."""
        from .dist import BucketReducer, DPContext, FlatBuffers
        from .optim import FlatAdam
        c = self.config
        self.dp = getattr(self, "dp", None) or DPContext.from_env()
        for name, m in (("G", self.G), ("D", self.D)):
            ops.name_layers(m, name)
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

    def _forward_backward(self, mixture, cleans, rs, s_adv, s_dce):
        """The fused pass shared by both step forms.  D(enhanced | mixture) and D(clean | mixture) share ONE batched pass of 2N
        rows, the D-step parameter gradients of the enhanced half are (-kt) x its G-step ones (per-utterance weights `rs` on the
        weight-gradient products only), and E is back-propagated once with d(adv)/d(enhanced) + d(dce)/d(enhanced).
        s_adv = w_adversarial / nElement, s_dce = 1 / nElement (python floats, or device scalars when data parallel)."""
        N = mixture.size(0)
        enhanced = self.G(mixture)
        leaf = enhanced.detach().requires_grad_(True)
        paired = torch.cat([torch.cat([leaf, mixture], 1), torch.cat([cleans, mixture], 1)], 0)   # forward_paired x 2 (model.py:233-238)
        ae = self.D(paired, wgrad_row_scale=rs)
        l_adv_ny_G = ops.l1_sum(ae[:N], leaf) * s_adv
        l_adv_cl = ops.l1_sum(ae[N:], cleans) * s_adv
        dce = ops.l1_sum(leaf, cleans) * s_dce
        total = l_adv_ny_G + l_adv_cl
        if not self.as_written:   # the reference only logs the DCE term (:161-163); the intended G loss back-propagates it
            total = total + dce
        total.backward()
        if self._reducer is not None:   # D's small parameters; its layer buckets are in flight: overlaps E's backward
            self._reducer.flush(self._flat["D"])
        enhanced.backward(leaf.grad)
        return enhanced, l_adv_ny_G, l_adv_cl, dce

    @ops.with_trainer_precision
    def train_step(self, data_list, iter=0):
        """:128-182 (intended semantics), host-synchronous: returns the scalars of the log lines (and the gradient
        norm of G).  Data parallel: `data_list` is this rank's shard; the losses are normalised by the GLOBAL nElement, so the
        SUM all-reduce of the flat gradient buffers gives the single-process gradients."""
        c = self.config
        if self._opts is None:
            self.make_optimizers()
        optimizer_g, optimizer_d = self._opts
        dp = self.dp
        if getattr(self, "_kt_dev_live", False):
            self.read_scalars()
        ops.sync_wgrad()
        for f in self._flat.values():
            f.zero_grad()
        mixture, cleans, mask = self._batch(data_list)
        N = mixture.size(0)
        nElement = getattr(mask, "n_valid", None)
        if nElement is None:
            nElement = int(mask.numel()) - int(mask.sum().item())
        if dp.active:
            (nElement,) = dp.global_counts([nElement])
        rs = torch.empty(2 * N, device=mixture.device, dtype=torch.float32)
        rs[:N] = -float(self.kt)
        rs[N:] = 1.0
        rs._aas_classes = [(0, N, rs[0:1]), (N, N, None)]   # the two utterance classes and their weights (ops.gemm_planes_tn)
        if self._reducer is not None:
            self._reducer.begin()
        enhanced, l_adv_ny_G, l_adv_cl, dce = self._forward_backward(mixture, cleans, rs, c.w_adversarial / nElement, 1.0 / nElement)
        ops.sync_wgrad()
        if dp.active:
            self._reducer.flush(self._flat["G"])
            self._reducer.wait()
        g_norm = self.get_gradient_norm(self.G)
        optimizer_g.step(); optimizer_d.step()
        ops.refresh_weight_planes(self.G); ops.refresh_weight_planes(self.D)
        packed = torch.stack([l_adv_ny_G.detach().reshape(()), l_adv_cl.detach().reshape(()), dce.detach().reshape(())])
        l_adv_ny_G_data, l_adv_cl_data, dce_loss = dp.reduce_scalars(packed).tolist()
        g_norm = float(g_norm)
        ops.check_rnn_health((l_adv_ny_G_data, l_adv_cl_data, dce_loss))
        self.dce_tr_local.update(dce_loss, nElement)
        g_d_balance = self.gamma * l_adv_cl_data - l_adv_ny_G_data
        self.kt += self.lb * g_d_balance
        self.kt = max(min(1, self.kt), 0)
        return dict(l_adv_ny_G=l_adv_ny_G_data, l_adv_cl=l_adv_cl_data, dce=dce_loss, kt=self.kt,
                    conv_measure=l_adv_cl_data + abs(g_d_balance), g_norm=g_norm, enhanced=enhanced)

    # ---- the same step without a host synchronisation (what train() queues on iterations that print nothing) -----------
    @ops.with_trainer_precision
    def train_step_async(self, data_list, iter=0):
        """train_step queued WITHOUT reading anything back: kt, the Adam bias corrections, the loss scalars and the running DCE
        average of the log line stay on the device (`aas_began_step`, `FlatAdam.step_dev`), so the host queues step i+1 while
        the GPU runs step i.  `read_scalars()` (one D2H copy) returns the last step's losses and updates the host-side kt.
        Data parallel: the global nElement is all-reduced on the utility stream and kept on the device, the gradient buffers
        are all-reduced bucket by bucket behind the weight-gradient products (dist.BucketReducer), and kt is advanced from the
        all-reduced loss scalars - still no host synchronisation."""
        c = self.config
        if self._opts is None:
            self.make_optimizers()
        optimizer_g, optimizer_d = self._opts
        dp = self.dp
        mixture, cleans, mask = self._batch(data_list)
        nElement = getattr(mask, "n_valid", None)
        if nElement is None:     # a device mask without a host-side count: the synchronous form counts it
            return self.train_step(data_list, iter)
        N, dev = mixture.size(0), mixture.device
        if getattr(self, "_kt_dev", None) is None:
            self._kt_dev = torch.zeros(1, device=dev, dtype=torch.float64)
            self._g_out = torch.zeros(6, device=dev, dtype=torch.float64)
            self._kt_ev = None
        if not getattr(self, "_kt_dev_live", False):
            self._kt_dev.fill_(float(self.kt))
        self._kt_dev_live = True
        ops.sync_wgrad()
        for f in self._flat.values():
            f.flat_g.zero_()
        aux = ops.refresh_stream(dev)
        if dp.active:
            from .dist import DeviceCounts
            cnt = DeviceCounts(dp, [nElement], dev, aux)
            n_glob = cnt.get(0)
            s_adv, s_dce = (c.w_adversarial / n_glob).float(), (1.0 / n_glob).float()
            self._reducer.begin()
            ops.WGRAD_HOOK[0] = self._reducer.on_wgrad
        else:
            n_glob = float(nElement)
            s_adv, s_dce = c.w_adversarial / nElement, 1.0 / nElement
        if self._kt_ev is not None:
            torch.cuda.current_stream().wait_event(self._kt_ev)
        rs = torch.empty(2 * N, device=dev, dtype=torch.float32)
        rs[:N].copy_((-self._kt_dev).to(torch.float32).expand(N))
        rs[N:] = 1.0
        rs._aas_classes = [(0, N, rs[0:1]), (N, N, None)]
        try:
            enhanced, l_adv_ny_G, l_adv_cl, dce = self._forward_backward(mixture, cleans, rs, s_adv, s_dce)
            ops.sync_wgrad()
            if dp.active:
                self._reducer.flush(self._flat["G"])
                self._reducer.wait()
        finally:
            ops.WGRAD_HOOK[0] = None
        optimizer_g.step_dev(); optimizer_d.step_dev()
        ops.refresh_weight_planes(self.G); ops.refresh_weight_planes(self.D)
        if not dp.active:   # controller (:175-179) + the log scalars in one tiny launch; slot 2 carries the DCE term
            ops.began_step(l_adv_ny_G, l_adv_cl, dce, self._kt_dev, self._g_out, self.gamma, self.lb, n_glob)
        else:
            main = torch.cuda.current_stream()
            packed = torch.stack([l_adv_ny_G.detach().reshape(()), l_adv_cl.detach().reshape(()), dce.detach().reshape(())]).double()
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                dp.reduce_scalars(packed)
                bal = self.gamma * packed[1] - packed[0]
                self._kt_dev.copy_(torch.clamp(self._kt_dev + self.lb * bal, 0.0, 1.0))
                self._g_out[:3].copy_(packed)
                self._g_out[3:4].copy_(self._kt_dev)
                self._g_out[4:5].add_(packed[2] * n_glob)
                self._g_out[5:6].add_(n_glob)
                self._kt_ev = torch.cuda.Event()
                self._kt_ev.record(aux)
            packed.record_stream(aux)
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


def _shard_paired(dp, data):
    """Strided shard of a `_collate_fn_paired` tuple (inputs, cleans, mask, targets, pct, target_sizes) handed out by a loader
    that is not data-parallel aware."""
    inputs, cleans, mask, targets, pct, tsz = data
    ny = dp.shard_collated((inputs, targets, pct, tsz, mask))
    idx = torch.tensor(dp.shard_rows(inputs.size(0)), dtype=torch.long)
    return (ny[0], cleans.index_select(0, idx.to(cleans.device)), ny[4], ny[1], ny[2], ny[3])
