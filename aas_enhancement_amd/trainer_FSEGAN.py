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


class Trainer(object):
    def __init__(self, config, data_loader=None, models=None):
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
        if config.mode == "train" and getattr(config, "write_log", True):
            os.makedirs(self.model_dir, exist_ok=True)
            self.logFile = open(self.model_dir + "/log.txt", "w")
        self._opts = None
        self._flat = None

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
        from .dist import FlatBuffers
        from .optim import FlatAdam
        c = self.config
        for name, m in (("G", self.G), ("D", self.D)):
            ops.name_layers(m, name)
        self._flat = {"G": FlatBuffers(self.G), "D": FlatBuffers(self.D)}
        mk = lambda f: FlatAdam(f, lr=c.lr, betas=(self.beta1, self.beta2), amsgrad=True)
        self._opts = (mk(self._flat["G"]), mk(self._flat["D"]))
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

    def train_step(self, data_list, iter=0):
        """:128-182 (intended semantics) with the same identities as the AAS trainer: D(enhanced | mixture) and
        D(clean | mixture) share ONE batched pass of 2N rows, the D-step parameter gradients of the enhanced half are (-kt) x
        its G-step ones (per-utterance weights on the weight-gradient products only), and E is back-propagated once with
        d(adv)/d(enhanced) + d(dce)/d(enhanced)."""
        c = self.config
        if self._opts is None:
            self.make_optimizers()
        optimizer_g, optimizer_d = self._opts
        ops.sync_wgrad()
        for f in self._flat.values():
            f.zero_grad()
        mask = data_list[2]
        attach_n_valid(mask) if not mask.is_cuda else None
        mixture, cleans, mask = _get_variable_nograd(data_list[0]), _get_variable_nograd(data_list[1]), _get_variable_nograd(mask)
        N = mixture.size(0)
        enhanced = self.G(mixture)
        leaf = enhanced.detach().requires_grad_(True)
        rs = torch.empty(2 * N, device=leaf.device, dtype=torch.float32)
        rs[:N] = -float(self.kt)
        rs[N:] = 1.0
        rs._aas_classes = [(0, N, rs[0:1]), (N, N, None)]   # the two utterance classes and their weights (ops.gemm_planes_tn)
        paired = torch.cat([torch.cat([leaf, mixture], 1), torch.cat([cleans, mixture], 1)], 0)   # forward_paired x 2 (model.py:233-238)
        ae = self.D(paired, wgrad_row_scale=rs)
        l_adv_ny_G, _ = self.diffLoss(ae[:N], leaf, mask)
        l_adv_ny_G = l_adv_ny_G * c.w_adversarial
        l_adv_cl, _ = self.diffLoss(ae[N:], cleans, mask)
        l_adv_cl = c.w_adversarial * l_adv_cl
        dce, nElement = self.diffLoss(leaf, cleans, mask)
        total = l_adv_ny_G + l_adv_cl
        if not self.as_written:   # the reference only logs the DCE term (:161-163); the intended G loss back-propagates it
            total = total + dce
        total.backward()
        enhanced.backward(leaf.grad)
        g_norm = self.get_gradient_norm(self.G)
        optimizer_g.step(); optimizer_d.step()
        l_adv_ny_G_data, l_adv_cl_data, dce_loss, g_norm = torch.stack(
            [l_adv_ny_G.detach().reshape(()), l_adv_cl.detach().reshape(()), dce.detach().reshape(()), g_norm.reshape(())]).tolist()
        ops.check_rnn_health((l_adv_ny_G_data, l_adv_cl_data, dce_loss))
        self.dce_tr_local.update(dce_loss, nElement)
        g_d_balance = self.gamma * l_adv_cl_data - l_adv_ny_G_data
        self.kt += self.lb * g_d_balance
        self.kt = max(min(1, self.kt), 0)
        return dict(l_adv_ny_G=l_adv_ny_G_data, l_adv_cl=l_adv_cl_data, dce=dce_loss, kt=self.kt,
                    conv_measure=l_adv_cl_data + abs(g_d_balance), g_norm=g_norm, enhanced=enhanced)

    def train(self):
        from tqdm import trange
        c = self.config
        for iter in trange(c.start_iter, c.max_iter):
            r = self.train_step(self.data_loader.next(cl_ny="ny", type="train"), iter)
            if (iter + 1) % c.log_iter == 0:
                for s in ("[{}/{}] (train) DCE: {:.7f}, ADV_cl: {:.7f}, ADV_ny: {:.7f}".format(iter, c.max_iter, self.dce_tr_local.avg, r["l_adv_cl"], r["l_adv_ny_G"]),
                          "[{}/{}] (train) conv_measure: {:.4f}, kt: {:.4f} ".format(iter, c.max_iter, r["conv_measure"], self.kt)):
                    print(s)
                    if self.logFile:
                        self.logFile.write(s + "\n")
                if self.logFile:
                    self.logFile.flush()
                self.dce_tr_local.reset()
