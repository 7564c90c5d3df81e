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
from .optim import Adam
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

    def zero_grad_all(self):
        self.G.zero_grad(); self.D.zero_grad()

    def build_model(self):
        c = self.config
        rt = supported_rnns[c.rnn_type]
        self.G = stackedBRNN(I=c.nFeat, O=c.nFeat, H=c.rnn_size, L=c.rnn_layers, rnn_type=rt)
        self.D = stackedBRNN(I=2 * c.nFeat, O=c.nFeat, H=c.rnn_size, L=c.rnn_layers, rnn_type=rt)

    def get_gradient_norm(self, model):
        acc = torch.zeros((1,), device=next(model.parameters()).device, dtype=torch.float64)
        for p in model.parameters():
            if p.grad is not None:
                ops.sqsum_into(acc, p.grad)
        return acc.sqrt().to(torch.float32)

    def train_step(self, data_list, iter=0):
        c = self.config
        if self._opts is None:
            mk = lambda m: Adam(m.parameters(), lr=c.lr, betas=(self.beta1, self.beta2), amsgrad=True)
            self._opts = (mk(self.G), mk(self.D))
        optimizer_g, optimizer_d = self._opts
        self.zero_grad_all()
        mask = data_list[2]
        attach_n_valid(mask) if not mask.is_cuda else None
        mixture, cleans, mask = _get_variable_nograd(data_list[0]), _get_variable_nograd(data_list[1]), _get_variable_nograd(mask)
        enhanced = self.G(mixture)
        ae_ny_G = self.D.forward_paired(enhanced, mixture)
        l_adv_ny_G, _ = self.diffLoss(ae_ny_G, enhanced, mask)
        l_adv_ny_G = l_adv_ny_G * c.w_adversarial
        l_adv_ny_G.backward(retain_graph=True)
        # D-step = (-kt) x the G-step's D-parameter gradients (same identity as the AAS trainer)
        for p in self.D.parameters():
            if p.grad is not None:
                ops.axpby_(p.grad, p.grad, -float(self.kt), 0.0)
        dce, nElement = self.diffLoss(enhanced, cleans, mask)
        if not self.as_written:
            dce.backward()
        ae_cl = self.D.forward_paired(cleans, mixture)
        l_adv_cl, _ = self.diffLoss(ae_cl, cleans, mask)
        l_adv_cl = c.w_adversarial * l_adv_cl
        l_adv_cl.backward()
        g_norm = self.get_gradient_norm(self.G)
        optimizer_g.step(); optimizer_d.step()
        l_adv_ny_G_data, l_adv_cl_data, dce_loss, g_norm = torch.stack(
            [l_adv_ny_G.detach().reshape(()), l_adv_cl.detach().reshape(()), dce.detach().reshape(()), g_norm.reshape(())]).tolist()
        self.dce_tr_local.update(dce_loss, nElement)
        g_d_balance = self.gamma * l_adv_cl_data - l_adv_ny_G_data
        self.kt += self.lb * g_d_balance
        self.kt = max(min(1, self.kt), 0)
        return dict(l_adv_ny_G=l_adv_ny_G_data, l_adv_cl=l_adv_cl_data, dce=dce_loss, kt=self.kt,
                    conv_measure=l_adv_cl_data + abs(g_d_balance), g_norm=g_norm)

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
