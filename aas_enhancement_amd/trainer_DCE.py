"""minimize_DCE trainer (reference Speech_enhancement_by_AAS/trainer_DCE.py; hot loop :111-127)."""
import os

import torch

from .model import L1Loss_mask, stackedBRNN, supported_rnns
from . import ops
from .utils import AverageMeter, _get_variable_nograd, attach_n_valid


class Trainer(object):
    def __init__(self, config, data_loader=None, models=None):
        self.config, self.data_loader = config, data_loader
        self.lr, self.beta1, self.beta2 = config.lr, config.beta1, config.beta2
        self.diffLoss = L1Loss_mask()
        self.model_dir = "logs/" + str(config.expnum)
        self.dce_tr = AverageMeter()
        if models is not None:
            (self.G,) = models
        else:
            self.build_model()
        if config.gpu >= 0:
            self.G.cuda()
        self.logFile = None
        if config.mode == "train" and getattr(config, "write_log", True):
            os.makedirs(self.model_dir, exist_ok=True)
            self.logFile = open(self.model_dir + "/log.txt", "w")
        self._opt = None

    def zero_grad_all(self):
        self.G.zero_grad()

    def build_model(self):
        c = self.config
        print("initialize enhancement model")
        self.G = stackedBRNN(I=c.nFeat, H=c.rnn_size, L=c.rnn_layers, rnn_type=supported_rnns[c.rnn_type])

    def train_step(self, data_list, iter=0):
        """:116-127; data_list = (inputs, cleans, mask, ...) (_collate_fn_paired order)."""
        if self._opt is None:
            from .dist import FlatBuffers
            from .optim import FlatAdam
            ops.name_layers(self.G, "G")
            self._flat = FlatBuffers(self.G)
            self._opt = FlatAdam(self._flat, lr=self.config.lr, betas=(self.beta1, self.beta2), amsgrad=True)
        mask = data_list[2]
        attach_n_valid(mask) if not mask.is_cuda else None
        inputs, cleans, mask = _get_variable_nograd(data_list[0]), _get_variable_nograd(data_list[1]), _get_variable_nograd(mask)
        outputs = self.G(inputs)
        dce, nElement = self.diffLoss(outputs, cleans, mask)
        ops.sync_wgrad()
        self._flat.zero_grad()
        dce.backward()
        ops.sync_wgrad()   # the recurrent layers' weight gradients accumulate into the flat buffer on a side stream
        self._opt.step()
        return dict(dce=dce, nElement=nElement, outputs=outputs)

    def train(self):
        from tqdm import trange
        c = self.config
        for iter in trange(c.start_iter, c.max_iter):
            r = self.train_step(self.data_loader.next(cl_ny="ny", type="train"), iter)
            if (iter + 1) % c.log_iter == 0:
                s = "[{}/{}] (train) DCE: {:.7f}".format(iter, c.max_iter, float(r["dce"]))
                print(s)
                if self.logFile:
                    self.logFile.write(s + "\n"); self.logFile.flush()
