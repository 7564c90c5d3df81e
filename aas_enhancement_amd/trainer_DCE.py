"""minimize_DCE trainer (reference Speech_enhancement_by_AAS/trainer_DCE.py; hot loop :111-127)."""
import os

from .model import L1Loss_mask, stackedBRNN, supported_rnns
from . import ops
from .utils import AverageMeter, _get_variable_nograd, attach_n_valid


class Trainer(ops.TrainerContext):
    def __init__(self, config, data_loader=None, models=None):
        self._init_context()   # arithmetic mode + launch settings this trainer runs in (ops.TrainerContext)
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
        if config.mode == "train" and getattr(config, "write_log", True) and int(os.environ.get("RANK", "0")) == 0:
            os.makedirs(self.model_dir, exist_ok=True)
            self.logFile = open(self.model_dir + "/log.txt", "w")
        self._opt = None
        self.dp = None

    def zero_grad_all(self):
        self.G.zero_grad()

    def build_model(self):
        c = self.config
        print("initialize enhancement model")
        self.G = stackedBRNN(I=c.nFeat, H=c.rnn_size, L=c.rnn_layers, rnn_type=supported_rnns[c.rnn_type])

    def make_optimizers(self):
        from .dist import BucketReducer, DPContext, FlatBuffers
        from .optim import FlatAdam
        self.dp = getattr(self, "dp", None) or DPContext.from_env()
        ops.name_layers(self.G, "G")
        self._flat = FlatBuffers(self.G)
        self._opt = FlatAdam(self._flat, lr=self.config.lr, betas=(self.beta1, self.beta2), amsgrad=True)
        self._reducer = BucketReducer(self.dp, [self._flat]) if self.dp.active else None
        return self._opt

    @ops.with_trainer_precision
    def train_step(self, data_list, iter=0):
        """:116-127; data_list = (inputs, cleans, mask, ...) (_collate_fn_paired order).  Data parallel: `data_list` is this
        rank's shard, the loss is normalised by the GLOBAL nElement and the flat gradient buffer is SUM-all-reduced bucket by
        bucket behind the weight-gradient products."""
        if self._opt is None:
            self.make_optimizers()
        dp = self.dp
        mask = data_list[2]
        attach_n_valid(mask) if not mask.is_cuda else None
        inputs, cleans, mask = _get_variable_nograd(data_list[0]), _get_variable_nograd(data_list[1]), _get_variable_nograd(mask)
        nElement = getattr(mask, "n_valid", None)
        if nElement is None:
            nElement = int(mask.numel()) - int(mask.sum().item())
        if dp.active:
            (nElement,) = dp.global_counts([nElement])
            self._reducer.begin()
            ops.WGRAD_HOOK[0] = self._reducer.on_wgrad
        try:
            outputs = self.G(inputs)
            dce = ops.l1_sum(outputs, cleans) / nElement
            ops.sync_wgrad()
            self._flat.zero_grad()
            dce.backward()
            ops.sync_wgrad()   # the recurrent layers' weight gradients accumulate into the flat buffer on a side stream
            if dp.active:
                self._reducer.flush(self._flat)
                self._reducer.wait()
        finally:
            ops.WGRAD_HOOK[0] = None
        self._opt.step()
        ops.refresh_weight_planes(self.G)
        if dp.active:
            dce = dp.reduce_scalars(dce.detach().reshape(1).clone()).reshape(())
        return dict(dce=dce, nElement=nElement, outputs=outputs)

    def train(self):
        from tqdm import trange
        from .trainer_FSEGAN import _shard_paired
        c = self.config
        self.make_optimizers()
        rank0 = self.dp.rank == 0
        presharded = getattr(self.data_loader, "dp", None) is not None
        for iter in trange(c.start_iter, c.max_iter, disable=not rank0):
            data = self.data_loader.next(cl_ny="ny", type="train")
            if self.dp.active and not presharded:
                data = _shard_paired(self.dp, data)
            r = self.train_step(data, iter)
            if (iter + 1) % c.log_iter == 0:
                v = float(r["dce"])
                ops.check_rnn_health((v,))
                s = "[{}/{}] (train) DCE: {:.7f}".format(iter, c.max_iter, v)
                if rank0:
                    print(s)
                if self.logFile:
                    self.logFile.write(s + "\n"); self.logFile.flush()
