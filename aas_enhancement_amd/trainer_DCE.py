"""minimize_DCE trainer on the MI355X HIP path - same API as the reference's Speech_enhancement_by_AAS/trainer_DCE.py
(`Trainer.__init__ / build_model / load_model / train / zero_grad_all / greedy_decoding`; hot loop :111-127, the save_iter block
with validation through the pre-trained acoustic model and the `G_<iter>.pth` / `G_valmin_<iter>.pth` lifecycle :130-207,
`greedy_decoding` :209-250).  BASELINE config 1's entry point: `python -m aas_enhancement_amd.main --trainer minimize_DCE`."""
import torch

from . import ops
from .decoder import GreedyDecoder
from .model import L1Loss_mask, stackedBRNN, supported_rnns
from .utils import _get_variable_nograd, _get_variable_volatile, attach_n_valid
from .validation import ValidationMixin


class Trainer(ops.TrainerContext, ValidationMixin):
    def __init__(self, config, data_loader=None, models=None):
        """models: (G,) or (G, ASR) - networks built by the caller (tests / bench); None -> build_model() (:73-80)."""
        self._init_context()   # arithmetic mode + launch settings this trainer runs in (ops.TrainerContext)
        self.config, self.data_loader = config, data_loader
        self.lr, self.beta1, self.beta2 = config.lr, config.beta1, config.beta2
        self.optimizer = getattr(config, "optimizer", "adam")
        self.batch_size = config.batch_size
        self.diffLoss = L1Loss_mask()
        self.model_dir = "logs/" + str(config.expnum)
        self.decoder = GreedyDecoder(data_loader.labels) if data_loader is not None and getattr(data_loader, "labels", None) else None
        self.kt, self.lb, self.conv_measure = 0, 0.001, 0      # (:46-48: carried by the reference class, unused by this trainer)
        self._init_validation_state(("dce_tr", "dce_val", "wer_tr", "cer_tr", "wer_val", "cer_val"))
        self.ASR = None
        if models is not None:
            self.G = models[0]
            if len(models) > 1:
                self.ASR = models[1]
        else:
            self.build_model()
        self.G.loss_stop = 100000
        if config.gpu >= 0:
            self.G.cuda()
            if self.ASR is not None:
                self.ASR.cuda()
        if len(getattr(config, "load_path", "")) > 0:
            self.load_model()
        self._open_log()
        self._opt = None
        self.dp = None

    def zero_grad_all(self):
        self.G.zero_grad()

    def build_model(self):
        """:73-80"""
        c = self.config
        print("initialize enhancement model")
        self.G = stackedBRNN(I=c.nFeat, H=c.rnn_size, L=c.rnn_layers, rnn_type=supported_rnns[c.rnn_type])
        self.ASR = self.load_asr_package()

    def load_model(self):
        """:82-108 (this trainer resumes from the newest checkpoint only for a NEGATIVE start_iter, :93)."""
        ValidationMixin.load_model(self, resume_newest_when=lambda start_iter: start_iter < 0)

    def make_optimizers(self):
        from .dist import BucketReducer, DPContext, FlatBuffers
        from .optim import FlatAdam
        self.dp = getattr(self, "dp", None) or DPContext.from_env()
        ops.name_layers(self.G, "G")
        if self.ASR is not None:
            ops.name_layers(self.ASR, "ASR")
            for p in self.ASR.parameters():      # A only decodes here: no gradient ever reaches it
                p.requires_grad_(False)
        self._flat = FlatBuffers(self.G)
        self._opt = FlatAdam(self._flat, lr=self.config.lr, betas=(self.beta1, self.beta2), amsgrad=True)
        self._reducer = BucketReducer(self.dp, [self._flat]) if self.dp.active else None
        return self._opt

    @ops.with_trainer_precision
    def train_step(self, data_list, iter=0):
        """:116-127; data_list = (inputs, cleans, mask, ...) (_collate_fn_paired order).  Nothing is read back: the loss stays a
        device scalar (the log line converts it).  Data parallel: `data_list` is this rank's shard, the loss is normalised by the
        GLOBAL nElement and the flat gradient buffer is SUM-all-reduced bucket by bucket behind the weight-gradient products."""
        if self._opt is None:
            self.make_optimizers()
        dp = self.dp
        mask = data_list[2]
        attach_n_valid(mask) if not mask.is_cuda else None
        inputs, cleans, mask = _get_variable_nograd(data_list[0]), _get_variable_nograd(data_list[1]), _get_variable_nograd(mask)
        nElement = getattr(mask, "n_valid", None)
        if nElement is None:
            nElement = int(mask.numel()) - int(mask.sum().item())
        dev = inputs.device
        ops.sync_wgrad()
        if dp.active:
            # the global nElement never leaves the device: all-reduced on the utility stream, turned into 1 / nElement by one launch
            from .dist import DeviceScales
            cnt = self._upload_small(torch.tensor([float(nElement)], dtype=torch.float64), dev)
            scales = DeviceScales(dp, cnt, [1.0, 1.0, 1.0], [0, 0, 0], ops.refresh_stream(dev))
            scale = scales[0]
        else:
            scale = 1.0 / nElement
        # the loss root is a RAW device sum in a ring of eight accumulators (a caller may read a step's loss up to seven steps later);
        # ONE prologue launch zeroes the flat gradient buffer and this step's accumulator
        if getattr(self, "_l1_ring", None) is None:
            self._l1_ring, self._l1_i = torch.zeros(16, device=dev, dtype=torch.float64), 0   # (16-byte slots)
        self._l1_i = (self._l1_i + 1) % 8
        acc = self._l1_ring[2 * self._l1_i:2 * self._l1_i + 1]
        ops.step_prologue([self._flat.flat_g, acc])
        if dp.active:
            self._reducer.begin()
            self.launch.wgrad_hook = self._reducer.on_wgrad
        try:
            outputs = self.G(inputs)
            root = ops.l1_scaled(outputs, cleans, scale, acc)
            torch.autograd.backward([root], [ops.unit_root(root)])
            ops.sync_wgrad()   # the recurrent layers' weight gradients accumulate into the flat buffer on a side stream
            if dp.active:
                self._reducer.flush(self._flat)
                self._reducer.wait()
        finally:
            self.launch.wgrad_hook = None
        self._opt.step_dev()
        ops.refresh_weight_planes(self.G)
        if dp.active:
            # the logged loss: all-reduced raw sum x 1 / global nElement, formed on the utility stream (a collective: every rank, every step)
            main, aux = torch.cuda.current_stream(), ops.refresh_stream(dev)
            if getattr(self, "_no_kt", None) is None:
                self._no_kt = torch.zeros(1, device=dev, dtype=torch.float64)
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                out3 = torch.empty(3, device=dev, dtype=torch.float64)
                out6 = torch.empty(6, device=dev, dtype=torch.float64)     # (this step's own: slot 2 = the loss; the running sums are not used here)
                ops.sums_pack(None, acc, out3)
                dp.reduce_scalars(out3)
                ops.began_step_sums(out3, out3[2:], 0.0, 0.0, 1.0, self._no_kt, out6, 0.0, 0.0, 0.0, d_scales3=scales.all, d_n_batch=scales.cnt)
                ev = torch.cuda.Event()
                ev.record(aux)
            main.wait_event(ev)
            for t_ in (scales.all, scales.cnt, out6):
                t_.record_stream(main)
            return ops.StepResult(dce=out6[2], nElement=nElement, outputs=outputs)

        def dce():     # formed when a log line (or a test) reads it: L = sum / nElement
            return (acc.clone() / nElement).reshape(()).to(torch.float32)
        return ops.StepResult(nElement=nElement, outputs=outputs, lazy=dict(dce=dce))

    def train(self):
        """:111-207"""
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
                self._log("[{}/{}] (train) DCE: {:.7f}".format(iter, c.max_iter, v), flush=True, echo=rank0)
            if (iter + 1) % c.save_iter == 0:
                self._save_iter_block(iter)

    # ---- validation + checkpoint lifecycle (:130-207) -------------------------------------------------------------------------
    @ops.with_trainer_precision
    def validate_and_checkpoint(self, iter):
        c = self.config
        if self.ASR is None or self.decoder is None:
            raise RuntimeError("minimize_DCE validation decodes through the pre-trained acoustic model: --ASR_path (build_model) "
                               "or models=(G, ASR), and a data loader with `labels`, are needed")
        self.G.eval()
        for (name, dl), (dce_m, wer_m, cer_m) in zip(self._validation_sets(), ((self.dce_tr, self.wer_tr, self.cer_tr),
                                                                               (self.dce_val, self.wer_val, self.cer_val))):
            for m in (dce_m, wer_m, cer_m):
                m.reset()
            for _ in range(self.data_loader.num_batches(dl)):
                d = self.data_loader.next(cl_ny="ny", type=dl)
                with torch.no_grad():
                    dce, nElement, wer, cer, nWord, nChar = self._validate_batch(d)
                dce_m.update(float(dce), nElement)
                wer_m.update(wer, nWord)
                cer_m.update(cer, nChar)
            self._log("[{}/{}] ({}) DCE: {:.7f}".format(iter, c.max_iter, name, dce_m.avg))
            self._log("[{}/{}] ({}) WER: {:.7f}, CER: {:.7f}".format(iter, c.max_iter, name, wer_m.avg * 100, cer_m.avg * 100))
        self.G.train()   # end of validation
        if self.logFile:
            self.logFile.flush()
        self._save_rotating("G", self.G, iter)
        self._keep_if_best(iter, self.wer_val.avg, ("G",))

    def _validate_batch(self, data_list):
        """One batch of :139-153: DCE of G(inputs) against the clean features and the greedy decoding of the SAME enhanced
        features (the reference runs G a second time inside greedy_decoding, :219; G is in eval mode and deterministic)."""
        mask = data_list[2]
        attach_n_valid(mask) if not mask.is_cuda else None
        inputs, cleans, mask = _get_variable_volatile(data_list[0]), _get_variable_volatile(data_list[1]), _get_variable_volatile(mask)
        targets, input_percentages, target_sizes = data_list[3], data_list[4], data_list[5]
        outputs = self.G(inputs)
        dce, nElement = self.diffLoss(outputs, cleans, mask)
        _, _, wer, cer, nWord, nChar = self._greedy_pass(outputs, targets, input_percentages, target_sizes)
        return dce, nElement, wer, cer, nWord, nChar

    @ops.with_trainer_precision
    def greedy_decoding(self, inputs, targets, input_percentages, target_sizes, transcript_prob=0.001):
        """:209-250 -> (wer, cer, total_word, total_char)."""
        enhanced = self.G(_get_variable_volatile(inputs))
        _, _, wer, cer, total_word, total_char = self._greedy_pass(enhanced, targets, input_percentages, target_sizes, transcript_prob)
        return wer, cer, total_word, total_char
