"""What the four enhancement trainers share OUTSIDE the training step: loading the pre-trained acoustic model, resuming from a
`G_valmin_*` checkpoint, the greedy-decoding half of every validation pass, the log file, and the `G_<iter>.pth` /
`G_valmin_<iter>.pth` checkpoint lifecycle.

Reference (Speech_enhancement_by_AAS/): trainer_AAS.py:88-123,215-351, trainer_DCE.py:60-108,130-250,
trainer_FSEGAN.py:85-121,199-317, trainer_acoustic.py:79-111,147-245.  The four files repeat these blocks with small
differences; the differences are arguments here (which networks are saved, which meters a pass fills, the comparison that triggers a
resume from the newest checkpoint: `start_iter < 0` in trainer_DCE.py:93, `<= 0` in the other three)."""
import os
import random
from glob import glob
from shutil import copyfile

import torch

from .utils import AverageMeter


class ValidationMixin(object):
    """Mixed into the trainer classes; expects `config`, `data_loader`, `G`, `ASR`, `decoder`, `model_dir`, `logFile`."""

    # ---- construction-time pieces -----------------------------------------------------------------------------------------
    def _init_validation_state(self, meters):
        """The bookkeeping of the checkpoint lifecycle (trainer_*.py __init__) and the AverageMeters of the validation passes."""
        self.valmin_iter = 0
        self.savename_G = self.savename_D = self.savename_ASR = ""
        for m in meters:
            setattr(self, m, AverageMeter())

    def load_asr_package(self):
        """`package_ASR = torch.load(config.ASR_path, ...); DeepSpeech.load_model_package(package_ASR)` (trainer_DCE.py:77-80 and
        the same lines of the other trainers)."""
        from .model import DeepSpeech
        print("load pre-trained ASR model")
        package_ASR = torch.load(self.config.ASR_path, map_location=lambda storage, loc: storage)
        return DeepSpeech.load_model_package(package_ASR)

    def _open_log(self):
        c = self.config
        self.logFile = None
        if c.mode == "train" and getattr(c, "write_log", True) and int(os.environ.get("RANK", "0")) == 0:
            os.makedirs(self.model_dir, exist_ok=True)
            self.logFile = open(self.model_dir + "/log.txt", "w")

    def _log(self, s, flush=False, echo=True):
        if echo:
            print(s)
        if self.logFile:
            self.logFile.write(s + "\n")
            if flush:
                self.logFile.flush()

    def load_model(self, resume_newest_when=lambda start_iter: start_iter <= 0):
        """Resume G from `<load_path>/G_valmin_<iter>.pth` (trainer_AAS.py:98-123, trainer_FSEGAN.py:96-121, trainer_DCE.py:82-108:
        only G is restored, D / A start from their initial state, as in the reference).  `start_iter` at or below the trigger
        -> the newest checkpoint's iteration."""
        c = self.config
        print("[*] Load models from {}...".format(c.load_path))
        postfix = "_valmin"
        paths = sorted(glob(os.path.join(c.load_path, "G{}*.pth".format(postfix))))
        if len(paths) == 0:
            print("[!] No checkpoint found in {}...".format(c.load_path))
            raise AssertionError("checkpoint not avilable")
        idxes = [int(os.path.basename(p.split(".")[0].split("_")[-1])) for p in paths]
        if resume_newest_when(c.start_iter):
            c.start_iter = max(idxes)
            if c.start_iter < 0:
                raise Exception("start iter is still less than 0 --> probably try to load initial random model")
        print("Load models from " + c.load_path + ", ITERATION = " + str(c.start_iter))
        self.G.load_state_dict(torch.load("{}/G{}_{}.pth".format(c.load_path.rstrip("/"), postfix, c.start_iter),
                                          map_location=lambda storage, loc: storage))
        print("[*] Model loaded")

    # ---- the decoding half of every validation function ---------------------------------------------------------------------
    def _greedy_pass(self, enhanced, targets, input_percentages, target_sizes, transcript_prob=0.001):
        """Step 1 of greedy_decoding / greedy_decoding_and_FSEGAN / greedy_decoding_and_AAS (trainer_DCE.py:209-250,
        trainer_FSEGAN.py:279-306, trainer_AAS.py:303-340) from the enhanced features on: prob = ASR(enhanced) time-major,
        sizes = int(pct * T'), argmax decoding on the device, WER / CER sums on the host.
        -> (prob [T', N, C], sizes, wer, cer, total_word, total_char); wer = we / total_word and - the reference's quirk,
        kept - cer = ce / total_word.  (`input_percentages` is not modified: the reference's in-place `mul_` would corrupt a
        batch that a caller holds on to.)"""
        split_targets, offset = [], 0
        for size in target_sizes:
            split_targets.append(targets[offset:offset + int(size)])
            offset += int(size)
        prob = self.ASR(enhanced).transpose(0, 1)
        T = prob.size(0)
        sizes = input_percentages.clone().mul_(int(T)).int()
        decoded_output, _ = self.decoder.decode(prob.detach(), sizes)
        target_strings = self.decoder.convert_to_strings(split_targets)
        we = ce = total_word = total_char = 0
        for x in range(len(target_strings)):
            decoding, reference = decoded_output[x][0], target_strings[x][0]
            nChar, nWord = len(reference), len(reference.split())
            we_i, ce_i = self.decoder.wer(decoding, reference), self.decoder.cer(decoding, reference)
            we += we_i; ce += ce_i; total_word += nWord; total_char += nChar
            if random.uniform(0, 1) < transcript_prob:
                print("reference = " + reference)
                print("decoding = " + decoding)
                print("wer = " + str(we_i / float(max(nWord, 1))) + ", cer = " + str(ce_i / float(max(nChar, 1))))
        # (an all-empty reference set would divide by zero in the reference; it reads 0 / 1 here)
        return prob, sizes, we / max(total_word, 1), ce / max(total_word, 1), total_word, total_char

    # ---- checkpoint lifecycle -----------------------------------------------------------------------------------------------
    def _save_rotating(self, tag, net, iter):
        """`<model_dir>/<tag>_<iter>.pth`, the previous file of that tag removed (trainer_DCE.py:194-198)."""
        os.makedirs(self.model_dir, exist_ok=True)
        attr = "savename_" + tag
        prev = getattr(self, attr, "")
        if len(prev) > 0 and os.path.exists(prev):
            os.remove(prev)
        name = "{}/{}_{}.pth".format(self.model_dir, tag, iter)
        torch.save(net.state_dict(), name)
        setattr(self, attr, name)
        return name

    def _keep_if_best(self, iter, wer_avg, tags=("G",)):
        """`<tag>_valmin_<iter>.pth` when the validation WER improved, the previous best removed (trainer_DCE.py:200-208)."""
        if self.G.loss_stop > wer_avg:
            self.G.loss_stop = wer_avg
            for tag in tags:
                prev = "{}/{}_valmin_{}.pth".format(self.model_dir, tag, self.valmin_iter)
                if os.path.exists(prev):
                    os.remove(prev)
                print("save model for this checkpoint")
                copyfile(getattr(self, "savename_" + tag), "{}/{}_valmin_{}.pth".format(self.model_dir, tag, iter))
            self.valmin_iter = iter
            return True
        return False

    def _validation_sets(self):
        """(log name, loader key) of the two passes every save_iter block runs."""
        return (("training subset", "trsub"), ("validation", "val"))

    def _save_iter_block(self, iter):
        """The data-parallel wrapper of a save_iter block: rank 0 validates and writes the checkpoints (every rank holds
        identical parameters), without SyncBN collectives that no other rank would match; afterwards every rank takes rank 0's
        BatchNorm buffers of A (A stays in train mode during validation, as in the reference, so they moved on rank 0 only)."""
        from . import ops
        dp = getattr(self, "dp", None)
        if dp is None or dp.rank == 0:
            armed, self.launch.sync_bn = self.launch.sync_bn, None
            try:
                self.validate_and_checkpoint(iter)
            finally:
                self.launch.sync_bn = armed
        if dp is not None and dp.active:
            dp.barrier()
            if getattr(self, "ASR", None) is not None:
                dp.broadcast_buffers(self.ASR, src=0)
