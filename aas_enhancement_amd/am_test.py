"""Evaluation / logits dump of a trained acoustic model - the reference's AM_training/test.py (:100-202):
`--decoder greedy` prints the test-set WER / CER (greedy CTC decoding on the device), `--decoder none` saves the list of
(logits [T',N,C], sizes [N]) per batch with np.save for the offline beam-search / LM tuning tools (`tune_decoder.py`, out of scope).

    python -m aas_enhancement_amd.am_test --model_path models/deepspeech_final.pth.tar --test_manifest test.csv --decoder greedy
"""
import argparse

import torch

from .am_train import AMTrainer, dump_logits
from .data_loader import DataLoader
from .model import DeepSpeech


def main(argv=None):
    ap = argparse.ArgumentParser(description="DeepSpeech acoustic-model evaluation (AM_training/test.py flags)")
    ap.add_argument("--model_path", required=True)
    ap.add_argument("--test_manifest", required=True)
    ap.add_argument("--batch_size", default=20, type=int)
    ap.add_argument("--num_workers", default=1, type=int)
    ap.add_argument("--decoder", default="greedy", choices=("greedy", "none"), help="beam / LM decoding needs ctcdecode + KenLM: out of scope")
    ap.add_argument("--output_path", default=None, help="where np.save writes the logits when --decoder none")
    ap.add_argument("--result_path", default=None)
    ap.add_argument("--gpu", default=0, type=int)
    ap.add_argument("--preprocess", default="file")
    ap.add_argument("--transcript_prob", default=0.0, type=float)
    a = ap.parse_args(argv)
    torch.cuda.set_device(a.gpu)
    model = DeepSpeech.load_model(a.model_path, gpu=a.gpu)
    labels = DeepSpeech.get_labels(model)
    dl = DataLoader(batch_size=a.batch_size, val_manifest=a.test_manifest, labels=labels, num_workers=a.num_workers, pin_memory=True,
                    preprocess=a.preprocess, n_mels=model.nFreq)
    batches = (dl.next("ny", "val") for _ in range(dl.num_batches("val")))
    if a.decoder == "none":
        out = dump_logits(model, batches, a.output_path)
        print("saved logits of %d batches to %s" % (len(out), a.output_path))
        return out
    tr = AMTrainer(model, labels=labels)
    wer, cer = tr.validate(batches, transcript_prob=a.transcript_prob)
    line = "Test Summary \tAverage WER {wer:.3f}\tAverage CER {cer:.3f}\t".format(wer=wer, cer=cer)
    print(line)
    if a.result_path:
        with open(a.result_path, "w") as f:
            f.write(line + "\n")
    return wer, cer


if __name__ == "__main__":
    main()
