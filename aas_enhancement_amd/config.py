"""Command-line configuration with the reference's flag names and defaults
(Speech_enhancement_by_AAS/config.py:18-64) plus the data-parallel / schedule flags this build adds."""
import argparse


def str2bool(v):
    return v.lower() in ("true", "1")


parser = argparse.ArgumentParser()
parser.add_argument("--trainer", type=str, default="AAS")
parser.add_argument("--mode", type=str, default="train", help="train | test | visualize")
parser.add_argument("--simul_real", type=str, default="real", help="simul | real | simulreal")
parser.add_argument("--DB_name", type=str, default="librispeech", help="librispeech | chime")
parser.add_argument("--expnum", type=int, default=0)
parser.add_argument("--gpu", default=-1, type=int)
parser.add_argument("--print_every", type=int, default=100)
parser.add_argument("--load_path", type=str, default="")
parser.add_argument("--ASR_path", type=str, default="")
parser.add_argument("--tr_cl_manifest", default="")
parser.add_argument("--tr_ny_manifest", default="")
parser.add_argument("--trsub_manifest", default="")
parser.add_argument("--val_manifest", default="")
parser.add_argument("--val2_manifest", default="")
parser.add_argument("--batch_size", default=20, type=int, help="Batch size for training")
parser.add_argument("--labels_path", default="labels.json", help="Contains all characters for transcription")
parser.add_argument("--nFeat", default=40, type=int)
parser.add_argument("--rnn_size", default=500, type=int, help="Hidden size of RNNs")
parser.add_argument("--rnn_layers", default=4, type=int, help="Number of RNN layers")
parser.add_argument("--rnn_type", default="lstm", help="Type of the RNN. gru|lstm are supported")
parser.add_argument("--epochs", default=300, type=int)
parser.add_argument("--start_iter", default=0, type=int)
parser.add_argument("--max_iter", default=30000000, type=int)
parser.add_argument("--log_iter", default=100, type=int)
parser.add_argument("--save_iter", default=1000, type=int)
parser.add_argument("--lr", "--learning-rate", default=1e-5, type=float, help="initial learning rate")
parser.add_argument("--w_acoustic", default=1, type=float)
parser.add_argument("--w_adversarial", default=1, type=float)
parser.add_argument("--allow_ASR_update_iter", type=int, default=0)
parser.add_argument("--gamma", type=float, default=0.5, help="began parameter")
parser.add_argument("--lambda_k", type=float, default=0.001, help="began parameter")
parser.add_argument("--optimizer", default="adam", help="adam|sgd")
parser.add_argument("--random_seed", type=int, default=123)
parser.add_argument("--beta1", type=float, default=0.5)
parser.add_argument("--beta2", type=float, default=0.999)
# additions of this build (not in the reference)
parser.add_argument("--schedule", default="fused", help="fused | as_executed (results identical)")
parser.add_argument("--fsegan_as_written", type=str2bool, default=False)
parser.add_argument("--sync_bn", type=str2bool, default=False,
                    help="data parallel: all-reduce A's BatchNorm statistics (global-batch BN; default: local-batch BN per rank)")
parser.add_argument("--dist_backend", default="nccl", help="torch.distributed backend under torchrun: nccl (= RCCL on ROCm) | gloo")
parser.add_argument("--precision", default=None, choices=("fp32", "fp32eq", "bf16x3"),
                    help="arithmetic of the GEMMs and recurrent products (absent: AAS_PRECISION from the environment, else fp32): fp32 (the reference's) | fp32eq (fp32 recurrent "
                         "products, large GEMMs as six bf16 products of three-term operands: fp32-equivalent, ~8 %% faster) | bf16x3 "
                         "(split-bf16 fast mode: hi*hi + lo*hi + hi*lo on bf16 MFMA, ~2x faster, inside the 1e-3 / 1e-2 parity budget)")
parser.add_argument("--preprocess", default="file", help="file: manifests list precomputed LMFB .pt7 tensors (the reference's only "
                                                         "mode) | code: manifests list 16 kHz waveform tensors and the LMFB "
                                                         "HIP kernel extracts log-Mel features on the fly (AM_training/train.py:59)")


def get_config(argv=None):
    config, unparsed = parser.parse_known_args(argv)
    if len(unparsed) > 0:
        print(unparsed)
        assert len(unparsed) == 0, "length of unparsed option should be 0"
    return config, unparsed
