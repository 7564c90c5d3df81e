"""Entry point with the reference's CLI (Speech_enhancement_by_AAS/main.py:8-89):
    python -m aas_enhancement_amd.main --trainer AAS --DB_name librispeech --ASR_path pkg.pth.tar ...
"""
import json
import os

import torch

from .config import get_config


def main(config):
    from .data_loader import DataLoader
    if config.trainer == "minimize_DCE":
        from .trainer_DCE import Trainer
        paired = True
    elif config.trainer == "acoustic_supervision":
        from .trainer_acoustic import Trainer  # E + A only, loss = CTC / N (trainer_acoustic.py:120-142): no discriminator runs
        paired = False
    elif config.trainer == "AAS":
        from .trainer_AAS import Trainer
        paired = False
    elif config.trainer == "FSEGAN":
        from .trainer_FSEGAN import Trainer
        paired = True
    else:
        raise ValueError("unknown trainer %r" % config.trainer)
    # Data parallel (new in this build; the reference is single-GPU, main.py:28-30): launched with
    #   python -m torch.distributed.run --nproc-per-node N -m aas_enhancement_amd.main --trainer AAS ...
    # one process per GPU; RANK / LOCAL_RANK / WORLD_SIZE come from the launcher's environment.  Every rank reads the same
    # manifests with the same seed, so all ranks draw the same global minibatch and keep their strided shard of it.
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        config.gpu = local_rank
        torch.cuda.set_device(local_rank)   # (device selection only: nothing has touched the GPU before this point)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            kw = dict(device_id=torch.device("cuda", local_rank)) if config.dist_backend == "nccl" else {}
            dist.init_process_group(config.dist_backend, **kw)
    from . import ops
    # --precision given: it selects the arithmetic of the run; absent: the library's setting stands (AAS_PRECISION in the
    # environment, else fp32)
    if getattr(config, "precision", None) is not None:
        ops.set_precision({"fp32": 0, "bf16x3": 1, "fp32eq": 2}[config.precision])
    import numpy as np
    np.random.seed(config.random_seed)      # FeatSampler shuffles with numpy: identical batch order on every rank
    if config.gpu >= 0:
        torch.cuda.manual_seed(config.random_seed)
        torch.cuda.set_device(config.gpu)
    sfx = "_paired" if paired else ""
    if config.DB_name == "librispeech":
        config.tr_ny_manifest = "data/libri_tr_ny%s.csv" % sfx
        config.trsub_manifest = "data/libri_trsub_ny%s.csv" % sfx
        config.val_manifest = "data/libri_val%s.csv" % sfx
        config.tr_cl_manifest = "data/libri_tr_cl.csv"
    elif config.DB_name == "chime":
        config.tr_ny_manifest = "data/chime_%s_tr_ny%s.csv" % (config.simul_real, sfx)
        config.trsub_manifest = "data/chime_%s_trsub_ny%s.csv" % (config.simul_real, sfx)
        config.val_manifest = "data/chime_real_val%s.csv" % sfx
        config.val2_manifest = "data/chime_simul_val%s.csv" % sfx  # (the reference has a typo here, main.py:50,55)
        config.tr_cl_manifest = "data/chime_tr_org.csv"
    with open(config.labels_path) as label_file:
        labels = str("".join(json.load(label_file)))
    from .dist import DPContext
    dp = DPContext.from_env()       # (collective when world > 1: also creates the host-side gloo group)
    data_loader = DataLoader(batch_size=config.batch_size, paired=paired, tr_cl_manifest=config.tr_cl_manifest,
                             tr_ny_manifest=config.tr_ny_manifest, trsub_manifest=config.trsub_manifest,
                             val_manifest=config.val_manifest, val2_manifest=config.val2_manifest, labels=labels,
                             pin_memory=config.gpu >= 0, preprocess=config.preprocess, dp=dp)
    os.makedirs("logs/" + str(config.expnum), exist_ok=True)
    trainer = Trainer(config, data_loader)
    torch.manual_seed(config.random_seed)
    try:
        if config.mode == "train":
            trainer.train()
        else:
            raise NotImplementedError("mode %r: the reference trainers define neither test() nor visualize()" % config.mode)
    finally:
        if world > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()


if __name__ == "__main__":
    config, unparsed = get_config()
    main(config)
