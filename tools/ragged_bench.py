"""Ragged noisy / clean pair at config 2 (noisy T = 200, clean T = Tc): the two-lane schedule against the batched-D schedule with two
row classes in D's recurrent launches (knobs.RAGGED_BATCHED), frozen and trainable A.  Usage: python tools/ragged_bench.py [Tc ...]"""
import sys
import types

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from aas_enhancement_amd import knobs  # noqa: E402
from aas_enhancement_amd.trainer_AAS import Trainer  # noqa: E402


def cfg(**kw):
    c = types.SimpleNamespace(lr=1e-5, beta1=0.5, beta2=0.999, optimizer="adam", batch_size=bench.N_PER, expnum=0, lambda_k=0.001, gamma=0.5,
                              gpu=0, load_path="", mode="train", write_log=False, w_adversarial=1.0, w_acoustic=1.0,
                              allow_ASR_update_iter=10 ** 9, schedule="fused", world_size=1, rank=0)
    c.__dict__.update(kw)
    return c


def main():
    dev = torch.device("cuda:0")
    ny, cl = bench.make_batches(0, dev)
    barrier = torch.cuda.synchronize
    for tc in [int(v) for v in sys.argv[1:]] or [184, 200]:
        cl_r = (cl[0][:, :, :tc].contiguous(), None, None, None, torch.zeros(bench.N_PER, 1, tc, dtype=torch.uint8, device=dev))
        cl_r[4].n_valid = bench.N_PER * tc
        for trainable in (False, True):
            for batched in (False, True):
                tr = Trainer(cfg(allow_ASR_update_iter=0) if trainable else cfg(), None, models=bench.build_models())
                tr.kt = 0.3
                with knobs.override(RAGGED_BATCHED=batched):
                    dt, _ = bench.time_steps(lambda it: tr.train_step_async(ny, cl_r, it), 12, 20, barrier, first=1)
                r = tr.read_scalars()
                print("Tc=%d %s A, RAGGED_BATCHED=%d (%s): %.2f ms/step  l_adv_ny_G %.6f l_adv_cl %.6f l_ctc %.5f kt %.6f" % (
                    tc, "trainable" if trainable else "frozen", batched, tr._last_schedule, 1e3 * dt / 20, r["l_adv_ny_G"], r["l_adv_cl"], r["l_ctc"], r["kt"]),
                    flush=True)
                del tr


if __name__ == "__main__":
    main()
