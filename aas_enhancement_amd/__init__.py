"""aas_enhancement_amd: the AAS speech-enhancement training step on MI355X (gfx950).

Python keeps the reference's module / trainer API (model.py, trainer_*.py, config.py, main.py);
every arithmetic op of the hot path is a hand-written HIP kernel behind the C ABI of
``lib/libaas_hip.so`` (include/aas_hip.h).  There is no CPU or eager-PyTorch fallback.
"""
__all__ = ["model", "ops", "ctc", "optim", "prng"]
