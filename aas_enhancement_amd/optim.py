"""Adam / Adam-amsgrad with the element-wise update fused into one HIP kernel per parameter tensor.

Replaces ``torch.optim.Adam(..., amsgrad=True)`` at Speech_enhancement_by_AAS/trainer_AAS.py:127-129
(and plain Adam at AM_training/train.py:246-247).  torch 2.x update rule.
"""
import torch

from . import ops


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, amsgrad=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, amsgrad=amsgrad))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                    if group["amsgrad"]:
                        st["max_exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adam_step(p, g, st["exp_avg"], st["exp_avg_sq"], st.get("max_exp_avg_sq"), group["lr"], b1, b2,
                              group["eps"], st["step"], group["amsgrad"], grad_scale)


class FlatAdam(object):
    """Adam(-amsgrad) over a FlatBuffers pair: a single fused HIP launch per network per step."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, amsgrad=False):
        self.flat, self.lr, self.betas, self.eps, self.amsgrad = flat, lr, betas, eps, amsgrad
        self.step_count = 0
        self.m = torch.zeros_like(flat.flat_p)
        self.v = torch.zeros_like(flat.flat_p)
        self.vmax = torch.zeros_like(flat.flat_p) if amsgrad else None

    def zero_grad(self):
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        self.step_count += 1
        if getattr(self, "_t_dev", None) is not None:
            self._t_dev.fill_(float(self.step_count))   # keep the device counter of step_dev() in step (paths may alternate)
        ops.adam_step(self.flat.flat_p, self.flat.flat_g, self.m, self.v, self.vmax, self.lr, self.betas[0],
                      self.betas[1], self.eps, self.step_count, self.amsgrad, grad_scale)
        self.flat.version += 1

    # ---- torch.optim.Adam-shaped state (the `optim_dict` of a DeepSpeech package, model.py:393-394 / train.py:170) -------
    def state_dict(self):
        state = {}
        for i, (p, (off, n)) in enumerate(zip(self.flat.params, self.flat.slices)):
            st = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.m[off:off + n].view_as(p).clone(),
                  "exp_avg_sq": self.v[off:off + n].view_as(p).clone()}
            if self.vmax is not None:
                st["max_exp_avg_sq"] = self.vmax[off:off + n].view_as(p).clone()
            state[i] = st
        group = dict(lr=self.lr, betas=tuple(self.betas), eps=self.eps, weight_decay=0, amsgrad=self.amsgrad,
                     params=list(range(len(self.flat.params))))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        for i, (off, n) in enumerate(self.flat.slices):
            st = sd["state"].get(i)
            if st is None:
                continue
            self.m[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            if self.vmax is not None and "max_exp_avg_sq" in st:
                self.vmax[off:off + n].copy_(st["max_exp_avg_sq"].reshape(-1))
            self.step_count = int(float(st["step"]))
        if sd.get("param_groups"):
            self.lr = sd["param_groups"][0].get("lr", self.lr)
        if getattr(self, "_t_dev", None) is not None:
            self._t_dev.fill_(float(self.step_count))

    # ---- device-resident step counter (hipGraph capture: no host-side scalars change between replays) -------
    @torch.no_grad()
    def step_dev(self, grad_scale=1.0):
        """Same update; the bias-correction scalars are computed ON the device from a device step counter, so the
        launch sequence is identical every step and can be replayed from a captured graph."""
        if getattr(self, "_t_dev", None) is None:
            dev = self.flat.flat_p.device
            self._t_dev = torch.full((1,), float(self.step_count), device=dev, dtype=torch.float64)
            self._hyper = torch.zeros(2, device=dev, dtype=torch.float32)
        ops.adam_tick(self._t_dev, self.lr, self.betas[0], self.betas[1], self._hyper)   # t += 1, bias corrections: one tiny launch
        ops.adam_step_dev(self.flat.flat_p, self.flat.flat_g, self.m, self.v, self.vmax, self.betas[0], self.betas[1],
                          self.eps, self._hyper, self.amsgrad, grad_scale)
        self.step_count += 1
        self.flat.version += 1


class FlatSGD(object):
    """torch.optim.SGD(lr, momentum, nesterov=True) over a FlatBuffers pair (AM_training/train.py:172-174,247-249, `--optim sgd`):
    one fused HIP launch per step.  The update has no step-dependent scalar, so `step` and `step_dev` are the same launch."""

    def __init__(self, flat, lr=1e-3, momentum=0.9):
        self.flat, self.lr, self.momentum = flat, lr, momentum
        self.step_count = 0
        self.buf = torch.zeros_like(flat.flat_p)

    def zero_grad(self):
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        ops.sgd_nesterov_step(self.flat.flat_p, self.flat.flat_g, self.buf, self.lr, self.momentum, grad_scale)
        self.step_count += 1
        self.flat.version += 1

    step_dev = step

    # ---- torch.optim.SGD-shaped state (the `optim_dict` of a DeepSpeech package) -------
    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (p, (off, n)) in enumerate(zip(self.flat.params, self.flat.slices)):
                state[i] = {"momentum_buffer": self.buf[off:off + n].view_as(p).clone()}
        group = dict(lr=self.lr, momentum=self.momentum, dampening=0, weight_decay=0, nesterov=True, params=list(range(len(self.flat.params))))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        for i, (off, n) in enumerate(self.flat.slices):
            st = sd["state"].get(i)
            if st is not None and st.get("momentum_buffer") is not None:
                self.buf[off:off + n].copy_(st["momentum_buffer"].reshape(-1))
                self.step_count = max(self.step_count, 1)
        if sd.get("param_groups"):
            self.lr = sd["param_groups"][0].get("lr", self.lr)
            self.momentum = sd["param_groups"][0].get("momentum", self.momentum)
