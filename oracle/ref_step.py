"""CPU restatement of the reference training steps (test oracle / CPU baseline only).

  aas_step     <- Speech_enhancement_by_AAS/trainer_AAS.py:131-194  (as executed:
                  E fwd x1 / bwd x2, D fwd x3 / bwd x3, A fwd x1 / bwd x1)
  dce_step     <- trainer_DCE.py:111-127
  fsegan_step  <- trainer_FSEGAN.py:128-182 with the fix list of SURVEY.md 0.13
                  ("intended": DCE term back-propagated; as_written=True keeps the
                  reference's behaviour where G only gets the adversarial gradient)
  am_step      <- AM_training/train.py:297-349
  acoustic_step <- trainer_acoustic.py:120-142 (E + A, loss = CTC / N: no discriminator, no w_acoustic)

Substitutions vs the reference source: .data[0] -> .item(); no
.cuda(); bool mask; warpctc CTCLoss(prob, ...) -> F.ctc_loss(log_softmax(prob)),
sum reduction, blank 0; A is left in train mode (the reference never calls
ASR.eval()).
"""
import math

import torch
import torch.nn.functional as F

from .ref_model import l1loss_mask


class StepConfig:
    def __init__(self, w_adversarial=1.0, w_acoustic=1.0, gamma=0.5, lambda_k=0.001,
                 allow_ASR_update_iter=0, lr=1e-5, beta1=0.5, beta2=0.999):
        self.w_adversarial = w_adversarial
        self.w_acoustic = w_acoustic
        self.gamma = gamma
        self.lambda_k = lambda_k
        self.allow_ASR_update_iter = allow_ASR_update_iter
        self.lr = lr
        self.beta1 = beta1
        self.beta2 = beta2


def make_optim(model, cfg, amsgrad=True):
    return torch.optim.Adam(model.parameters(), lr=cfg.lr, betas=(cfg.beta1, cfg.beta2), amsgrad=amsgrad)


def ctc_sum(acts_tnc, targets, sizes, target_sizes):
    """warpctc_pytorch.CTCLoss() semantics: sum over utterances of -log p(l|x), softmax inside."""
    return F.ctc_loss(F.log_softmax(acts_tnc, dim=2), targets.long(), sizes.long(), target_sizes.long(),
                      blank=0, reduction="sum", zero_infinity=False)


def grad_norm(model):
    """trainer_AAS.py:353-361"""
    s = 0.0
    for p in model.parameters():
        if p.grad is not None:
            s = s + p.grad.pow(2).sum()
    return float(torch.as_tensor(s).sqrt())


def frame_sizes(input_percentages, t_out):
    """trainer_AAS.py:166-167: sizes = int(pct * T') (float32 multiply, truncation)."""
    return (input_percentages.clone().float().mul_(int(t_out))).int()


def aas_step(G, D, A, opt_g, opt_d, opt_a, noisy, clean, cfg, kt, it):
    """One iteration of trainer_AAS.py:131-194.  noisy = (inputs, targets, pct, target_sizes, mask),
    clean = (inputs, ..., mask) (only [0] and [4] are used, :176).  Returns (kt_new, scalars)."""
    G.zero_grad(); D.zero_grad(); A.zero_grad()
    inputs, targets, pct, target_sizes, mask = noisy
    mask = mask.bool()
    N = inputs.size(0)
    enhanced = G(inputs)
    enhanced_D = enhanced.detach()
    # G-step
    ae = D(enhanced)
    l_g, _ = l1loss_mask(ae, enhanced, mask)
    l_g = l_g * cfg.w_adversarial
    l_adv_ny_G = l_g.item()
    l_g.backward(retain_graph=True)
    g_adv = grad_norm(G)
    D.zero_grad()
    # D-step
    ae_d = D(enhanced_D)
    l_d, _ = l1loss_mask(ae_d, enhanced_D, mask)
    l_d = l_d * (-kt) * cfg.w_adversarial
    l_d.backward()
    # CTC
    prob = A(enhanced).transpose(0, 1)
    sizes = frame_sizes(pct, prob.size(0))
    l_ctc = cfg.w_acoustic * ctc_sum(prob, targets, sizes, target_sizes) / N
    l_ctc_v = l_ctc.item()
    l_ctc.backward()
    g_ctc_adv = grad_norm(G)
    # clean
    cl_in, cl_mask = clean[0], clean[4].bool()
    ae_cl = D(cl_in)
    l_cl, _ = l1loss_mask(ae_cl, cl_in, cl_mask)
    l_cl = cfg.w_adversarial * l_cl
    l_cl.backward()
    l_adv_cl = l_cl.item()
    opt_g.step(); opt_d.step()
    if it > cfg.allow_ASR_update_iter:
        opt_a.step()
    bal = cfg.gamma * l_adv_cl - l_adv_ny_G
    kt = max(min(1.0, kt + cfg.lambda_k * bal), 0.0)
    conv = l_adv_cl + abs(bal)
    return kt, dict(l_adv_ny_G=l_adv_ny_G, l_adv_cl=l_adv_cl, l_ctc=l_ctc_v, g_adv=g_adv,
                    g_ctc_adv=g_ctc_adv, kt=kt, conv_measure=conv,
                    enhanced=enhanced.detach(), logits=prob.detach())


def acoustic_step(G, A, opt_g, opt_a, noisy, cfg, it):
    """One iteration of trainer_acoustic.py:120-142: enhanced = G(x); loss = CTC(A(enhanced)) / N; Adam on G, on A once
    iter > allow_ASR_update_iter.  noisy = (inputs, targets, pct, target_sizes, ...)."""
    inputs, targets, pct, target_sizes = noisy[0], noisy[1], noisy[2], noisy[3]
    N = inputs.size(0)
    enhanced = G(inputs)
    prob = A(enhanced).transpose(0, 1)
    sizes = frame_sizes(pct, prob.size(0))
    loss = ctc_sum(prob, targets, sizes, target_sizes) / N
    G.zero_grad(); A.zero_grad()
    loss.backward()
    opt_g.step()
    if it > cfg.allow_ASR_update_iter:
        opt_a.step()
    return dict(l_ctc=loss.item(), enhanced=enhanced.detach(), logits=prob.detach())


def dce_step(G, opt_g, batch):
    """trainer_DCE.py:116-127; batch = (inputs, cleans, mask, ...) (paired collate order)."""
    inputs, cleans, mask = batch[0], batch[1], batch[2].bool()
    out = G(inputs)
    loss, n_el = l1loss_mask(out, cleans, mask)
    G.zero_grad()
    loss.backward()
    gn = grad_norm(G)
    opt_g.step()
    return dict(loss=loss.item(), nElement=n_el, g_norm=gn, outputs=out.detach())


def fsegan_step(G, D, opt_g, opt_d, batch, cfg, kt, as_written=False):
    """trainer_FSEGAN.py:128-182, intended semantics.  D has I=2F, O=F."""
    mixture, cleans, mask = batch[0], batch[1], batch[2].bool()
    G.zero_grad(); D.zero_grad()
    enhanced = G(mixture)
    enhanced_D = enhanced.detach()
    ae = D.forward_paired(enhanced, mixture)
    l_g, _ = l1loss_mask(ae, enhanced, mask)
    l_g = l_g * cfg.w_adversarial
    l_adv_ny_G = l_g.item()
    l_g.backward(retain_graph=True)
    D.zero_grad()
    ae_d = D.forward_paired(enhanced_D, mixture)
    l_d, _ = l1loss_mask(ae_d, enhanced_D, mask)
    (l_d * (-kt) * cfg.w_adversarial).backward()
    dce, _ = l1loss_mask(enhanced, cleans, mask)
    if not as_written:
        dce.backward()
    ae_cl = D.forward_paired(cleans, mixture)
    l_cl, _ = l1loss_mask(ae_cl, cleans, mask)
    l_cl = cfg.w_adversarial * l_cl
    l_cl.backward()
    l_adv_cl = l_cl.item()
    opt_g.step(); opt_d.step()
    bal = cfg.gamma * l_adv_cl - l_adv_ny_G
    kt = max(min(1.0, kt + cfg.lambda_k * bal), 0.0)
    return kt, dict(l_adv_ny_G=l_adv_ny_G, l_adv_cl=l_adv_cl, dce=dce.item(), kt=kt,
                    conv_measure=l_adv_cl + abs(bal), g_norm=grad_norm(G))


def am_step(A, opt, batch):
    """AM_training/train.py:297-349: A(x) -> CTC/N -> Adam."""
    inputs, targets, pct, target_sizes = batch[0], batch[1], batch[2], batch[3]
    out = A(inputs).transpose(0, 1)
    sizes = frame_sizes(pct, out.size(0))
    loss = ctc_sum(out, targets, sizes, target_sizes) / inputs.size(0)
    opt.zero_grad()
    loss.backward()
    opt.step()
    v = loss.item()
    return dict(loss=v, is_inf=(v == math.inf or v == -math.inf), logits=out.detach())
