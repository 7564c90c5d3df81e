"""CPU restatement of the reference training steps (test oracle / CPU baseline only).

  aas_step     <- Speech_enhancement_by_AAS/trainer_AAS.py:131-194  (as executed:
                  E fwd x1 / bwd x2, D fwd x3 / bwd x3, A fwd x1 / bwd x1)
  dce_step     <- trainer_DCE.py:111-127
  fsegan_step  <- trainer_FSEGAN.py:128-182 with the fix list of SURVEY.md 0.13
                  ("intended": DCE term back-propagated; as_written=True keeps the
                  reference's behaviour where G only gets the adversarial gradient)
  am_step      <- AM_training/train.py:297-349
  acoustic_step <- trainer_acoustic.py:120-142 (E + A, loss = CTC / N: no discriminator, no w_acoustic)
  greedy_decoding / greedy_decoding_and_FSEGAN / greedy_decoding_and_AAS and the validation loops around them
               <- trainer_DCE.py:130-190,209-250, trainer_FSEGAN.py:199-243,277-317, trainer_AAS.py:215-263,301-351
                  (pinned by tests/golden/f13_validation.npz)

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


# ---- validation passes (the save_iter blocks of the trainers) -------------------------------------------------------------------
class Meter:
    """utils.py:35-51 AverageMeter."""

    def __init__(self):
        self.sum = self.count = self.avg = 0

    def update(self, val, n=1):
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def greedy_decoding(G, A, labels, inputs, targets, input_percentages, target_sizes):
    """trainer_DCE.py:209-250 (= step 1 of trainer_FSEGAN.py:277-306 and trainer_AAS.py:301-340): enhanced = G(inputs), prob =
    A(enhanced) time-major, sizes = int(pct * T'), greedy strings vs the transcripts; wer = we / total_word and - sic - cer =
    ce / total_word.  -> (wer, cer, total_word, total_char, enhanced, prob, sizes)."""
    from . import decode_np as DN
    split, offset = [], 0
    for size in target_sizes.tolist():
        split.append(targets[offset:offset + size].tolist())
        offset += size
    enhanced = G(inputs)
    prob = A(enhanced).transpose(0, 1)
    sizes = frame_sizes(input_percentages, prob.size(0))
    decoded = DN.greedy_strings(prob.detach().numpy(), sizes.tolist(), labels)
    refs = [DN.labels_to_string(s_, labels) for s_ in split]
    we = ce = total_word = total_char = 0
    for d_, r_ in zip(decoded, refs):
        we += DN.wer(d_, r_)
        ce += DN.cer(d_, r_)
        total_word += len(r_.split())
        total_char += len(r_)
    return we / total_word, ce / total_word, total_word, total_char, enhanced, prob, sizes


def greedy_decoding_and_FSEGAN(G, D, A, labels, mixture, cleans, targets, input_percentages, target_sizes, mask, w_adversarial):
    """trainer_FSEGAN.py:277-317 -> (dce, l_adv_ny, nElement, wer, cer, total_word, total_char); D through forward_paired."""
    wer, cer, nW, nC, enhanced, _, _ = greedy_decoding(G, A, labels, mixture, targets, input_percentages, target_sizes)
    l_adv_ny, nElement = l1loss_mask(D.forward_paired(enhanced, mixture), enhanced, mask.bool())
    l_adv_ny = l_adv_ny * w_adversarial
    dce, nElement_ = l1loss_mask(enhanced, cleans, mask.bool())
    assert nElement == nElement_
    return dce, l_adv_ny, nElement, wer, cer, nW, nC


def greedy_decoding_and_AAS(G, D, A, labels, inputs, targets, input_percentages, target_sizes, mask, w_adversarial, w_acoustic):
    """trainer_AAS.py:301-351 -> (l_CTC, l_adv_ny, nElement, wer, cer, total_word, total_char)."""
    wer, cer, nW, nC, enhanced, prob, sizes = greedy_decoding(G, A, labels, inputs, targets, input_percentages, target_sizes)
    l_adv_ny, nElement = l1loss_mask(D(enhanced), enhanced, mask.bool())
    l_adv_ny = l_adv_ny * w_adversarial
    l_ctc = w_acoustic * ctc_sum(prob, targets, sizes, target_sizes) / inputs.size(0)
    return l_ctc, l_adv_ny, nElement, wer, cer, nW, nC


def dce_validation(G, A, labels, batches):
    """trainer_DCE.py:137-153 / :163-179 over paired-collate batches (inputs, cleans, mask, targets, pct, target_sizes):
    -> dict(dce, wer, cer) of the AverageMeters (dce weighted by nElement, wer by words, cer by characters)."""
    m_dce, m_wer, m_cer = Meter(), Meter(), Meter()
    with torch.no_grad():
        for inputs, cleans, mask, targets, pct, tsz in batches:
            dce, n_el = l1loss_mask(G(inputs), cleans, mask.bool())
            m_dce.update(dce.item(), n_el)
            wer, cer, nW, nC, _, _, _ = greedy_decoding(G, A, labels, inputs, targets, pct, tsz)
            m_wer.update(wer, nW)
            m_cer.update(cer, nC)
    return dict(dce=m_dce.avg, wer=m_wer.avg, cer=m_cer.avg)


def fsegan_validation(G, D, A, labels, batches, w_adversarial):
    """trainer_FSEGAN.py:207-222 / :235-250 -> dict(dce, adv_ny, wer, cer)."""
    m = dict(dce=Meter(), adv_ny=Meter(), wer=Meter(), cer=Meter())
    with torch.no_grad():
        for mixture, cleans, mask, targets, pct, tsz in batches:
            dce, adv, n_el, wer, cer, nW, nC = greedy_decoding_and_FSEGAN(G, D, A, labels, mixture, cleans, targets, pct, tsz, mask, w_adversarial)
            m["dce"].update(dce.item(), n_el); m["adv_ny"].update(adv.item(), n_el)
            m["wer"].update(wer, nW); m["cer"].update(cer, nC)
    return {k: v.avg for k, v in m.items()}
