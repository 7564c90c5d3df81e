"""Independent numpy fp64 CTC (test oracle only).

The reference calls the third-party ``warpctc_pytorch.CTCLoss`` (SeanNaren binding of
baidu-research/warp-ctc; not vendored, no version pinned - reference README.md:8; call
sites Speech_enhancement_by_AAS/trainer_AAS.py:10,62,168,349).  This restates the
published algorithm (Graves et al. 2006, as implemented by warp-ctc): softmax over the
alphabet, alpha/beta recursions in log space over the blank-extended label sequence of
S = 2L+1 states (blank index 0), cost = -log p(l|x), gradient wrt the PRE-softmax
activations = softmax - (1/p) * sum_{s: l'_s = k} alpha_t(s) beta_t(s), zero for t >= act_len.
"""
import itertools

import numpy as np

NEG_INF = -np.inf


def _logsumexp2(a, b):
    if a == NEG_INF:
        return b
    if b == NEG_INF:
        return a
    m = max(a, b)
    return m + np.log(np.exp(a - m) + np.exp(b - m))


def log_softmax(x):
    m = x.max(axis=-1, keepdims=True)
    z = x - m
    return z - np.log(np.exp(z).sum(axis=-1, keepdims=True))


def ctc_one(acts_tc, labels, blank=0):
    """acts_tc [T,C] pre-softmax (already cut to act_len), labels list[int].
    Returns (cost, grad[T,C])."""
    acts_tc = np.asarray(acts_tc, np.float64)
    T, C = acts_tc.shape
    L = len(labels)
    S = 2 * L + 1
    ext = [blank] * S
    for i, l in enumerate(labels):
        ext[2 * i + 1] = int(l)
    lp = log_softmax(acts_tc)
    alpha = np.full((T, S), NEG_INF)
    beta = np.full((T, S), NEG_INF)
    if T == 0 or T < L:
        return np.inf, np.zeros_like(acts_tc)
    alpha[0, 0] = lp[0, ext[0]]
    if S > 1:
        alpha[0, 1] = lp[0, ext[1]]
    for t in range(1, T):
        for s in range(S):
            a = alpha[t - 1, s]
            if s >= 1:
                a = _logsumexp2(a, alpha[t - 1, s - 1])
            if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
                a = _logsumexp2(a, alpha[t - 1, s - 2])
            alpha[t, s] = a + lp[t, ext[s]] if a != NEG_INF else NEG_INF
    beta[T - 1, S - 1] = lp[T - 1, ext[S - 1]]
    if S > 1:
        beta[T - 1, S - 2] = lp[T - 1, ext[S - 2]]
    for t in range(T - 2, -1, -1):
        for s in range(S):
            b = beta[t + 1, s]
            if s + 1 < S:
                b = _logsumexp2(b, beta[t + 1, s + 1])
            if s + 2 < S and ext[s + 2] != blank and ext[s + 2] != ext[s]:
                b = _logsumexp2(b, beta[t + 1, s + 2])
            beta[t, s] = b + lp[t, ext[s]] if b != NEG_INF else NEG_INF
    ll = alpha[T - 1, S - 1]
    if S > 1:
        ll = _logsumexp2(ll, alpha[T - 1, S - 2])
    if ll == NEG_INF:
        return np.inf, np.zeros_like(acts_tc)
    grad = np.exp(lp)
    for t in range(T):
        acc = np.full(C, NEG_INF)
        for s in range(S):
            ab = alpha[t, s] + beta[t, s]
            if ab != NEG_INF:
                acc[ext[s]] = _logsumexp2(acc[ext[s]], ab)
        for k in range(C):
            if acc[k] != NEG_INF:
                # alpha*beta counts y_t(l'_s) twice -> divide once by y
                grad[t, k] -= np.exp(acc[k] - lp[t, k] - ll)
    return -ll, grad


def ctc_batch(acts_tnc, flat_labels, act_lens, label_lens, blank=0):
    """warp-ctc call shape: acts [T,N,C], flat labels, per-utterance lengths.
    Returns (costs[N], grads[T,N,C])."""
    acts_tnc = np.asarray(acts_tnc, np.float64)
    T, N, C = acts_tnc.shape
    costs = np.zeros(N)
    grads = np.zeros_like(acts_tnc)
    off = 0
    for n in range(N):
        L = int(label_lens[n])
        tl = int(act_lens[n])
        c, g = ctc_one(acts_tnc[:tl, n], list(flat_labels[off:off + L]), blank)
        off += L
        costs[n] = c
        grads[:tl, n] = g
    return costs, grads


def ctc_bruteforce(acts_tc, labels, blank=0):
    """-log sum over ALL alignments pi with collapse(pi)==labels of prod_t softmax(acts)[t,pi_t].
    Exponential; for T<=6, C<=4 known-answer pins."""
    acts_tc = np.asarray(acts_tc, np.float64)
    T, C = acts_tc.shape
    p = np.exp(log_softmax(acts_tc))
    total = 0.0
    for path in itertools.product(range(C), repeat=T):
        out, prev = [], None
        for k in path:
            if k != prev and k != blank:
                out.append(k)
            prev = k
        if out == list(labels):
            pr = 1.0
            for t, k in enumerate(path):
                pr *= p[t, k]
            total += pr
    return -np.log(total) if total > 0 else np.inf
