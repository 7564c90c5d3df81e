"""log-Mel filterbank front-end on the HIP LMFB kernel.

Conventions (the reference's extractor source is absent - SURVEY.md 0.10; restated from
AM_training/train.py:39-42,55-60,199 and Speech_enhancement_by_AAS/model.py:194-198): 16 kHz, 320-sample
periodic hamming window, hop 160, n_fft 320 (161 bins), centre=True reflect padding, power spectrum,
Slaney area-normalised mel (fmin 0, fmax sr/2), log1p; T = 1 + S // hop; no CMVN.
The DFT and mel tables are host-built constants (fp64 -> fp32), uploaded once per device.
"""
import numpy as np
import torch
import torch.nn as nn

from ._lib import check, lib, ptr, require_cuda, stream


def _hz_to_mel(f):
    f = np.asarray(f, np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels):
    """[n_mels, n_fft//2+1] triangular Slaney filters, area normalised."""
    freqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    w = np.zeros((n_mels, len(freqs)))
    for i in range(n_mels):
        up = (freqs - pts[i]) / (pts[i + 1] - pts[i])
        down = (pts[i + 2] - freqs) / (pts[i + 2] - pts[i + 1])
        w[i] = np.maximum(0.0, np.minimum(up, down)) * (2.0 / (pts[i + 2] - pts[i]))
    return w


def dft_table(win):
    """[win, 2*nbins]: hamming(periodic)[j] * cos(2 pi j b / win) | -hamming[j] * sin(...)"""
    j = np.arange(win)[:, None].astype(np.float64)
    b = np.arange(win // 2 + 1)[None, :].astype(np.float64)
    ham = 0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(win) / win)
    ang = 2.0 * np.pi * j * b / win
    return np.concatenate([ham[:, None] * np.cos(ang), -ham[:, None] * np.sin(ang)], axis=1)


def folded_dft_tables():
    """bf16 hi | lo twiddles of the twice-folded 320-point real DFT (csrc/lmfb320.hip): uint16 [2, 4, 96, 96] =
    [plane][segment][column c][k = j]; segment 0: cos(2 pi j 2c / 320), j <= 80, c <= 80; 1: cos(2 pi j (2c+1) / 320),
    j <= 79, c <= 79; 2: sin(2 pi j 2c / 320), 1 <= j <= 79; 3: sin(2 pi j (2c+1) / 320), 1 <= j <= 80; zero elsewhere."""
    tab = np.zeros((4, 96, 96), np.float64)
    c = np.arange(96)[:, None].astype(np.float64)
    k = np.arange(96)[None, :].astype(np.float64)
    w = 2.0 * np.pi / 320.0
    tab[0] = np.where((c <= 80) & (k <= 80), np.cos(w * k * 2 * c), 0.0)
    tab[1] = np.where((c <= 79) & (k <= 79), np.cos(w * k * (2 * c + 1)), 0.0)
    tab[2] = np.where((c <= 80) & (k >= 1) & (k <= 79), np.sin(w * k * 2 * c), 0.0)
    tab[3] = np.where((c <= 79) & (k >= 1) & (k <= 80), np.sin(w * k * (2 * c + 1)), 0.0)
    t32 = torch.from_numpy(tab.astype(np.float32))
    hi = t32.to(torch.bfloat16)
    lo = (t32 - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo]).contiguous()      # bf16 [2,4,96,96]


def sparse_mel(sr, n_fft, n_mels, maxw=24):
    """Triangular filters as (first bin, width, weights[n_mels, maxw]); None if a filter is wider than maxw bins."""
    fb = mel_filterbank(sr, n_fft, n_mels)
    start, cnt, w = np.zeros(n_mels, np.int32), np.zeros(n_mels, np.int32), np.zeros((n_mels, maxw), np.float32)
    for m in range(n_mels):
        nz = np.nonzero(fb[m])[0]
        if len(nz) == 0:
            continue
        lo, hi = int(nz[0]), int(nz[-1]) + 1
        if hi - lo > maxw:
            return None
        start[m], cnt[m] = lo, hi - lo
        w[m, :hi - lo] = fb[m, lo:hi]
    return start, cnt, w


class LMFB(nn.Module):
    def __init__(self, sample_rate=16000, window_size=0.02, window_stride=0.01, n_mels=80):
        super().__init__()
        self.win = int(round(sample_rate * window_size))
        self.hop = int(round(sample_rate * window_stride))
        self.n_mels = n_mels
        self.register_buffer("dft", torch.from_numpy(dft_table(self.win).astype(np.float32)), persistent=False)
        self.register_buffer("melT", torch.from_numpy(mel_filterbank(sample_rate, self.win, n_mels).T.copy().astype(np.float32)), persistent=False)
        # matrix-core path (csrc/lmfb320.hip) for the reference's 320 / 160 framing
        sp = sparse_mel(sample_rate, self.win, n_mels) if (self.win == 320 and self.hop == 160) else None
        self.fast = sp is not None
        if self.fast:
            ham = 0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(320) / 320.0)
            self.register_buffer("tab", folded_dft_tables(), persistent=False)
            self.register_buffer("window", torch.from_numpy(ham.astype(np.float32)), persistent=False)
            self.register_buffer("mel_start", torch.from_numpy(sp[0]), persistent=False)
            self.register_buffer("mel_cnt", torch.from_numpy(sp[1]), persistent=False)
            self.register_buffer("mel_w", torch.from_numpy(sp[2]), persistent=False)

    def forward(self, wave, lens=None, force_scalar=False):
        """wave [N,S] (zero padded) -> [N,n_mels,T], T = 1 + S // hop.  `lens` [N] int32 device tensor: valid samples per
        utterance (reflect padding at each utterance's own end, zero frames beyond its 1 + len // hop) - matrix-core path only."""
        require_cuda(wave, self.dft)
        wave = wave.contiguous().float()
        N, S = wave.shape
        T = 1 + S // self.hop
        out = torch.empty((N, self.n_mels, T), device=wave.device, dtype=torch.float32)
        if self.fast and not force_scalar:
            if lens is not None:
                lens = lens.to(device=wave.device, dtype=torch.int32).contiguous()
            check(lib().aas_lmfb320_fwd(stream(), ptr(wave), ptr(lens), N, S, self.n_mels, ptr(self.tab), ptr(self.window), ptr(self.mel_start),
                                        ptr(self.mel_cnt), ptr(self.mel_w), int(self.mel_w.shape[1]), ptr(out)), "aas_lmfb320_fwd")
            return out
        if lens is not None:
            raise NotImplementedError("per-utterance lengths need the 320/160 matrix-core LMFB path")
        check(lib().aas_lmfb_fwd(stream(), ptr(wave), N, S, self.win, self.hop, self.n_mels, ptr(self.dft), ptr(self.melT), ptr(out)), "aas_lmfb_fwd")
        return out
