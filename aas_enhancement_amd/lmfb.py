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


class LMFB(nn.Module):
    def __init__(self, sample_rate=16000, window_size=0.02, window_stride=0.01, n_mels=80):
        super().__init__()
        self.win = int(round(sample_rate * window_size))
        self.hop = int(round(sample_rate * window_stride))
        self.n_mels = n_mels
        self.register_buffer("dft", torch.from_numpy(dft_table(self.win).astype(np.float32)), persistent=False)
        self.register_buffer("melT", torch.from_numpy(mel_filterbank(sample_rate, self.win, n_mels).T.copy().astype(np.float32)), persistent=False)

    def forward(self, wave):  # [N,S] -> [N,n_mels,T]
        require_cuda(wave, self.dft)
        wave = wave.contiguous().float()
        N, S = wave.shape
        T = 1 + S // self.hop
        out = torch.empty((N, self.n_mels, T), device=wave.device, dtype=torch.float32)
        check(lib().aas_lmfb_fwd(stream(), ptr(wave), N, S, self.win, self.hop, self.n_mels, ptr(self.dft), ptr(self.melT), ptr(out)), "aas_lmfb_fwd")
        return out
