"""numpy fp64 log-Mel filterbank (LMFB) restatement (test oracle only).  PARITY UNPINNED.

The reference's extractor (AM_training/data/data_loader.py, a fork of
SeanNaren/deepspeech.pytorch's SpectrogramDataset + librosa) is ABSENT from
/root/reference and librosa is not installed here, so there is nothing to pin
against.  Conventions restated from what the reference does show:
  AM_training/train.py:39-42,199  16 kHz, 20 ms hamming window (n_fft = win = 320 ->
                                  161 bins), 10 ms hop (160)
  AM_training/train.py:55-56      n_mels
  Speech_enhancement_by_AAS/model.py:194-198  power = re^2+im^2 -> mel_basis -> log1p
  AM_training/requirements.txt:5  librosa => stft(center=True, reflect pad, periodic
                                  window), filters.mel (Slaney scale, area-normalised,
                                  fmin 0, fmax sr/2)
T = 1 + floor(S / hop).
"""
import numpy as np


def hamming_periodic(n):
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(n) / n)


def _hz_to_mel(f):
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    mel = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mel)


def _mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sr=16000, n_fft=320, n_mels=80, fmin=0.0, fmax=None):
    """Slaney-scale triangular filters, area-normalised -> [n_mels, 1 + n_fft//2] fp64."""
    fmax = sr / 2.0 if fmax is None else fmax
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return w * enorm[:, None]


def lmfb(wave, sr=16000, win=320, hop=160, n_mels=80):
    """wave [S] -> log1p(mel(|STFT|^2)) [n_mels, T], T = 1 + S//hop (center=True, reflect pad)."""
    wave = np.asarray(wave, np.float64)
    pad = win // 2
    y = np.pad(wave, (pad, pad), mode="reflect")
    T = 1 + (len(y) - win) // hop
    idx = np.arange(win)[None, :] + hop * np.arange(T)[:, None]
    frames = y[idx] * hamming_periodic(win)[None, :]
    spec = np.fft.rfft(frames, n=win, axis=1)
    power = spec.real ** 2 + spec.imag ** 2  # [T, 161]
    mel = power @ mel_basis(sr, win, n_mels).T
    return np.log1p(mel).T
