"""CPU oracle for the audio front end.  TEST INFRASTRUCTURE ONLY (same rules as conformer_oracle.py).

Parity status: **UNPINNED**.  The reference delegates this arithmetic to torchaudio 2.1.0
(processing/processor.py:10,53-63,156; processing/augment.py:3,9-16), which is neither vendored in /root/reference
nor importable in this image, and the reference holds no fixture for it.  This file restates the PUBLISHED semantics
of torchaudio.transforms.MelSpectrogram / functional.melscale_fbanks / transforms.SpecAugment (as of 2.1.0) with
torch.stft; tests cross-check it against an independent float64 numpy DFT written here.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np
import torch

SAMPLE_RATE, N_FFT, WIN, HOP, N_MELS, F_MIN, F_MAX = 16000, 400, 400, 160, 80, 0.0, 8000.0   # processor.py:18-25
LOG_FLOOR = 1e-5                                                                            # processor.py:157


def _hz_to_mel_slaney(f: float) -> float:
    f_sp = 200.0 / 3
    if f >= 1000.0:
        return 1000.0 / f_sp + math.log(f / 1000.0) / (math.log(6.4) / 27.0)
    return f / f_sp


def _mel_to_hz_slaney(m: torch.Tensor) -> torch.Tensor:
    f_sp = 200.0 / 3
    min_log_mel = 1000.0 / f_sp
    logstep = math.log(6.4) / 27.0
    lin = f_sp * m
    log = 1000.0 * torch.exp(logstep * (m - min_log_mel))
    return torch.where(m >= min_log_mel, log, lin)


def mel_filterbank(n_freqs: int = N_FFT // 2 + 1, n_mels: int = N_MELS, sample_rate: int = SAMPLE_RATE,
                   f_min: float = F_MIN, f_max: float = F_MAX, dtype=torch.float32) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(norm='slaney', mel_scale='slaney'): (n_freqs, n_mels)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=dtype)
    m_pts = torch.linspace(_hz_to_mel_slaney(f_min), _hz_to_mel_slaney(f_max), n_mels + 2, dtype=dtype)
    f_pts = _mel_to_hz_slaney(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.minimum(down, up), min=0.0)
    enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
    return fb * enorm.unsqueeze(0)


def log_mel(wave: torch.Tensor) -> torch.Tensor:
    """ConformerProcessor.mel_spectrogram (processor.py:155-158): (B, L) -> (B, 80, L//160 + 1)."""
    win = torch.hann_window(WIN, periodic=True, dtype=wave.dtype)
    spec = torch.stft(wave, N_FFT, hop_length=HOP, win_length=WIN, window=win, center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    power = spec.abs().pow(2.0)                                        # (B, 201, T)
    mel = torch.matmul(power.transpose(-1, -2), mel_filterbank(dtype=wave.dtype)).transpose(-1, -2)
    return torch.log(torch.clamp(mel, min=LOG_FLOOR))


def log_mel_numpy64(wave: np.ndarray) -> np.ndarray:
    """Independent float64 restatement (explicit reflect padding, explicit DFT matrix) used to cross-check log_mel."""
    wave = np.asarray(wave, dtype=np.float64)
    pad = N_FFT // 2
    xp = np.pad(wave, ((0, 0), (pad, pad)), mode="reflect")
    T = wave.shape[1] // HOP + 1
    n = np.arange(N_FFT)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * n / WIN)                     # periodic Hann
    k = np.arange(N_FFT // 2 + 1)[:, None]
    basis = np.exp(-2j * np.pi * k * n[None, :] / N_FFT) * win[None, :]
    frames = np.stack([xp[:, t * HOP:t * HOP + N_FFT] for t in range(T)], axis=1)   # (B, T, 400)
    power = np.abs(frames @ basis.T) ** 2                              # (B, T, 201)
    fb = mel_filterbank(dtype=torch.float64).numpy()
    return np.log(np.maximum(power @ fb, LOG_FLOOR)).transpose(0, 2, 1)


def batch_lengths(sample_lengths: Sequence[int]) -> List[int]:
    """processor.py:392: frames = samples // hop + 1."""
    return [n // HOP + 1 for n in sample_lengths]


def specaugment_bands(n_frames: int, n_freq: int, n_time_masks: int, time_mask_param: int, n_freq_masks: int,
                      freq_mask_param: int, p: float, generator: torch.Generator) -> List[Tuple[int, int, int]]:
    """The (axis, start, end) bands torchaudio.transforms.SpecAugment(iid_masks=False) would draw: for every mask,
    value = rand*mask_param, min_value = rand*(size - value), band = [long(min_value), long(min_value)+long(value));
    with p < 1 the mask_param is capped at int(size*p) (functional._get_mask_param); time masks first, then
    frequency masks (augment.py:9-16).  axis: 2 = time, 1 = frequency."""
    bands = []
    for axis, size, n, param in ((2, n_frames, n_time_masks, time_mask_param), (1, n_freq, n_freq_masks, freq_mask_param)):
        mp = param if p == 1.0 else min(param, int(size * p))
        for _ in range(n):
            if mp < 1:
                continue
            value = torch.rand(1, generator=generator) * mp
            min_value = torch.rand(1, generator=generator) * (size - value)
            s = int(min_value.long())
            bands.append((axis, s, s + int(value.long())))
    return bands


def specaugment_apply(spec: torch.Tensor, bands, value: float = 0.0) -> torch.Tensor:
    out = spec.clone()
    for axis, s, e in bands:
        if axis == 2:
            out[..., s:e] = value
        else:
            out[..., s:e, :] = value
    return out
