"""Audio front end on gfx950 kernels: log-mel features and SpecAugment (reference surface: the audio half of
processing/processor.py -- ConformerProcessor.mel_spectrogram / __call__, processor.py:155-158,373-394 -- and
processing/augment.py:7-19).  The text/tokenizer half of ConformerProcessor is CPU string processing and out of scope.

MelSpectrogram(16 kHz, n_fft 400, win 400 periodic Hann, hop 160, centre reflect padding, power 2, 80 slaney/slaney mel
bins 0-8000 Hz) -> log(clamp(., 1e-5)):
    cfm_reflect_pad_f32 -> ONE batched MFMA GEMM (frames are overlapping rows of the padded wave, lda = hop; the window is
    folded into the (416, 400) [cos; 0; -sin; 0] basis) -> cfm_power_mel_log_mfma_f32 (mel products on the matrix pipe) writes
    (B, 80, T) directly.
The DFT basis and the mel filterbank are built once on the host in float64 (they are constants of the configuration).
Parity: torchaudio is not available to pin against -> "parity unpinned" (DESIGN.md); tests compare with an fp64 DFT.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib, ops


def _slaney_filterbank(n_freqs: int, n_mels: int, sample_rate: int, f_min: float, f_max: float) -> torch.Tensor:
    f_sp = 200.0 / 3
    min_log_mel = 1000.0 / f_sp
    logstep = math.log(6.4) / 27.0

    def hz2mel(f):
        return min_log_mel + math.log(f / 1000.0) / logstep if f >= 1000.0 else f / f_sp

    m = torch.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2, dtype=torch.float64)
    f_pts = torch.where(m >= min_log_mel, 1000.0 * torch.exp(logstep * (m - min_log_mel)), f_sp * m)
    freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - freqs[:, None]
    fb = torch.clamp(torch.minimum(-slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]), min=0.0)
    return fb * (2.0 / (f_pts[2:] - f_pts[:-2]))[None, :]


class ConformerAudioFrontend:
    """Mirror of the audio API of ConformerProcessor (processor.py:17-63,155-158,373-394)."""

    def __init__(self, sample_rate: int = 16000, n_fft: int = 400, win_length: int = 400, hop_length: int = 160,
                 n_mels: int = 80, fmin: float = 0.0, fmax: float = 8000.0, device="cuda:0") -> None:
        if n_fft != 400 or win_length != n_fft:
            raise NotImplementedError("the gfx950 log-mel kernel is built for n_fft = win_length = 400 (processor.py:19-20)")
        self.sample_rate, self.n_fft, self.hop_length, self.n_mels = sample_rate, n_fft, hop_length, n_mels
        self.device = torch.device(device)
        n = torch.arange(n_fft, dtype=torch.float64)
        win = 0.5 - 0.5 * torch.cos(2 * math.pi * n / win_length)                    # periodic Hann
        k = torch.arange(n_fft // 2 + 1, dtype=torch.float64)[:, None]
        ang = 2 * math.pi * k * n[None, :] / n_fft
        nb = n_fft // 2 + 1
        self.nk = (nb + 7) // 8 * 8                                                  # 208: Re / Im blocks start 16-byte aligned
        basis = torch.zeros(2 * self.nk, n_fft, dtype=torch.float64)                 # (416, 400): [cos | 0 | -sin | 0] x window
        basis[:nb] = torch.cos(ang) * win
        basis[self.nk:self.nk + nb] = -torch.sin(ang) * win
        self.basis = basis.to(torch.float32).to(self.device).contiguous()
        fb = _slaney_filterbank(nb, n_mels, sample_rate, fmin, fmax)                 # (201, 80)
        self.fb = fb.to(torch.float32).to(self.device).contiguous()
        if n_mels > 96:
            raise NotImplementedError("the matrix-pipe mel kernel holds at most 96 mel bins")
        fbT = torch.zeros(96, self.nk, dtype=torch.float32)                          # transposed, zero-padded: the MFMA A operand
        fbT[:n_mels, :nb] = fb.to(torch.float32).t()
        self.fbT = fbT.to(self.device).contiguous()

    def mel_spectrogram(self, signal: torch.Tensor) -> torch.Tensor:
        """(B, L) fp32 on the HIP device -> log-mel (B, n_mels, L // hop + 1)   (processor.py:155-158)."""
        if signal.dim() == 1:
            signal = signal[None]
        x = ops._req(signal, "signal")
        B, L = x.shape
        pad = self.n_fft // 2
        if L <= pad:
            raise _lib.ConformerHipError(f"signal of {L} samples is shorter than the reflect padding ({pad})")
        T = L // self.hop_length + 1
        hop = self.hop_length
        rpu = T + (self.n_fft + hop - 1) // hop                       # rows per utterance: pitch rpu * hop >= L + n_fft
        Lp = rpu * hop
        lib = _lib.load()
        # (B, Lp) padded waves back to back + n_fft floats of slack: the junk frames behind the last utterance read into it (rows are
        # independent: whatever those floats hold only reaches spectrum rows nobody reads)
        xbuf = torch.empty(B * Lp + self.n_fft, device=x.device, dtype=torch.float32)
        xp = xbuf[:B * Lp].view(B, Lp)
        _lib.check(lib.cfm_reflect_pad_f32(x.data_ptr(), xp.data_ptr(), B, L, pad, Lp, ops._stream()), "cfm_reflect_pad_f32")
        nb2 = self.basis.shape[0]                                                    # 2 * nk = 416 DFT columns (Re | Im, zero-padded)
        spec = torch.empty(B * rpu, nb2, device=x.device, dtype=torch.float32)
        if hop % 4 == 0:
            # frames = overlapping rows of the padded waves (row b * rpu + t); the tuned forward GEMM with lda = hop
            _lib.check(lib.cfm_dft_frames_f32(xp.data_ptr(), self.basis.data_ptr(), spec.data_ptr(), B * rpu, nb2, self.n_fft, hop,
                                              ops._stream()), "cfm_dft_frames_f32")
        else:
            ops.gemm_bwd(xp, False, self.basis, False, T, nb2, self.n_fft, out=spec, lda=hop, ldb=self.n_fft,
                         ldc=nb2, nbatch=B, nb1=1, sa=(Lp, 0), sb=(0, 0), sc=(rpu * nb2, 0))
        out = torch.empty(B, self.n_mels, T, device=x.device, dtype=torch.float32)
        _lib.check(lib.cfm_power_mel_log_mfma_f32(spec.data_ptr(), nb2, self.fbT.data_ptr(), out.data_ptr(), B, T, rpu,
                                                  self.n_fft // 2 + 1, self.n_mels, 1e-5, ops._stream()), "cfm_power_mel_log_mfma_f32")
        return out

    def __call__(self, audios: Sequence[torch.Tensor], augment: Optional["ConformerAugment"] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """processor.py:373-394: zero-pad to the longest, log-mel, lengths = samples // hop + 1."""
        lengths = [int(a.numel()) for a in audios]
        batch = torch.zeros(len(audios), max(lengths), device=self.device, dtype=torch.float32)
        for i, a in enumerate(audios):
            batch[i, :lengths[i]] = a.to(self.device)
        mels = self.mel_spectrogram(batch)
        if augment is not None:
            mels = augment(mels)
        return mels, torch.tensor(lengths, device=self.device) // self.hop_length + 1


class ConformerAugment:
    """processing/augment.py:7-19: SpecAugment(n_time_masks, time_mask_param, n_freq_masks, freq_mask_param, p=ratio,
    zero_masking), one mask set shared by the whole batch (iid_masks=False).  Band positions are drawn on the host with
    torch.rand (where torchaudio draws them); the kernel fills the bands in place."""

    def __init__(self, n_time_masks: int = 2, time_mask_param: int = 100, n_freq_masks: int = 2, freq_mask_param: int = 27,
                 ratio: float = 1, zero_masking: bool = True, device="cuda:0") -> None:
        self.n_time_masks, self.time_mask_param = n_time_masks, time_mask_param
        self.n_freq_masks, self.freq_mask_param = n_freq_masks, freq_mask_param
        self.p, self.zero_masking = float(ratio), zero_masking
        self.generator: Optional[torch.Generator] = None            # None = torch's default CPU generator

    def draw_bands(self, n_freq: int, n_frames: int) -> List[Tuple[int, int, int]]:
        bands = []
        for axis, size, n, param in ((2, n_frames, self.n_time_masks, self.time_mask_param),
                                     (1, n_freq, self.n_freq_masks, self.freq_mask_param)):
            mp = param if self.p == 1.0 else min(param, int(size * self.p))
            for _ in range(n):
                if mp < 1:
                    continue
                value = torch.rand(1, generator=self.generator) * mp
                min_value = torch.rand(1, generator=self.generator) * (size - value)
                s = int(min_value.long())
                bands.append((axis, s, s + int(value.long())))
        return bands

    def __call__(self, mels: torch.Tensor) -> torch.Tensor:
        x = ops._req(mels, "mels")
        B, F, T = x.shape
        bands = self.draw_bands(F, T)
        if not bands:
            return x
        if self.zero_masking:
            value = 0.0
        else:
            value = float(x.mean())                                  # torchaudio: mask_value = specgram.mean() (one sync)
        bt = torch.tensor(bands, dtype=torch.int32, device=x.device)
        _lib.check(_lib.load().cfm_specaugment_apply_f32(x.data_ptr(), B, F, T, bt.data_ptr(), len(bands), value,
                                                         ops._stream()), "cfm_specaugment_apply_f32")
        return x
