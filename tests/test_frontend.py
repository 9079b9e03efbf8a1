"""Front end: (CPU) the torch.stft-based oracle against an independent float64 DFT; (GPU) the HIP log-mel and SpecAugment
against the oracle.  PARITY UNPINNED: torchaudio (where the reference's arithmetic lives) is not available, so these
tests pin the kernels to the published semantics restated in oracle/frontend_oracle.py, not to torchaudio itself."""
import numpy as np
import pytest
import torch

from oracle import frontend_oracle as FO


def _wave(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(L) / 16000.0
    tone = 0.3 * torch.sin(2 * torch.pi * 440.0 * t)[None] + 0.1 * torch.sin(2 * torch.pi * 3100.0 * t)[None]
    return tone + 0.05 * torch.randn(B, L, generator=g)


def test_oracle_matches_float64_dft():
    w = _wave(2, 4000, 1)
    a = FO.log_mel(w)
    b = FO.log_mel_numpy64(w.numpy())
    assert a.shape == (2, 80, 4000 // 160 + 1)
    assert np.abs(a.numpy() - b).max() < 2e-3          # fp32 FFT vs fp64 DFT in the log domain


def test_filterbank_properties():
    fb = FO.mel_filterbank(dtype=torch.float64)
    assert fb.shape == (201, 80) and float(fb.min()) >= 0
    assert (fb.sum(0) > 0).all()                       # no empty filter at 80 mels / 201 bins
    peaks = fb.argmax(0)
    assert (peaks[1:] >= peaks[:-1]).all()             # centre frequencies increase


def test_lengths_and_specaugment_bands():
    assert FO.batch_lengths([16000, 159, 160]) == [101, 1, 2]
    g = torch.Generator().manual_seed(3)
    bands = FO.specaugment_bands(500, 80, 10, 35, 10, 35, 0.05, g)
    assert len(bands) == 20
    for ax, s, e in bands:
        size, cap = (500, min(35, int(500 * 0.05))) if ax == 2 else (80, min(35, int(80 * 0.05)))
        assert 0 <= s <= e <= size and e - s <= cap


@pytest.mark.gpu
@pytest.mark.parametrize("B,L", [(1, 400), (3, 16000), (2, 159840), (2, 7777)])
def test_logmel_gpu_vs_oracle(B, L):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from conformer_amd.frontend import ConformerAudioFrontend
    fe = ConformerAudioFrontend()
    w = _wave(B, L, 2)
    got = fe.mel_spectrogram(w.cuda()).cpu()
    ref = FO.log_mel_numpy64(w.numpy())
    assert got.shape == ref.shape == (B, 80, L // 160 + 1)
    err = np.abs(got.numpy() - ref)
    assert err.max() < 5e-3 and err.mean() < 1e-4       # log domain; fp32 DFT of a 400-sample frame
    # the filterbank constant built by the product equals the oracle's
    assert float((fe.fb.cpu() - FO.mel_filterbank()).abs().max()) < 1e-6


@pytest.mark.gpu
def test_frontend_call_and_specaugment_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from conformer_amd.frontend import ConformerAudioFrontend, ConformerAugment
    fe = ConformerAudioFrontend()
    audios = [_wave(1, n, i)[0] for i, n in enumerate((16000, 12345, 4000))]
    mels, lengths = fe(audios)
    assert mels.shape == (3, 80, 101) and lengths.tolist() == [101, 78, 26]
    ref0 = FO.log_mel(torch.nn.functional.pad(audios[1], (0, 16000 - 12345))[None])[0]
    assert float((mels[1].cpu() - ref0).abs().max()) < 5e-3
    aug = ConformerAugment(n_time_masks=10, time_mask_param=35, n_freq_masks=10, freq_mask_param=35, ratio=0.05)
    aug.generator = torch.Generator().manual_seed(7)
    before = mels.clone()
    out = aug(mels)
    bands = FO.specaugment_bands(101, 80, 10, 35, 10, 35, 0.05, torch.Generator().manual_seed(7))
    assert torch.equal(out.cpu(), FO.specaugment_apply(before.cpu(), bands, 0.0))


@pytest.mark.gpu
def test_frontend_pipeline_matches_serial_path():
    """N3: batches prepared on the side stream equal the serial front end + the collate's length sort, also while the
    consumer stream is busy."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from conformer_amd.frontend import ConformerAudioFrontend
    from conformer_amd.pipeline import FrontendPipeline
    dev = torch.device("cuda:0")
    fe = ConformerAudioFrontend(device=dev)
    g = torch.Generator().manual_seed(0)
    batches = [[torch.randn(n, generator=g) for n in lens] for lens in ([16000, 4000, 12345], [800, 8000], [3200], [401, 16000, 7000, 160])]
    busy = torch.randn(2048, 2048, device=dev)
    seen = 0
    for (mels, frames, order), src in zip(FrontendPipeline(batches, fe), batches):
        for _ in range(3):
            busy = busy @ busy * 1e-3                                   # keep the consumer stream occupied
        ref_m, ref_l = fe(src)
        ref_l, ref_o = torch.sort(ref_l, descending=True)
        assert torch.equal(frames, ref_l)
        assert torch.equal(mels, ref_m[order])
        assert sorted(order.tolist()) == list(range(len(src)))
        assert [len(src[i]) // 160 + 1 for i in order.tolist()] == frames.tolist()
        seen += 1
    assert seen == len(batches)
