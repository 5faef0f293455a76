"""CPU: the front end's constant tables (artspeech_amd/frontend.py) against the torch.stft restatement (oracle/frontend.py) and
against torch.fft: the filterbank is the published torchaudio HTK formula, the DFT basis reproduces rfft of the windowed frame."""
import torch

from artspeech_amd import frontend as FE


def test_basis_and_filterbank_tables():
    from oracle import frontend as ofe
    assert torch.equal(FE.mel_filterbank(), ofe.melscale_fbanks().t())
    x = torch.randn(2048, dtype=torch.float64)
    win = torch.zeros(2048, dtype=torch.float64)
    win[424:1624] = torch.hann_window(1200, dtype=torch.float64)
    ref = torch.fft.rfft(x * win)
    got = FE.dft_basis().double() @ x
    assert float((got[:1025] - ref.real).abs().max()) <= 1e-4 and float((got[1025:] - ref.imag).abs().max()) <= 1e-4
