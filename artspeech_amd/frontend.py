"""wav -> normalised log-mel on the HIP path -- SURVEY.md section 8(f) row N3, ``test.py:40-47``:
``to_mel = torchaudio.transforms.MelSpectrogram(n_mels=80, n_fft=2048, win_length=1200, hop_length=300)`` (torchaudio's
defaults otherwise: sample_rate 16000 -- the reference does not pass its 24 kHz --, power 2, centre + reflect padding, periodic
Hann window, HTK mel scale, no filter normalisation) and ``(log(1e-5 + mel) - mean) / std`` with mean -4, std 4.

The arithmetic is two conv GEMMs: the windowed DFT as a [2 x 1025][2048] basis applied to the framed signal, and the
[80][1025] mel filterbank applied to the power spectrum (the basis and the filterbank are the "weights"); framing, power and
log are small kernels (csrc/frontend.hip).  A batch of waves of different lengths is framed into packed columns: no padding
is transformed.  PARITY: torchaudio is absent from the reference tree and from this image, so this row is checked against
torch.stft + the published filterbank formula (oracle/frontend.py), not against vectors of the reference's own front end.
"""
import math

import numpy as np
import torch

from . import _lib, ops
from .ops import _ld, _p, check, stream

N_FFT, WIN, HOP, N_MELS, SR = 2048, 1200, 300, 80, 16000
MEAN, STD, EPS = -4.0, 4.0, 1e-5                                      # test.py:41, :46


def dft_basis(n_fft=N_FFT, win=WIN):
    """[2 (n_fft/2 + 1)][n_fft] fp32: rows f = w[k] cos(2 pi f k / n_fft), then -w[k] sin(...); w = periodic Hann(win) centred in
    n_fft (what torch.stft does with win_length < n_fft).  Computed in fp64."""
    k = np.arange(n_fft, dtype=np.float64)
    w = np.zeros(n_fft)
    left = (n_fft - win) // 2
    w[left:left + win] = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win) / win)
    f = np.arange(n_fft // 2 + 1, dtype=np.float64)[:, None]
    ang = 2.0 * np.pi * f * k[None, :] / n_fft
    return torch.from_numpy(np.concatenate([np.cos(ang) * w, -np.sin(ang) * w], 0).astype(np.float32))


def mel_filterbank(n_freqs=N_FFT // 2 + 1, f_min=0.0, f_max=SR / 2, n_mels=N_MELS, sample_rate=SR):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk"), transposed to [n_mels][n_freqs]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    fb = torch.max(torch.zeros(1), torch.min((-1.0 * slopes[:, :-2]) / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))
    return fb.t().contiguous()


class LogMel:
    """``preprocess(wave)`` of test.py:43-47 for one wave or a batch of waves."""

    def __init__(self, device=None):
        from .models import _need_gpu
        self.device = _need_gpu(device if device is not None else "cuda")
        self.basis = ops.prep_weight(dft_basis()[:, :, None].contiguous(), self.device)            # [1][2048][2050]
        self.fb = ops.prep_weight(mel_filterbank()[:, :, None].contiguous(), self.device)           # [1][1025 -> 1056][80]

    @torch.no_grad()
    def forward_packed(self, waves):
        """waves: list of 1-D fp32 arrays / tensors -> (mel [80][sum frames] packed, frames layout)."""
        dev = self.device
        lens = [int(len(w)) for w in waves]
        if min(lens) <= N_FFT // 2:
            raise ValueError("reflect padding needs more than n_fft / 2 = 1024 samples per wave (torch.stft raises too)")
        frames = [1 + n // HOP for n in lens]
        lay = ops.layout(frames, dev)
        wave = torch.cat([torch.as_tensor(w, dtype=torch.float32).reshape(-1) for w in waves]).to(dev)
        wav_off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=dev)
        X = lay.new(N_FFT)
        L = _lib.lib()
        check(L.as_frame_signal_f32(_p(wave), _p(wav_off), _p(lay.col_off), lay.B, max(frames), N_FFT, HOP, _p(X), _ld(X), stream()),
              "as_frame_signal_f32")
        nf = N_FFT // 2 + 1
        Y = ops.conv_gemm(self.basis, X, lay, lay.new(2 * nf), [(0, 0)])
        P = lay.new(nf)
        check(L.as_spec_power_f32(_p(Y), _ld(Y), nf, lay.N, _p(P), _ld(P), stream()), "as_spec_power_f32")
        M = ops.conv_gemm(self.fb, P, lay, lay.new(N_MELS), [(0, 0)])
        out = lay.new(N_MELS)
        check(L.as_log_norm_f32(_p(M), _ld(M), N_MELS, lay.N, EPS, MEAN, STD, _p(out), _ld(out), stream()), "as_log_norm_f32")
        return out, lay

    def __call__(self, wave):
        """one wave -> [1, 80, frames] (test.py:46's unsqueeze(0)); a list of waves -> ([B, 80, max frames] zero padded, lengths)."""
        from .models import unpack
        single = not isinstance(wave, (list, tuple))
        mel, lay = self.forward_packed([wave] if single else list(wave))
        out = unpack(mel, lay)
        return out if single else (out, [int(v) for v in lay.widths_host])
