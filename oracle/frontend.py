"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's wav -> normalised log-mel front end,
``test.py:40-47`` (= ``meldataset.py`` preprocess): ``torchaudio.transforms.MelSpectrogram(n_mels=80, n_fft=2048,
win_length=1200, hop_length=300)`` with torchaudio's defaults (sample_rate 16000 -- the reference does not pass 24000 --
f_min 0, f_max sample_rate/2, power 2, centre + reflect padding, periodic Hann window, HTK mel scale, no filter norm), then
``(log(1e-5 + mel) - (-4)) / 4``.

PARITY UNPINNED: torchaudio is a third-party dependency of the reference that is absent from /root/reference and from this
image (the reference pins no version: no requirements file lists one), so no vector of the reference's own front end
can be generated here.  This restates torchaudio's published algorithm (transforms.Spectrogram -> torch.stft;
functional.melscale_fbanks, HTK branch) on torch.stft, which IS in the image; the call site it serves is test.py:42-46."""
import math

import torch

N_FFT, WIN, HOP, N_MELS, SR = 2048, 1200, 300, 80, 16000          # test.py:40 + torchaudio defaults
MEAN, STD = -4.0, 4.0                                                # test.py:41


def melscale_fbanks(n_freqs=N_FFT // 2 + 1, f_min=0.0, f_max=SR / 2, n_mels=N_MELS, sample_rate=SR):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -> [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def preprocess(wave):
    """wave fp32 [L] -> normalised log-mel [80, 1 + L // 300]   (test.py:43-47 without the leading unsqueeze)."""
    wave = torch.as_tensor(wave).float()
    spec = torch.stft(wave, N_FFT, hop_length=HOP, win_length=WIN, window=torch.hann_window(WIN), center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True).abs().pow(2.0)
    mel = torch.matmul(spec.transpose(-1, -2), melscale_fbanks()).transpose(-1, -2)
    return (torch.log(1e-5 + mel) - MEAN) / STD
