"""Monotonic alignment search: the reference's ``maximum_path`` / ``mask_from_lens`` call surface
on the HIP kernel (csrc/mas.hip).

Selected in the training scripts by ``MAS_type`` (train_first.py:50-66); this build adds
``MAS_type: "hip"``:

    from artspeech_amd.mas import maximum_path, mask_from_lens

``maximum_path(value, mask)`` keeps the signature of S_monotonic_align.maximum_path1/2 and of the
Triton wrapper (S_monotonic_align_Triton.py:56-71): value/mask fp32 [B,Tx,Ty] -> 0/1 path
[B,Tx,Ty].  Unlike the Triton wrapper it does not modify ``value``.
"""
import torch

from . import _lib

TIE_STAY = 0   # maximum_path2 / Triton / (default)
TIE_MOVE = 1   # maximum_path1


def mask_from_len(lens, max_len=None):
    """S_monotonic_align.py:99-113."""
    if max_len is None:
        max_len = lens.max()
    index = torch.arange(max_len).to(lens).view(1, -1)
    return index < lens.unsqueeze(1)


def mask_from_lens(similarity, symbol_lens, mel_lens):
    """S_monotonic_align.py:117-133: (B,S,T) 0/1 mask in similarity's dtype."""
    _, S, T = similarity.size()
    mask_S = mask_from_len(symbol_lens, S)
    mask_T = mask_from_len(mel_lens, T)
    return (mask_S.unsqueeze(2) * mask_T.unsqueeze(1)).to(similarity)


def maximum_path_lens(value, x_lens, y_lens, tie="stay", want=("path",)):
    """MAS from explicit lengths (no dense mask).  value fp32 [B,Tx,Ty] on the GPU.
    Returns a dict with the requested ones of: path fp32 [B,Tx,Ty], dur int32 [B,Tx], rows int32 [B,Ty]."""
    if value.dim() != 3:
        raise ValueError("value must be [B, Tx, Ty]")
    if not value.is_cuda:
        raise _lib.HipLibraryError("maximum_path: value must live in GPU memory (no CPU fallback)")
    value = value.contiguous()
    if value.dtype != torch.float32:
        value = value.float()
    B, Tx, Ty = value.shape
    dev = value.device
    tx = x_lens.to(device=dev, dtype=torch.int32).contiguous()
    ty = y_lens.to(device=dev, dtype=torch.int32).contiguous()
    L = _lib.lib()
    out = {}
    path = torch.empty_like(value) if "path" in want else None
    dur = torch.empty((B, Tx), dtype=torch.int32, device=dev) if "dur" in want else None
    rows = torch.empty((B, Ty), dtype=torch.int32, device=dev) if "rows" in want else None
    if B == 0 or Tx == 0 or Ty == 0:
        for k, t in (("path", path), ("dur", dur), ("rows", rows)):
            if t is not None:
                out[k] = t.zero_()
        return out
    nbytes = L.as_mas_workspace_bytes(B, Tx, Ty)
    if nbytes == 0:
        raise ValueError(f"maximum_path: unsupported lattice shape {tuple(value.shape)} (Tx <= 8192)")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    tie_mode = {"stay": TIE_STAY, "move": TIE_MOVE}[tie]
    with torch.cuda.device(dev):
        rc = L.as_mas_f32(_lib.ptr(value), _lib.ptr(tx), _lib.ptr(ty), B, Tx, Ty, tie_mode,
                          _lib.ptr(path), _lib.ptr(dur), _lib.ptr(rows), _lib.ptr(ws), nbytes, _lib.stream())
    _lib.check(rc, "as_mas_f32")
    for k, t in (("path", path), ("dur", dur), ("rows", rows)):
        if t is not None:
            out[k] = t
    return out


@torch.no_grad()
def soft_maximum_path(feat, symbol_lens, mel_lens, dim=-1, tie="stay"):
    """train_second.py:181-185 (train_first.py:171-177) in one call on the GPU, from lengths:
        s2s_attn = F.softmax(feat, dim);  mono = maximum_path(s2s_attn, mask_from_lens(s2s_attn, symbol_lens, mel_lens));
        d_gt = mono.sum(-1)
    feat fp32 [B, S, T]; returns (s2s_attn [B,S,T], s2s_attn_mono [B,S,T], d_gt int32 [B,S])."""
    if feat.dim() != 3:
        raise ValueError("feat must be [B, S, T]")
    if not feat.is_cuda:
        raise _lib.HipLibraryError("soft_maximum_path: feat must live in GPU memory (no CPU fallback)")
    feat = feat.contiguous().float()
    B, Tx, Ty = feat.shape
    dev = feat.device
    sdim = {-1: 2, 2: 2, 1: 1, -2: 1}.get(dim)
    if sdim is None:
        raise ValueError("dim must be 1 or -1")
    tx = symbol_lens.to(device=dev, dtype=torch.int32).contiguous()
    ty = mel_lens.to(device=dev, dtype=torch.int32).contiguous()
    attn, path = torch.empty_like(feat), torch.empty_like(feat)
    dur = torch.empty((B, Tx), dtype=torch.int32, device=dev)
    if B == 0 or Tx == 0 or Ty == 0:
        return attn, path.zero_(), dur.zero_()
    L = _lib.lib()
    nbytes = L.as_mas_workspace_bytes(B, Tx, Ty)
    if nbytes == 0:
        raise ValueError(f"soft_maximum_path: unsupported lattice shape {tuple(feat.shape)} (Tx <= 8192)")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.as_softmax_mas_f32(_lib.ptr(feat), _lib.ptr(tx), _lib.ptr(ty), B, Tx, Ty, sdim, {"stay": TIE_STAY, "move": TIE_MOVE}[tie],
                                  _lib.ptr(attn), _lib.ptr(path), _lib.ptr(dur), None, _lib.ptr(ws), nbytes, _lib.stream())
    _lib.check(rc, "as_softmax_mas_f32")
    return attn, path, dur


@torch.no_grad()
def maximum_path(value, mask, tie="stay"):
    """Drop-in for S_monotonic_align.maximum_path2 (tie="stay", also the Triton rule) or
    maximum_path1 (tie="move").  Lengths are recovered from the mask exactly as the reference does
    (S_monotonic_align.py:15-16)."""
    x_len = mask[:, :, 0].sum(dim=1).long()
    y_len = mask[:, 0, :].sum(dim=1).long()
    return maximum_path_lens(value, x_len, y_len, tie=tie, want=("path",))["path"].to(value.dtype)


def maximum_path1(value, mask):
    return maximum_path(value, mask, tie="move")


def maximum_path2(value, mask):
    return maximum_path(value, mask, tie="stay")
