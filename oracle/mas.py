"""ORACLE (test infrastructure only): monotonic alignment search on the CPU.

numpy restatement + ctypes binding of oracle/mas_oracle.c.  Follows
S_monotonic_align.py:5-47 (maximum_path1, tie -> move) and :50-95 (maximum_path2, tie -> stay);
mask_from_lens follows S_monotonic_align.py:117-133.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_NEG = np.float32(-1e32)


def mask_from_lens(shape, x_lens, y_lens):
    """[B,Tx,Ty] 0/1 fp32 mask (S_monotonic_align.py:117-133)."""
    B, Tx, Ty = shape
    mx = (np.arange(Tx)[None, :] < np.asarray(x_lens)[:, None])
    my = (np.arange(Ty)[None, :] < np.asarray(y_lens)[:, None])
    return (mx[:, :, None] & my[:, None, :]).astype(np.float32)


def maximum_path_np(value, mask, tie_move):
    """numpy restatement; value/mask [B,Tx,Ty] fp32 -> path [B,Tx,Ty] fp32 (inputs untouched)."""
    value = np.asarray(value, np.float32)
    mask = np.asarray(mask, np.float32)
    B, Tx, Ty = value.shape
    v = (value * mask).astype(np.float32)                       # :12
    x_len = mask[:, :, 0].sum(1).astype(np.int64)               # :15
    y_len = mask[:, 0, :].sum(1).astype(np.int64)               # :16
    v[:, 1:, 0] = _NEG                                          # :22
    neg_col = np.full((B, 1), _NEG, np.float32)
    for ty in range(1, Ty):                                     # :24-29
        p1 = v[:, :, ty - 1]
        p2 = np.concatenate([neg_col, p1[:, :-1]], 1)
        v[:, :, ty] = (v[:, :, ty] + np.where(p1 > p2, p1, p2)).astype(np.float32)
    path = np.zeros_like(v)
    for b in range(B):
        if x_len[b] <= 0 or y_len[b] <= 0:
            continue
        idx = x_len[b] - 1
        path[b, idx, y_len[b] - 1] = 1
        for ty in range(y_len[b] - 1, 0, -1):
            a = v[b, idx, ty - 1]
            c = v[b, idx - 1, ty - 1] if idx > 0 else _NEG
            if tie_move:                                        # :40  direction = where(p1 > p2, 0, -1)
                if not (a > c) and idx > 0:
                    idx -= 1
            else:                                               # :91
                if idx != 0 and c > a:
                    idx -= 1
            path[b, idx, ty - 1] = 1
    return path


_lib = None


def _load():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "_build", "libmas_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = ctypes.CDLL(so)
        _lib.mas_oracle_f32.restype = ctypes.c_int
        _lib.mas_oracle_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def maximum_path_c(value, mask, tie_move, want_dur=False):
    """C restatement (oracle/mas_oracle.c)."""
    value = np.ascontiguousarray(value, np.float32)
    mask = np.ascontiguousarray(mask, np.float32)
    B, Tx, Ty = value.shape
    path = np.empty_like(value)
    dur = np.zeros((B, Tx), np.int32)
    rc = _load().mas_oracle_f32(value.ctypes.data, mask.ctypes.data, B, Tx, Ty, int(bool(tie_move)),
                                path.ctypes.data, dur.ctypes.data)
    if rc != 0:
        raise MemoryError("mas_oracle_f32")
    return (path, dur) if want_dur else path
