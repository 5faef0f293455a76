"""Checkpoint -> the weights blob as_model_create reads (include/artspeech_hip.h).

    "ASWBLOB1" | u32 n | n x { u16 name_len | name | u8 ndim | u32 dims[ndim] | u64 data_offset } | u64 data_bytes | fp32 data

The blob holds the reference's ``ArtsSpeech`` state_dict as it is (models.py:685-701 loads ``params['ArtsSpeech']``: weight_norm
``weight_g / weight_v``, spectral_norm ``weight_orig / weight_u / weight_v``, plain tensors); folding and laying out happens
in the library, so a C / C++ host needs nothing but this file format.  ``python -m artspeech_amd.blob ckpt.pth out.aswb``
converts a reference checkpoint.
"""
import struct
import sys

import numpy as np

from .spec import EXTRACTOR_PREFIXES


def state_dict_to_blob(sd):
    """reference-format state dict (tensors or numpy arrays) -> bytes"""
    entries, chunks, off = [], [], 0
    for k, v in sd.items():
        k = k[len("module."):] if k.startswith("module.") else k
        if k.startswith(EXTRACTOR_PREFIXES):
            continue
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        if not np.issubdtype(a.dtype, np.floating):
            continue                                                  # (BatchNorm counters etc.: nothing on this path reads them)
        a = np.ascontiguousarray(a, dtype="<f4")
        name = k.encode("utf-8")
        entries.append(struct.pack("<H", len(name)) + name + struct.pack("<B", a.ndim) + struct.pack(f"<{a.ndim}I", *a.shape)
                       + struct.pack("<Q", off))
        chunks.append(a.tobytes())
        pad = (-len(chunks[-1])) % 16
        if pad:
            chunks.append(b"\0" * pad)
        off += a.nbytes + pad
    return b"ASWBLOB1" + struct.pack("<I", len(entries)) + b"".join(entries) + struct.pack("<Q", off) + b"".join(chunks)


def main(argv):
    import torch
    if len(argv) != 3:
        raise SystemExit("usage: python -m artspeech_amd.blob checkpoint.pth out.aswb")
    state = torch.load(argv[1], map_location="cpu")
    sd = state["net"]["ArtsSpeech"] if "net" in state else state
    with open(argv[2], "wb") as f:
        f.write(state_dict_to_blob(sd))


if __name__ == "__main__":
    main(sys.argv)
