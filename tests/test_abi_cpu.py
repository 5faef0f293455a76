"""CPU: the C-ABI library loads and exports every symbol include/artspeech_hip.h declares, and the ctypes
signature table covers them all (no compute calls here)."""
import os
import re
import subprocess

from artspeech_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    hdr = open(os.path.join(ROOT, "include", "artspeech_hip.h")).read()
    return sorted(set(re.findall(r"\b(as_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from artspeech_amd import _build
        _build.build_lib(verbose=False)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    names = declared()
    assert len(names) >= 20
    for s in names:
        assert re.search(rf"\b{s}\b", out), s


def test_ctypes_table_matches_header():
    assert set(_lib._SIGNATURES) == set(declared())
    L = _lib.lib()                     # dlopen + resolve every symbol (torch is imported first by _lib)
    assert L.as_abi_version() >= 1
    assert L.as_mas_workspace_bytes(2, 40, 100) > 0
    assert L.as_mas_workspace_bytes(2, 1 << 20, 100) == 0      # unsupported geometry -> 0, not a crash


def test_invalid_arguments_are_rejected_without_a_gpu():
    L = _lib.lib()
    assert L.as_mas_f32(None, None, None, 1, 4, 4, 0, None, None, None, None, 0, None) == -1
    assert L.as_conv_gemm_f32(None, None) == -1
    assert L.as_bilstm_f32(None, 1, None, 1, 128, None) == -1
