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
    hdr = open(os.path.join(ROOT, "include", "artspeech_hip.h")).read()
    assert L.as_abi_version() == int(re.search(r"#define AS_ABI_VERSION (\d+)", hdr).group(1)) == _lib.AS_ABI_VERSION
    assert L.as_mas_workspace_bytes(2, 40, 100) > 0
    assert L.as_mas_workspace_bytes(2, 1 << 20, 100) == 0      # unsupported geometry -> 0, not a crash


def test_ctypes_struct_layouts_match_the_header(tmp_path):
    """Every struct that crosses the C ABI by value or by pointer has the SAME size and the same offset of its last field in the header
    (compiled here with gcc, as a C host would) and in the ctypes binding (artspeech_amd/_lib.py): a field added on one side only -- the
    way ConvGemmArgs.slab_tr, n_valid, AsAdainArgs.col_w, as_forward_io.frame_cap / segs and as_host_io came in -- fails here, on the CPU."""
    import ctypes
    pairs = [("ConvGemmArgs", _lib.ConvGemmArgs, "n_valid"), ("AsAdainArgs", _lib.AdainArgs, "col_w"), ("AsLnArgs", _lib.LnArgs, "yh"),
             ("AsDownArgs", _lib.DownArgs, "n_out"), ("AsResPairArgs", _lib.ResPairArgs, None), ("as_model_cfg", _lib.ModelCfg, "stats"),
             ("as_batch", _lib.Batch, "frames"), ("as_forward_io", _lib.ForwardIO, "segs"), ("as_host_io", _lib.HostIO, "frame_off"),
             ("BiLstmJob", _lib.BiLstmJob, "ldo")]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "artspeech_hip.h"', 'int main(void) {']
    for cname, _, last in pairs:
        src.append(f'  printf("{cname} %zu %zu\\n", sizeof({cname}), {"offsetof(" + cname + ", " + last + ")" if last else "(size_t)0"});')
    src.append('  printf("as_segments %zu %zu\\n", sizeof(as_segments), offsetof(as_segments, frame_off));')
    src += ['  return 0;', '}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    got = {ln.split()[0]: (int(ln.split()[1]), int(ln.split()[2])) for ln in subprocess.check_output([str(exe)], text=True).splitlines()}
    for cname, cls, last in pairs:
        want = (ctypes.sizeof(cls), getattr(cls, last).offset if last else 0)
        assert got[cname] == want, (cname, got[cname], want)
    # as_segments (built by csrc/lanes.hip only; a C host may pass one): 4 + 17 * 4 + 16 * 4 ints, then pointers and ints
    assert got["as_segments"][0] % 8 == 0 and got["as_segments"][1] > 0


def test_invalid_arguments_are_rejected_without_a_gpu():
    L = _lib.lib()
    assert L.as_mas_f32(None, None, None, 1, 4, 4, 0, None, None, None, None, 0, None) == -1
    assert L.as_conv_gemm_f32(None, None) == -1
    assert L.as_bilstm_f32(None, 1, None, 1, 128, None) == -1
    # entry points added in round 2: the same contract (AS_EINVAL = -1 before anything touches a device)
    assert L.as_bilstm_cluster_f32(None, 1, None, 1, 256, 40, None, 0, None) == -1
    assert L.as_bilstm_cluster_bytes(0, 4) == 0 and L.as_bilstm_cluster_bytes(1, 32) > 0
    assert L.as_relpos_attention_image_f32(None, 0, None, 0, 512, 4, 4, None, None, None, None, 0, None, 1, 40, None, 0, None, None) == -1
    assert L.as_adain_image_f32(None, None) == -1
    assert L.as_model_create(None, 0, None, None) == -1
    assert L.as_plan_create(None, None) == -1
    assert L.as_forward_test(None, None, None, None, None, 0, None, 0, None, None) == -1
    assert L.as_module_workspace_bytes(None, None, 0, None) == 0
    assert L.as_lanes_create(None, 4, None) == -1 and L.as_lanes_submit(None, None, None, None, None) == -1
    assert L.as_lanes_count(None) == 0 and L.as_lanes_next(None) == -1 and L.as_lanes_wait(None, -1) == -1 and L.as_lanes_destroy(None) == 0
    assert not L.as_lanes_stream(None, 0)
    assert L.as_lanes_set_graph_cap(None, 4) == -1 and L.as_lanes_set_layout_cap(None, 64) == -1 and L.as_lanes_stats(None, 0, None) == -1 and L.as_lanes_reserve(None, 0, 0) == -1
    assert L.as_plan_layout_count(None) == -1 and L.as_plan_reset_layouts(None) == -1
    assert L.as_embed_groups_f32(None, 0, None, None, 0, 8, 8, 8, 1.0, None, 8, None) == -1
    assert L.as_split_f16x2_bytes(0, 10) == 0 and L.as_split_f16x2_bytes(20, 10) == 4 * 4 * 11 * 16      # 20 channels: 2 k-blocks, padded to 4; 4 planes; N + 1 columns of 16 bytes


def test_weight_preparation_host_matches_python():
    """as_prep_weight_f16x2_host (what a C host calls, and what as_model_create runs) and ops.split_f16x2_weight (what the Python
    mirror runs) produce the same split fp16 weight image and the same power-of-two scale -- host arithmetic only, no GPU."""
    import ctypes

    import numpy as np
    import torch

    from artspeech_amd import ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    for G, M, K, T, amp in [(1, 80, 130, 3, 0.02), (2, 64, 16, 1, 3.0), (3, 7, 1, 9, 1e-4), (1, 128, 64, 5, 40.0)]:
        w = torch.randn(G, M, K, T, generator=g) * amp
        img, scale = ops.split_f16x2_weight(w)
        n = L.as_prep_weight_f16x2_bytes(G, M, K, T)
        assert n == img.numel() * 2
        out = np.zeros(n // 2, np.int16)
        sc = ctypes.c_float(0)
        wn = np.ascontiguousarray(w.numpy())
        assert L.as_prep_weight_f16x2_host(wn.ctypes.data, G, M, K, T, out.ctypes.data, ctypes.byref(sc)) == 0
        assert sc.value == scale and 2.0 ** 13 <= float(w.abs().max()) * scale < 2.0 ** 14
        assert np.array_equal(out, img.numpy().reshape(-1))
        # the parts reproduce the scaled weights to 2^-22
        parts = img.view(torch.float16).reshape(G, T, -1, 2, 2, M, 8).float()            # [G][T][kb][p][kh][M][8]
        back = (parts[:, :, :, 0] + parts[:, :, :, 1]).permute(0, 4, 2, 3, 5, 1).reshape(G, M, -1, T)[:, :, :K]
        assert float((back / scale - w).abs().max()) <= 2.0 ** -21 * float(w.abs().max())


def test_weight_preparation_with_shortcut_host():
    """as_prep_weight_f16x2_sc_host: the conv's weight sets with a learned shortcut's [Cout][Cin2] weights behind the taps of each set
    (ConvGemmArgs.Xh2 / K2), ONE common power-of-two scale; with Cin2 = 0 it is as_prep_weight_f16x2_host."""
    import ctypes

    import numpy as np
    import torch

    from artspeech_amd import ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(9)
    for G, M, K, T, K2 in [(1, 80, 130, 3, 40), (3, 64, 16, 9, 64), (1, 128, 64, 1, 1216)]:
        w = torch.randn(G, M, K, T, generator=g) * 0.05
        w2 = torch.randn(G, M, K2, generator=g) * 0.4                     # the shortcut sets the scale here
        n = L.as_prep_weight_f16x2_sc_bytes(G, M, K, T, K2)
        kbx, kbx2 = ops.kbx(K), ops.kbx(K2)
        assert n == G * (T * kbx + kbx2) * 4 * M * 16
        out = np.zeros(n // 2, np.int16)
        sc = ctypes.c_float(0)
        wn, w2n = np.ascontiguousarray(w.numpy()), np.ascontiguousarray(w2.numpy())
        assert L.as_prep_weight_f16x2_sc_host(wn.ctypes.data, w2n.ctypes.data, G, M, K, T, K2, out.ctypes.data, ctypes.byref(sc)) == 0
        mx = max(float(w.abs().max()), float(w2.abs().max()))
        assert 2.0 ** 13 <= mx * sc.value < 2.0 ** 14
        img = torch.from_numpy(out).view(torch.float16).reshape(G, T * kbx + kbx2, 2, 2, M, 8).float()      # [G][block][p][kh][M][8]
        full = (img[:, :, 0] + img[:, :, 1]).permute(0, 3, 1, 2, 4)                                           # [G][M][block][kh][8]
        main = full[:, :, : T * kbx].reshape(G, M, T, kbx * 16)[:, :, :, :K].permute(0, 1, 3, 2)
        short = full[:, :, T * kbx:].reshape(G, M, kbx2 * 16)
        assert float((main / sc.value - w).abs().max()) <= 2.0 ** -21 * mx
        assert float((short[:, :, :K2] / sc.value - w2).abs().max()) <= 2.0 ** -21 * mx and not short[:, :, K2:].any()
        # no shortcut: the plain function's bytes
        n0 = L.as_prep_weight_f16x2_bytes(G, M, K, T)
        assert L.as_prep_weight_f16x2_sc_bytes(G, M, K, T, 0) == n0
        a, b = np.zeros(n0 // 2, np.int16), np.zeros(n0 // 2, np.int16)
        sa, sb = ctypes.c_float(0), ctypes.c_float(0)
        assert L.as_prep_weight_f16x2_host(wn.ctypes.data, G, M, K, T, a.ctypes.data, ctypes.byref(sa)) == 0
        assert L.as_prep_weight_f16x2_sc_host(wn.ctypes.data, None, G, M, K, T, 0, b.ctypes.data, ctypes.byref(sb)) == 0
        assert np.array_equal(a, b) and sa.value == sb.value
    # a shortcut weight without a width (and the reverse) is refused
    assert L.as_prep_weight_f16x2_sc_host(wn.ctypes.data, None, G, M, K, T, 8, out.ctypes.data, ctypes.byref(sc)) == -1
    assert L.as_prep_weight_f16x2_sc_host(wn.ctypes.data, w2n.ctypes.data, G, M, K, T, 0, out.ctypes.data, ctypes.byref(sc)) == -1


def test_conv_gemm_second_operand_and_source_arguments_are_validated():
    """ConvGemmArgs.Xh2 / K2 and src_col / N_in: inconsistent combinations return AS_EINVAL before anything touches a device."""
    import ctypes
    L = _lib.lib()
    def args(**kw):
        a = _lib.ConvGemmArgs()
        a.Wh, a.Xh, a.Y = 4096, 8192, 16384                   # (never dereferenced: every case below is refused by the checks)
        a.M, a.N, a.K, a.T, a.Kp, a.ldy, a.acc_scale, a.n_prod, a.in_slope, a.act_slope = 64, 100, 64, 1, 64, 100, 1.0, 3, 0.2, 0.2
        for k, v in kw.items():
            setattr(a, k, v)
        return a
    assert L.as_conv_gemm_f32(ctypes.byref(args(K2=64)), None) == -1                          # a width without an image
    assert L.as_conv_gemm_f32(ctypes.byref(args(K2=-1, Xh2=4096)), None) == -1
    assert L.as_conv_gemm_f32(ctypes.byref(args(K2=64, Xh2=4096 + 8)), None) == -1            # 16-byte alignment
    assert L.as_conv_gemm_f32(ctypes.byref(args(K2=64, Xh2=4096, Xh=None, X=8192, ldx=100)), None) == -1   # beside an image only
    assert L.as_conv_gemm_f32(ctypes.byref(args(src_col=4096, N_in=400)), None) == -1         # own positions need the descriptors
    assert L.as_conv_gemm_f32(ctypes.byref(args(src_col=4096, N_in=0, meta=4096)), None) == -1
    assert L.as_conv_gemm_f32(ctypes.byref(args(src_col=4096, N_in=400, meta=4096, K2=64, Xh2=4096)), None) == -1   # not both
