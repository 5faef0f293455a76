"""every LDS-using kernel of the path on one stream while conv GEMMs run on another: outputs must be bit-identical to a run alone
(follow-up of the AdaIN hazard in DESIGN.md section 5)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artspeech_amd import ops
from artspeech_amd.ops import Layout
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to(dev)

def case_attention(L):
    C = 512; lens = [L] * (64 if L <= 64 else 8); lay = Layout(lens, dev)
    w = ops.prep_weight(torch.randn(3 * C, C, 1, generator=g) / C ** 0.5, dev)
    X = R(C, lay.N); qkv, qkv_h = lay.new(3 * C), ops.new_image(3 * C, lay.N, dev)
    ops.conv_gemm(w, X, lay, qkv, ops.taps_1d(1), yh=qkv_h)
    ek, ev = R(9, 128) * 0.1, R(9, 128) * 0.1
    out = lay.new(C)
    return lambda: ops.relpos_attention_image(qkv, qkv_h, C, 4, 4, ek, ev, lay, out=out).clone()

def case_lstm(H, L, B, cluster):
    lay = Layout([L] * B, dev)
    jobs = [(R(lay.N, 8 * H) * 0.1, R(2, H, 4 * H) * 0.05, lay.new(2 * H))]
    xchg = ops.bilstm_exchange_buffer(1, B, dev) if cluster else None
    return lambda: ops.bilstm(jobs, lay, H, xchg)[0].clone()

def case_ln():
    C = 512; lay = Layout([40] * 96, dev); X = R(C, lay.N); ga, be = R(C), R(C)
    return lambda: ops.channel_layernorm_split(X, lay, ga, be, relu=True).clone()

def case_adain(up):
    B, L, C = 32, 100, 512; lay = Layout([L] * B, dev); lay2 = Layout([2 * L] * B, dev)
    X = R(C, lay.N); gb = R(B, 2 * C); pw, pb = R(C, 3), R(C); xup = lay2.new(C)
    if up:
        return lambda: ops.adain_image(X, lay, gb, 1, lay2.N, ldgb=2 * C, pool_w=pw, pool_b=pb, x_up=xup).clone()
    return lambda: ops.adain_image(X, lay, gb, 1, lay.N, ldgb=2 * C).clone()

def case_gemm(M, N, K, T, tile):
    lay = Layout([N // 32] * 32, dev); w = ops.prep_weight(torch.randn(M, K, T, generator=g) / (K * T) ** 0.5, dev)
    xs = ops.split_act(R(K, lay.N), lay); Y = lay.new(M)
    def run():
        os.environ["AS_GEMM_TILE"] = tile
        y = ops.conv_gemm(w, None, lay, Y, ops.taps_1d(T), xs=xs, K=K).clone()
        os.environ.pop("AS_GEMM_TILE")
        return y
    return run

def case_project():
    lay = Layout([200] * 96, dev); X = R(256, lay.N); w, b = R(10, 256), R(10); Y = lay.new(10)
    return lambda: ops.project_cols(X, lay.N, w, b, Y).clone()

cases = {"attention 40": case_attention(40), "attention 300": case_attention(300), "lstm H128": case_lstm(128, 200, 32, False),
         "lstm H256 cluster": case_lstm(256, 40, 32, True), "layernorm split": case_ln(), "adain": case_adain(False), "adain up": case_adain(True),
         "gemm 21": case_gemm(512, 6400, 512, 3, "21"), "gemm 11": case_gemm(512, 1280, 512, 3, "11"), "project_cols": case_project()}
# the company: the decoder's big GEMM
layg = Layout([200] * 32, dev)
wg = ops.prep_weight(torch.randn(1024, 1024, 3, generator=g) / 55.0, dev)
xsg = ops.split_act(R(1024, layg.N), layg); Yg = layg.new(1024)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for name, fn in cases.items():
    ref = fn(); torch.cuda.synchronize()
    alone = sum(int(not torch.equal(fn(), ref)) for _ in range(10)); torch.cuda.synchronize()
    bad, outs = 0, []
    for i in range(150):
        with torch.cuda.stream(sb):
            ops.conv_gemm(wg, None, layg, Yg, ops.taps_1d(3), xs=xsg, K=1024)
        with torch.cuda.stream(sa):
            outs.append(fn())
        if len(outs) == 15:
            torch.cuda.synchronize()
            bad += sum(int(not torch.equal(o, ref)) for o in outs); outs = []
    torch.cuda.synchronize()
    print(f"{name:20s} alone: {alone}/10 differ   beside the GEMM: {bad}/150 differ")
