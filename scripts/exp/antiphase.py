"""Round 6: two lanes in ANTI-PHASE.  Two identical chains that share the matrix cores fall into step (DESIGN.md section 5 "Wider calls"): both
are in a GEMM at the same time, both in a bandwidth kernel at the same time, and what overlaps is little.  Here every lane's 64-utterance
call is two hipGraphs -- the first half (encoders, towers, duration predictor: as_forward_test_begin) and the second (predictors, decoder:
as_forward_test_finish) -- and events between the lanes keep them half a call apart: lane 1's first half runs beside lane 0's second half and
vice versa, so that different kernel mixes meet.  Against the same graphs free-running (what as_lanes does).  ms per 32 utterances."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from artspeech_amd import _lib, models, synth
from artspeech_amd.models import _p
from artspeech_amd.weights import DEFAULT_STATS, load_distribution
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
sd = synth.synth_state_dict(512, 64, seed=bench.WEIGHT_SEED)
model = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80), None, "second", load_distribution(DEFAULT_STATS), dev)
models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
net = model.ArtsSpeech
L = _lib.lib()
PER_CALL = int(os.environ.get("PER_CALL", "2"))

class Lane:
    def __init__(self, i):
        self.net = net.replica()
        self.net.rt.set_serial(True)
        rt = self.net.rt
        host = bench.merge_hosts([bench.make_inputs(None, seed0=bench.DATA_SEED + 100 * (PER_CALL * i + j))[0] for j in range(PER_CALL)])
        g = self.g = bench.pack_inputs(host, list(range(len(host["frames"]))), dev)
        self.stream = torch.cuda.Stream()
        io = self.io = _lib.ForwardIO()
        io.tokens, io.mel, io.ld_mel = _p(g["tok"]), _p(g["mel"]), g["mel"].stride(0)
        io.f0_raw, io.ema_raw, io.ld_ema = _p(g["f0"]), _p(g["ema"]), g["ema"].stride(0)
        io.forced_dur = _p(g["forced"])
        n2 = 2 * sum(g["frames"])
        self.mel = torch.empty((80, n2), device=dev)
        io.mel_out, io.ld_out = _p(self.mel), n2
        self.ba = rt.batch(tok_lens=g["tok_lens"], ref_lens=g["ref_lens"], frames=g["frames"])
        self.wa, self.na = rt.workspace("a", _lib.AS_MOD_FORWARD_A, self.ba)
        self.wb, self.nb = rt.workspace("b", _lib.AS_MOD_FORWARD_B, self.ba)
        self.first = lambda: _lib.check(L.as_forward_test_begin(rt.model, rt.plan, ctypes.byref(self.ba), ctypes.byref(io), _p(self.wa), self.na,
                                                                torch.cuda.current_stream().cuda_stream), "begin")
        self.second = lambda: _lib.check(L.as_forward_test_finish(rt.model, rt.plan, ctypes.byref(self.ba), ctypes.byref(io), _p(self.wa), self.na,
                                                                  _p(self.wb), self.nb, torch.cuda.current_stream().cuda_stream), "finish")
        with torch.cuda.stream(self.stream):
            self.first(); self.second()
            torch.cuda.synchronize()
            self.want = self.mel.clone()
            self.ga, self.gb, self.gw = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.ga, stream=self.stream):
                self.first()
            with torch.cuda.graph(self.gb, stream=self.stream):
                self.second()
            with torch.cuda.graph(self.gw, stream=self.stream):
                self.first(); self.second()

lanes = [Lane(0), Lane(1)]
torch.cuda.synchronize()

def run(mode, calls):
    """calls = calls per lane"""
    ev = [[torch.cuda.Event() for _ in range(calls + 1)] for _ in lanes]      # ev[l][c]: lane l's first half of call c is done
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in range(calls):
        for li, ln in enumerate(lanes):
            with torch.cuda.stream(ln.stream):
                if mode == "whole":
                    ln.gw.replay()
                elif mode == "halves":
                    ln.ga.replay(); ln.gb.replay()
                else:                                                         # anti-phase: a lane's first half starts when the OTHER lane's first half is done
                    other = ev[1 - li][c if li == 1 else c - 1] if (li == 1 or c > 0) else None
                    if other is not None:
                        ln.stream.wait_event(other)
                    ln.ga.replay()
                    ev[li][c].record(ln.stream)
                    ln.gb.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (2 * calls * PER_CALL) * 1e3

for mode in ("whole", "halves", "anti", "whole", "halves", "anti"):
    run(mode, 3)
    ts = [run(mode, 10) for _ in range(3)]
    ok = all(torch.equal(ln.mel, ln.want) for ln in lanes)
    print(f"{mode:7s} {PER_CALL * 32} utterances per call, two lanes: " + " ".join(f"{t:.3f}" for t in ts) + f" ms per 32 utterances   results equal {ok}", flush=True)
