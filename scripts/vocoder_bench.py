"""Throughput of the HiFi-GAN generator (artspeech_amd/vocoder.py) on the acoustic benchmark's output shape:
32 utterances x 200 mel frames -> 32 x 60 000 samples (2.5 s each at 24 kHz).  hipGraph replay; synthetic weights."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from artspeech_amd import vocoder as V, ops, synth
B, T = int(os.environ.get("B", 32)), int(os.environ.get("T", 200))
dev = torch.device("cuda:0")
gen = V.Generator(None, device=dev).load_state_dict(V.synth_generator_state_dict(None, seed=3407))
lay = ops.layout([T] * B, dev)
mel = lay.new(80); mel.copy_(torch.from_numpy(synth.hash_tensor("voc/bench", (80, lay.N), 1234, 1.0)))
for _ in range(2):
    wav, lay_w = gen.forward_packed(mel, lay)
torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    gen.forward_packed(mel, lay); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        wav, lay_w = gen.forward_packed(mel, lay)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
steps = int(os.environ.get("STEPS", 20))
t0 = time.perf_counter()
for _ in range(steps): g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
samples = lay_w.N
print(json.dumps({"metric": "vocoder audio samples/s (HiFi-GAN generator, batch %d x %d frames)" % (B, T), "value": samples / dt,
                  "ms_per_step": dt * 1e3, "audio_seconds_per_step": samples / 24000.0, "rtf": dt / (samples / 24000.0),
                  "x_realtime": samples / 24000.0 / dt, "finite": bool(torch.isfinite(wav).all())}))
if os.environ.get("PROFILE"):
    import ctypes, csv, collections
    from artspeech_amd import _lib
    L = _lib.lib()
    os.environ["AS_PROF_CSV"] = "/tmp/voc_prof.csv"
    if os.path.exists("/tmp/voc_prof.csv"): os.remove("/tmp/voc_prof.csv")
    L.as_prof_enable(1)
    gen.forward_packed(mel, lay); torch.cuda.synchronize()
    n = 7
    ms = (ctypes.c_double * n)(); fl = (ctypes.c_double * n)(); by = (ctypes.c_double * n)(); cnt = (ctypes.c_int32 * n)()
    L.as_prof_collect(ms, fl, by, cnt, n); L.as_prof_enable(0)
    print("per class ms:", [round(v, 2) for v in ms], "launches", list(cnt), "GEMM TFLOP/s", fl[0] / ms[0] / 1e9)
    agg = collections.OrderedDict()
    for r in csv.reader(open("/tmp/voc_prof.csv")):
        a = agg.setdefault((r[0], r[1]), [0, 0.0, 0.0]); a[0] += 1; a[1] += float(r[2]); a[2] += float(r[3])
    for (c, tag), (k, t, f) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOP', 22))]:
        print(f"cls{c} {tag:44s} n={k:3d} {t:8.3f} ms  {f / t / 1e9 if t else 0:7.1f} TF/s")
