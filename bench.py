#!/usr/bin/env python3
"""Headline benchmark: mel frames/s of the acoustic-model inference path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher around it: the script starts its own N ranks as a
                                                        child `python -m torch.distributed.run` BEFORE touching the GPU and relays rank 0's line)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path -- text encoder + articulatory encoder + style towers +
duration predictor + integer alignment expansion + F0/energy/TV predictors + AdaIN mel decoder -- over one
batch of 32 synthetic utterances (config C3 of SURVEY.md section 8: N = 40 tokens, forced integer durations
summing to M = 100 => 200 mel frames per utterance, T_ref = 200; full-size model, seeded synthetic weights),
as ONE call of the library's as_forward_test (csrc/model.hip) captured into a hipGraph and replayed.
`value` is measured with the inputs resident in HBM when the timed region starts (the contract of this benchmark); the SAME K steps are
then timed again with the host <-> device copies inside the region ("transfers": as_lanes_submit_host -- the library copies a submission's
pinned host arrays into the lane's device block, launches the group and copies the mel back: the reference's test.py:96-113 boundary, in
the arrangement `value` reports) -- the two differ by 2-3 per cent because the copies run beside the lanes' kernels.
Consecutive steps are independent batches: by default FOUR are kept in flight per GPU (`--in-flight 4`: each lane its own plan +
workspaces on the same weights, its own hipGraph and HIP stream; the K timed steps alternate between the lanes), so that one batch's tail
rounds, launch gaps and latency-bound stretches are filled by the others' kernels; every lane has its own batch (other seeds) in its own
buffers.  A lane in flight runs its step as ONE chain on ONE stream (`as_plan_set_serial`: the step's independent branches back to
back) -- measured on one box: four single-stream lanes 3.84 ms per step, three 3.94, five 4.19, two lanes with their branches on side
streams (rounds 2-3's arrangement, `--lane-branches`) 4.42; DESIGN.md section 5.  `value` / `ms_per_step` are K steps / wall time; `ms_per_step_one_in_flight` is the same K steps one
at a time with the branches on side streams (the best latency of a step).  Every lane's result is compared bitwise with its own first
eager step and lane 0's with the one-at-a-time run; if a lane ever differed the in-flight timing would be discarded for the
one-at-a-time number (`in_flight_note`).
With N > 1 every rank runs its own 32-utterance batch on its own GPU (utterance batches shard embarrassingly; no data-path
collective) => weak scaling; the only torch.distributed use is the barrier and the max-over-ranks of the elapsed time.
`--global-batch G` instead builds ONE length-varied batch of G utterances, shards it over the ranks (artspeech_amd.shard) and
checks the merged result against a single-rank run (BASELINE config C4).

Prints ONE JSON line (rank 0).  Beside the contract's keys:
  roofline       the dominant kernel, the implicit-GEMM conv on the fp16 matrix cores (f16x3 split-operand arithmetic, fp32-accurate):
                 algorithmic flop / its HIP-event time on the launch stream (an instrumented pass of the same step, branches back to back)
  roofline_hbm   the bandwidth-bound group (AdaIN, LayerNorm, pooling / expansion / image writers): algorithmic bytes / event time / 8 TB/s
  mas            monotonic alignment search (K1) at [32,40,100] and [8,1024,2000]: cells/s, us/column, GB/s, bit-exactness against the
                 reference-generated golden paths
  configs        C2 (batch 1, 150 frames: latency; also in the 16-bit-operand mode with its measured error) and C5 (long form)
  predicted      the same K submissions with the durations PREDICTED on the device (models.py:361-368 as test.py:113 calls it): a frame capacity
                 (as_forward_io.frame_cap), no read-back, hipGraph replay, coalesced; beside the same durations handed over as known counts
  transfers      the headline arrangement with HOST buffers at the boundary (as_lanes_submit_host: the library's copies inside the timed
                 region); `lanes_of_32`: copies around the replays of 32-utterance graphs on this script's own streams
  lanes_native   the same K steps through the library's own lanes (as_lanes_submit: plans, streams, workspaces, hipGraphs kept in C++)
  cpu_baseline   the oracle's CPU restatement timed on this box's host cores on a bounded sample of the same workload
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

B, N_TOK, M_HALF, T_REF = 32, 40, 100, 200          # config C3
FRAME_SEC = 300.0 / 24000.0                          # hop 300 @ 24 kHz (test.py:40)
WEIGHT_SEED, DATA_SEED = 3407, 1234
PEAK_HBM_TBS = 8.0                                   # MI355X_MICROARCH.md (6.3 achievable)
PEAK_F32_MFMA_TFLOPS = 157.3                         # v_mfma_f32_32x32x2_f32 (64 cycles per SIMD)
# f16x3: every fp32 product is three fp16 MFMA products; dense fp16/bf16 peak 256 CU x 4 SIMD x 1024 flop/clk x 2.4 GHz = 2516.6
PEAK_F16_MFMA_TFLOPS = 2516.6
PEAK_X6_TFLOPS = PEAK_F16_MFMA_TFLOPS / 6.0          # round 1's arithmetic (bf16x6): 419.4 -- kept for comparison across rounds
CLASSES = ["conv_gemm", "adain", "layernorm", "attention", "lstm", "mas", "other"]
HBM_GROUP = ("adain", "layernorm", "other")          # the bandwidth-bound kernels (K2, K6, K7, K11)
REPEATS = 3                                          # the K timed steps of the headline arrangements are measured this many times (the first is `value`)


def source_id():
    """sha of the kernel sources (comments and white space removed: a reworded comment is the same build): ties a committed PMC
    profile to the build it was taken from"""
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "artspeech_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) and f != "version.hip":    # (the ABI version is the interface's, not a kernel's: it never bends to a profile's label)
            src = open(os.path.join(d, f), "r", encoding="utf-8", errors="replace").read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            src = re.sub(r"//[^\n]*", "", src)
            h.update(f.encode())
            h.update("".join(src.split()).encode())
    return h.hexdigest()[:16]


def make_inputs(dev, n_utt=B, n_tok=N_TOK, m_half=M_HALF, t_ref=T_REF, vary=False, seed0=DATA_SEED):
    """n_utt synthetic utterances: tokens, normalised reference features, forced integer durations summing to m_half.
    vary: lengths spread over [0.6, 1] of the nominal ones (a ragged batch)."""
    from artspeech_amd import synth
    from artspeech_amd.weights import DEFAULT_STATS
    rng = np.random.default_rng(seed0)
    toks, mels, f0s, emas, forced, frames, tl, rl = [], [], [], [], [], [], [], []
    for b in range(n_utt):
        nt = int(n_tok * rng.uniform(0.6, 1.0)) if vary else n_tok
        tr = max(int(t_ref * rng.uniform(0.6, 1.0)), 70) if vary else t_ref
        mh = max(int(m_half * nt / n_tok), nt)
        toks.append(synth.synth_tokens(nt, seed0 + b))
        mel, f0, ema = synth.synth_ref_features(tr, seed0 + b)
        mels.append(mel)
        f0s.append((f0 * np.float32(DEFAULT_STATS["pitch"][3]) + np.float32(DEFAULT_STATS["pitch"][2])).astype(np.float32))
        emas.append((ema * np.asarray(DEFAULT_STATS["EMA"][3], np.float32)[:, None]
                     + np.asarray(DEFAULT_STATS["EMA"][2], np.float32)[:, None]).astype(np.float32))
        d = np.full(nt, mh // nt, np.int32)
        d[: mh - int(d.sum())] += 1                    # integer durations summing to mh, each >= 1
        forced.append(d)
        frames.append(int(d.sum()))
        tl.append(nt)
        rl.append(tr)
    host = dict(tokens=toks, mel=mels, f0=f0s, ema=emas, forced=forced, frames=frames, tok_lens=tl, ref_lens=rl)
    if dev is None:
        return host, None
    return host, pack_inputs(host, list(range(n_utt)), dev)


def pack_inputs(host, idx, dev):
    """the packed-frames device tensors of utterances idx (as_forward_io's inputs)"""
    cat = lambda key, ax: np.ascontiguousarray(np.concatenate([host[key][i] for i in idx], ax))
    h = dict(tok=torch.from_numpy(cat("tokens", 0).astype(np.int32)), mel=torch.from_numpy(cat("mel", 1)),
             f0=torch.from_numpy(cat("f0", 1).reshape(1, -1)), ema=torch.from_numpy(cat("ema", 1)),
             forced=torch.from_numpy(cat("forced", 0).astype(np.int32)))
    # ONE staging buffer on each side: what a host hands over per batch is one pinned block, moved by ONE host->device copy; the
    # device tensors the path reads are views of the device block (256-byte aligned)
    offs, total = {}, 0
    for k, v in h.items():
        offs[k] = total
        total += (v.numel() * v.element_size() + 255) // 256 * 256
    pin_blob = torch.empty(total, dtype=torch.uint8).pin_memory()
    dev_blob = torch.empty(total, dtype=torch.uint8, device=dev)
    g = {}
    for k, v in h.items():
        nb = v.numel() * v.element_size()
        pin_blob[offs[k]: offs[k] + nb].view(v.dtype).view(v.shape).copy_(v)
        g[k] = dev_blob[offs[k]: offs[k] + nb].view(v.dtype).view(v.shape)
    dev_blob.copy_(pin_blob)
    g["pin_blob"], g["dev_blob"], g["in_bytes"] = pin_blob, dev_blob, sum(v.numel() * v.element_size() for v in h.values())
    g["tok_lens"] = [host["tok_lens"][i] for i in idx]
    g["ref_lens"] = [host["ref_lens"][i] for i in idx]
    g["frames"] = [host["frames"][i] for i in idx]
    return g


def adjacent_submissions(g, per):
    """a packed block g (pack_inputs) of k * per utterances as k submissions of `per` utterances whose tensors are ADJACENT column ranges of
    the block -- what as_lanes_set_coalesce launches as one call, without a copy -- each with its own column range of one output block"""
    n = len(g["frames"])
    assert n % per == 0
    mel_all = torch.empty((g["mel"].shape[0], 2 * sum(g["frames"])), dtype=torch.float32, device=g["mel"].device)
    subs, t0, r0, f0 = [], 0, 0, 0
    for i in range(0, n, per):
        nt, nr, nf = sum(g["tok_lens"][i:i + per]), sum(g["ref_lens"][i:i + per]), sum(g["frames"][i:i + per])
        subs.append(dict(tok=g["tok"][t0:t0 + nt], mel=g["mel"][:, r0:r0 + nr], f0=g["f0"][:, r0:r0 + nr], ema=g["ema"][:, r0:r0 + nr],
                         forced=g["forced"][t0:t0 + nt], tok_lens=g["tok_lens"][i:i + per], ref_lens=g["ref_lens"][i:i + per],
                         frames=g["frames"][i:i + per], out=dict(mel=mel_all[:, 2 * f0:2 * (f0 + nf)])))
        t0, r0, f0 = t0 + nt, r0 + nr, f0 + nf
    return subs, mel_all


def cpu_baseline(host, sd, n_utt):
    """The oracle (CPU restatement of the reference, oracle/acoustic.py) on this box's host cores."""
    from artspeech_amd.weights import DEFAULT_STATS, fold_state_dict, load_distribution
    from oracle import acoustic
    cores = min(os.cpu_count() or 1, 16)              # small per-utterance ops: more threads only add sync cost
    torch.set_num_threads(cores)
    W = fold_state_dict(sd)
    dist = load_distribution(DEFAULT_STATS)
    nb = len(host["tokens"])
    args = [(torch.from_numpy(host["tokens"][b % nb]), torch.from_numpy(host["mel"][b % nb]), torch.from_numpy(host["f0"][b % nb]),
             torch.from_numpy(host["ema"][b % nb]), host["forced"][b % nb]) for b in range(n_utt)]
    acoustic.forward_test(W, *args[0][:4], dist, forced_dur=args[0][4])          # warm-up
    t0 = time.perf_counter()
    outs = []
    for a in args:                                     # bounded: stop after ~20 s of CPU work
        outs.append(acoustic.forward_test(W, *a[:4], dist, forced_dur=a[4]))
        if time.perf_counter() - t0 > 20.0:
            break
    dt = time.perf_counter() - t0
    frames = sum(2 * host["frames"][b % nb] for b in range(len(outs)))
    return dict(value=frames / dt, unit="mel frames/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{len(outs)} utterances of the same C3 workload, one at a time (the reference is batch-1), "
                       f"{dt:.1f} s of CPU work, torch {torch.__version__} fp32"), outs


FORWARD_CALLS = [0]      # eager as_forward_test calls of this process (a kernel trace of the run holds exactly these steps)


class Runner:
    """one geometry of the forward, eager or as a replayed hipGraph"""

    def __init__(self, net, g):
        self.net, self.g, self.out = net, g, None

    def step(self):
        g = self.g
        FORWARD_CALLS[0] += 1
        self.out = self.net.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], forced=g["forced"],
                                           frames_hint=g["frames"], out=self.out)
        return self.out

    def capture(self, with_copies=False):
        """the step as a hipGraph; with_copies: the pinned host->device copy of the inputs and the device->host copy of the mel are
        nodes of the same graph (one launch per batch moves, computes and returns it)"""
        self.step()                                     # also uploads the geometry tables (one blocking upload per new geometry)
        torch.cuda.synchronize()
        if with_copies and getattr(self, "out_host", None) is None:
            self.out_host = torch.empty_like(self.out["mel"], device="cpu").pin_memory()
        graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())

        def body():
            if with_copies:
                self.g["dev_blob"].copy_(self.g["pin_blob"], non_blocking=True)
            self.step()
            if with_copies:
                self.out_host.copy_(self.out["mel"], non_blocking=True)
        with torch.cuda.stream(s):
            body()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=s):
                body()
        torch.cuda.current_stream().wait_stream(s)
        if with_copies:
            self.graph_io = graph
        else:
            self.graph = graph
        return graph.replay

    def timed(self, run, steps, warmup, barrier=lambda: None, repeats=None):
        """W warm-up steps, then EXACTLY `steps` steps between barrier + synchronize on both sides: the elapsed seconds.  repeats (a list):
        the same K steps are timed that many more times AFTERWARDS and every elapsed time -- the first, which is the one returned,
        included -- is appended to it (the driver's K = 20 is 70 ms of measurement: the spread of three says what one is worth)"""
        for _ in range(warmup):
            run()
        out = []
        for _ in range(1 + (REPEATS - 1 if repeats is not None else 0)):
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                run()
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            out.append(time.perf_counter() - t0)
        if repeats is not None:
            repeats.extend(out)
        return out[0]


_BRACKET_MS = None


def bracket_overhead_ms():
    """what one event bracket adds to the kernel inside it (as_prof_bracket_overhead: the intercept over brackets of 1, 2, 4, 8 empty
    kernels), measured once per run on the launch stream"""
    global _BRACKET_MS
    if _BRACKET_MS is None:
        from artspeech_amd import _lib
        o = ctypes.c_double()
        _lib.check(_lib.lib().as_prof_bracket_overhead(_lib.stream(), ctypes.byref(o)), "as_prof_bracket_overhead")
        _BRACKET_MS = float(o.value)
    return _BRACKET_MS


def mfma_sustained():
    """TFLOP/s (fp16 MFMA) this device sustains on random operands: (operands in registers, operands re-read from LDS at the GEMM's ratio)"""
    from artspeech_amd import _lib
    a, b = ctypes.c_double(), ctypes.c_double()
    _lib.check(_lib.lib().as_prof_mfma_sustained(_lib.stream(), ctypes.byref(a), ctypes.byref(b)), "as_prof_mfma_sustained")
    return float(a.value), float(b.value)


def profile_classes(net, runner, steps=3):
    """per-kernel-class time with HIP events on the launch stream, eager launches, the concurrent branches run back to back so that
    an event-bracketed duration is the kernel's own; per class the median of `steps` steps.  `ms_per_step` = the bracketed time minus the brackets' own cost (launches x
    bracket_overhead_ms(): the command processor's work between two event markers, ~4-5 us, which a kernel trace does not count --
    with it removed the figures agree with rocprofv3's kernel durations, profiles/README.md); `ms_per_step_bracketed` = as measured."""
    from artspeech_amd import _lib
    L = _lib.lib()
    net.rt.set_serial(True)
    runner.step()
    torch.cuda.synchronize()
    o = bracket_overhead_ms()
    n = len(CLASSES)
    per_step = []
    for _ in range(steps):                                  # one collection per step: the MEDIAN step per class (an eager launch that the host
        L.as_prof_enable(1)                                 # submits late, e.g. under a profiler, stretches its bracket by milliseconds)
        runner.step()
        ms, fl, by = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)()
        cnt = (ctypes.c_int32 * n)()
        _lib.check(L.as_prof_collect(ms, fl, by, cnt, n), "as_prof_collect")
        per_step.append((list(ms), list(fl), list(by), list(cnt)))
    L.as_prof_enable(0)
    net.rt.set_serial(False)
    out = {}
    for i in range(n):
        if not per_step[0][3][i]:
            continue
        order = sorted(range(steps), key=lambda k: per_step[k][0][i])
        ms_i, fl_i, by_i, cnt_i = (per_step[order[steps // 2]][j][i] for j in range(4))
        out[CLASSES[i]] = dict(ms_per_step=max(ms_i - cnt_i * o, 0.0), ms_per_step_bracketed=ms_i, launches_per_step=cnt_i,
                               gflop_per_step=fl_i / 1e9, gbyte_per_step=by_i / 1e9)
    return out


def gemm_roofline(kern, n_prod):
    k = kern["conv_gemm"]
    tf = k["gflop_per_step"] / k["ms_per_step"] if k["ms_per_step"] > 0 else 0.0           # GFLOP / ms = TFLOP/s
    return tf, PEAK_F16_MFMA_TFLOPS / n_prod


def bench_mas(dev):
    """K1 at the two shapes of SURVEY.md D2, timed with HIP events, checked against the paths the REFERENCE's maximum_path1 / 2 produced
    (tests/golden/mas_*.npz, made by tests/golden/make_golden.py from the reference itself)."""
    from artspeech_amd import mas, synth
    gd = os.path.join(ROOT, "tests", "golden")
    out = {}
    small = np.load(os.path.join(gd, "mas_small.npz"))
    large = np.load(os.path.join(gd, "mas_large.npz"))
    Bl, Txl, Tyl = (int(v) for v in large["shape"])
    u = synth.hash_tensor(f"mas/{Bl}x{Txl}x{Tyl}", (Bl, Txl, Tyl), int(large["seed"]))
    cases = {"32x40x100": (small["c3_32x40x100/value"], small["c3_32x40x100/x_lens"], small["c3_32x40x100/y_lens"],
                           small["c3_32x40x100/rows_v1"], small["c3_32x40x100/rows_v2"]),
             f"{Bl}x{Txl}x{Tyl}": ((u * u).astype(np.float32), large["x_lens"], large["y_lens"], large["rows_v1"], large["rows_v2"])}
    for name, (value, xl, yl, r1, r2) in cases.items():
        v = torch.from_numpy(value).to(dev)
        xt, yt = torch.from_numpy(xl).to(dev), torch.from_numpy(yl).to(dev)   # (resident, as the lattice: a host tensor is an H2D copy per call)
        exact = True
        for tie, rows in (("move", r1), ("stay", r2)):
            got = mas.maximum_path_lens(v, xt, yt, tie=tie, want=("rows",))["rows"].cpu().numpy()
            exact = exact and bool(np.array_equal(got, rows))
        run = lambda: mas.maximum_path_lens(v, xt, yt, tie="stay", want=("path", "dur"))
        for _ in range(3):
            run()
        # n calls captured into ONE hipGraph and replayed: the device's time per call (called one by one from Python, the 20-30 us of
        # host work per call would be what is measured on the small lattice)
        n = 20
        torch.cuda.synchronize()
        graph, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.stream(st):
            run()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=st):
                for _ in range(n):
                    run()
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        reps = 5
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) / (n * reps) / 1e3
        cells = int(xl.astype(np.int64) @ yl.astype(np.int64))             # valid lattice cells
        nbytes = 4 * cells + 4 * value.size                                 # lattice read + dense 0/1 path written
        out[name] = dict(us=sec * 1e6, gcells_per_s=cells / sec / 1e9, us_per_column=sec * 1e6 / int(yl.max()),
                         gb_per_s=nbytes / sec / 1e9, frac_of_hbm_peak=nbytes / sec / 1e12 / PEAK_HBM_TBS, bit_exact_vs_reference=exact)
    return out


def bench_config(net, dev, name, n_utt, n_tok, m_half, t_ref, steps, n_prod_modes=(3,), lanes=0):
    """another BASELINE config on the same model: graph-replayed step time, frames/s, its conv-GEMM rate; lanes > 0: also the rate with
    that many batches of the config in flight (as_lanes)"""
    host, g = make_inputs(dev, n_utt, n_tok, m_half, t_ref, seed0=DATA_SEED + 1000)
    res = {}
    if lanes:
        batches = [g] + [make_inputs(dev, n_utt, n_tok, m_half, t_ref, seed0=DATA_SEED + 1000 + 100 * i)[1] for i in range(1, lanes)]
        chain = net.replica()                                  # a plan like a lane's (one chain, branches' conv GEMMs sharing launches), alone
        chain.rt.set_serial(True)
        firsts = [Runner(chain, b).step()["mel"].clone() for b in batches]
        rc = Runner(chain, g)
        res["ms_per_step_one_chain_alone"] = rc.timed(rc.capture(), steps, 3) / steps * 1e3
        # the same batch with the durations PREDICTED on the device: one lane, a frame capacity (no read-back, hipGraph replay) -- and through
        # the read-back path (as_forward_test_begin -> B + 1 integers to the host -> _finish), one at a time
        try:
            ref = chain.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"])
            tot = int(ref["frame_off"][-1])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                chain.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], out=ref)
            torch.cuda.synchronize()
            rb_ms = (time.perf_counter() - t0) / steps * 1e3
            cap = int(1.25 * tot) + 8
            r1 = chain.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], frame_cap=cap)
            graph, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                chain.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], frame_cap=cap, out=r1)
                torch.cuda.synchronize()
                with torch.cuda.graph(graph, stream=st):
                    chain.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], frame_cap=cap, out=r1)
                for _ in range(3):
                    graph.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    graph.replay()
                torch.cuda.synchronize()
                cap_ms = (time.perf_counter() - t0) / steps * 1e3
            ok = bool(torch.equal(r1["frame_off"], ref["frame_off"]) and float((r1["mel"][:, : 2 * tot] - ref["mel"]).abs().max()) <= 3e-5)
            res["predicted_durations"] = dict(ms_per_step_frame_capacity_one_chain_alone=cap_ms, ms_per_step_read_back_path=rb_ms, mel_frames=2 * tot,
                                              frame_cap=cap, results_verified=ok,
                                              note="the durations PREDICTED on the device: as a replayed hipGraph under a frame capacity (one chain alone, like "
                                                   "ms_per_step_one_chain_alone for forced durations) and eagerly through the read-back path")
        except Exception as e:
            res["predicted_durations"] = {"error": repr(e)[:200]}
        nl = bench_native_lanes(net, batches, firsts, 4 * steps, 0)
        res["in_flight"] = dict(lanes=lanes, ms_per_step=nl["ms_per_step"], ms_per_utt=nl["ms_per_step"] / n_utt,
                                frames_per_s=2 * sum(g["frames"]) / (nl["ms_per_step"] * 1e-3), results_bitwise_equal=nl["results_bitwise_equal"],
                                note="throughput with this many batches of the config in flight (as_lanes); ms_per_step above is ONE batch alone")
    ref_mel = None
    for n_prod in n_prod_modes:
        net.rt.set_operand_mode(n_prod)
        r = Runner(net, g)
        run = r.capture()
        el = r.timed(run, steps, 3)
        mel = r.out["mel"].clone()
        kern = profile_classes(net, Runner(net, g), steps=2)
        tf, peak = gemm_roofline(kern, n_prod)
        frames = 2 * sum(g["frames"])
        d = dict(ms_per_step=el / steps * 1e3, ms_per_utt=el / steps * 1e3 / n_utt, frames_per_s=frames * steps / el,
                 x_realtime=frames * steps / el * FRAME_SEC, gemm_tflops=tf, gemm_frac_of_peak=tf / peak,
                 gemm_gflop_per_step=kern["conv_gemm"]["gflop_per_step"],
                 launches_per_step={c: kern[c]["launches_per_step"] for c in kern}, launches=sum(kern[c]["launches_per_step"] for c in kern),
                 # the conv GEMMs' algorithmic bytes per step (at batch 1: the weights, 4 bytes each) over the step's time against the HBM peak:
                 # what a batch-1 step is bound by if it were bound by any unit (it is bound by its count of dependent launches)
                 gemm_algorithmic_gbyte_per_step=kern["conv_gemm"]["gbyte_per_step"],
                 weight_stream_frac_of_hbm=kern["conv_gemm"]["gbyte_per_step"] / (el / steps * 1e3) / PEAK_HBM_TBS,
                 attention_ms_per_step=kern.get("attention", {}).get("ms_per_step"), lstm_ms_per_step=kern.get("lstm", {}).get("ms_per_step"))
        if n_prod == 3:
            ref_mel = mel
            res.update(d)
        else:
            d["arithmetic"] = "fp16 operands (h parts only, one matrix-core product per fp32 product), fp32 accumulate"
            d["mel_max_abs_vs_f16x3"] = float((mel - ref_mel).abs().max())
            res["f16_operand_mode"] = d
    net.rt.set_operand_mode(3)
    res["workload"] = f"{name}: batch {n_utt}, {n_tok} tokens -> {2 * m_half} mel frames per utterance, T_ref {t_ref}, forced durations"
    return res


def class_profile(fn, reps=3):
    """per-class (HIP events, bracket cost removed) kernel time / algorithmic flop / bytes / launches of fn(): the median of `reps` runs"""
    from artspeech_amd import _lib
    L = _lib.lib()
    o = bracket_overhead_ms()
    n = len(CLASSES)
    runs = []
    for _ in range(reps):
        L.as_prof_enable(1)
        fn()
        ms, fl, by = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)()
        cnt = (ctypes.c_int32 * n)()
        _lib.check(L.as_prof_collect(ms, fl, by, cnt, n), "as_prof_collect")
        runs.append((list(ms), list(fl), list(by), list(cnt)))
    L.as_prof_enable(0)
    out = {}
    for i in range(n):
        if not runs[0][3][i]:
            continue
        order = sorted(range(reps), key=lambda k: runs[k][0][i])
        ms_i, fl_i, by_i, cnt_i = (runs[order[reps // 2]][j][i] for j in range(4))
        out[CLASSES[i]] = dict(ms=max(ms_i - cnt_i * o, 0.0), ms_bracketed=ms_i, launches=cnt_i, gflop=fl_i / 1e9, gbyte=by_i / 1e9)
    return out


def bench_surface(net, dev, reps=5):
    """The test.py chain AROUND the path for the C3 batch (/root/reference/test.py:94-125): reference wave ->
    log-mel front end -> JDCNet + EMA_Predictor -> the acoustic model (forced durations as in the headline) -> HiFi-GAN generator.
    Per-stage device time (HIP events around the stage's eager launches, warm, median of `reps`), the chain's audio real-time factor, and
    a roofline of the generator's convs (algorithmic flop of its launches / their kernel time / the f16x3 ceiling).  Outside the headline's
    timed region; seeded synthetic weights for every module (none ship with the reference)."""
    from artspeech_amd import ema as E
    from artspeech_amd import frontend as FE
    from artspeech_amd import jdc as J
    from artspeech_amd import ops, synth
    from artspeech_amd import vocoder as V
    host, g = make_inputs(dev)
    nb = len(g["ref_lens"])
    # (the reference waves are resident in HBM when the clock starts, like every other input of this benchmark: with host arrays the
    #  stage's time is the pageable copy and whatever else runs on the box's host cores -- 1.7 ms on one box, 76 ms on another)
    waves = [torch.from_numpy(0.1 * synth.hash_tensor(f"surface/wave/{b}", ((t - 1) * FE.HOP + 1,), 77, 1.0)).to(dev)
             for b, t in enumerate(g["ref_lens"])]
    fe = FE.LogMel(device=dev)
    jd = J.JDCNet(device=dev).load_state_dict(J.synth_jdc_state_dict(1, seed=3407))
    em = E.EMA_Predictor(device=dev).load_state_dict(E.synth_ema_state_dict(seed=3407))
    gen = V.Generator(None, device=dev).load_state_dict(V.synth_generator_state_dict(None, seed=3407))
    state = {}

    def s_front():
        state["mel"], state["lay"] = fe.forward_packed(waves)

    def s_jdc():
        state["f0"] = jd.forward_packed(state["mel"], state["lay"])

    def s_ema():
        m = state["mel"][:, : state["lay"].N]
        n_raw = torch.log(torch.exp(m * 4 - 4).norm(dim=0, keepdim=True)).contiguous()          # models.py:431, :655-660
        state["ema"] = em.forward_packed(state["f0"], n_raw, state["mel"], state["lay"])

    def s_acoustic():
        N = state["lay"].N
        state["ac"] = net.forward_packed(g["tok"], g["tok_lens"], state["mel"][:, :N].contiguous(), state["f0"][:, :N].contiguous(),
                                         state["ema"][:, :N].contiguous(), g["ref_lens"], forced=g["forced"], frames_hint=g["frames"], out=state.get("ac"))

    def s_vocoder():
        lay2 = ops.layout([2 * f for f in g["frames"]], dev)
        state["wav"], state["lay_w"] = gen.forward_packed(state["ac"]["mel"], lay2)

    stages = [("frontend_wave_to_logmel", s_front), ("jdcnet_f0", s_jdc), ("ema_predictor", s_ema), ("acoustic_model", s_acoustic),
              ("hifigan_generator", s_vocoder)]
    for _, fn in stages:                                   # warm: weights prepared, layouts cached
        fn()
    torch.cuda.synchronize()
    per = {k: [] for k, _ in stages}
    whole = []
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(stages) + 1)]
        ev[0].record()
        for i, (_, fn) in enumerate(stages):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        for i, (k, _) in enumerate(stages):
            per[k].append(ev[i].elapsed_time(ev[i + 1]))
        whole.append(ev[0].elapsed_time(ev[-1]))
    med = lambda v: sorted(v)[len(v) // 2]
    samples = state["lay_w"].N
    audio_s = samples / 24000.0
    res = dict(workload=f"{nb} utterances x {g['ref_lens'][0]}-frame reference waves -> {2 * sum(g['frames'])} mel frames -> {samples} samples "
                        f"({audio_s:.1f} s of 24 kHz audio)",
               stage_ms={k: med(v) for k, v in per.items()}, chain_ms=med(whole), audio_seconds=audio_s, rtf=med(whole) * 1e-3 / audio_s,
               x_realtime=audio_s / (med(whole) * 1e-3), finite=bool(torch.isfinite(state["wav"]).all()),
               timing="eager launches of every stage (Python callers of the C ABI), HIP events between the stages, warm, median of %d" % reps)
    # every stage as a replayed hipGraph: the device's time (called eagerly from Python, the extractors' 25-75 launches are host-paced)
    def graph_ms(fn, n=10):
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=st):
                fn()
        torch.cuda.synchronize()
        for _ in range(2):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            graph.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    mel_v, lay2 = state["ac"]["mel"], ops.layout([2 * f for f in g["frames"]], dev)
    gms = {"jdcnet_f0": graph_ms(s_jdc), "ema_predictor": graph_ms(s_ema), "acoustic_model": graph_ms(s_acoustic)}
    voc_ms = graph_ms(lambda: gen.forward_packed(mel_v, lay2))
    gms["hifigan_generator"] = voc_ms
    res["stage_ms_graph_replay"] = gms
    res["chain_ms_graph_replay"] = res["stage_ms"]["frontend_wave_to_logmel"] + sum(gms.values())
    res["x_realtime_graph_replay"] = audio_s / (res["chain_ms_graph_replay"] * 1e-3)
    kc = class_profile(lambda: gen.forward_packed(mel_v, lay2))
    k0 = kc["conv_gemm"]
    tf = k0["gflop"] / k0["ms"]
    res["vocoder"] = dict(ms_per_batch_graph_replay=voc_ms, samples_per_s=samples / (voc_ms * 1e-3), x_realtime=audio_s / (voc_ms * 1e-3),
                          kernel_classes=kc,
                          roofline=dict(bound="mfma", kernel="the generator's convs: conv_gemm_h3_kernel at 512-128 channels, respair_kernel (a residual step per "
                                                               "launch) at 64 / 32 channels x up to 1.92 M columns, dilated k = 3 / 7 / 11, conv_post; "
                                                               "Vocoder/vocoder.py:75-125",
                                        achieved=tf, peak=PEAK_F16_MFMA_TFLOPS / 3, unit="TFLOP/s", frac=tf / (PEAK_F16_MFMA_TFLOPS / 3),
                                        algorithmic_gflop_per_batch=k0["gflop"], gemm_ms_per_batch=k0["ms"], launches=k0["launches"],
                                        algorithmic_gbyte_per_batch=k0["gbyte"], hbm_frac_if_bytes_bound=k0["gbyte"] / k0["ms"] / PEAK_HBM_TBS))
    for name, fn in (("jdcnet", s_jdc), ("ema_predictor", s_ema)):
        kc = class_profile(fn)
        k0 = kc["conv_gemm"]
        res[name] = dict(ms_per_batch=res["stage_ms"]["jdcnet_f0" if name == "jdcnet" else "ema_predictor"], kernel_classes=kc,
                         gemm_tflops=k0["gflop"] / k0["ms"], gemm_frac_of_f16x3_peak=k0["gflop"] / k0["ms"] / (PEAK_F16_MFMA_TFLOPS / 3))
    return res


def bench_transfers(lanes, steps, warmup, barrier=lambda: None):
    """The boundary hands over HOST buffers (test.py:96-113 moves tokens / mel to the device inside `synthesis`): per step one pinned
    host->device copy of the batch's inputs, the step, one device->host copy of the mel into pinned memory.  Two forms, both timed
    exactly like the headline (K steps alternating over the lanes, wall clock, synchronised on both sides), the faster one reported:
      "stream": copy, hipGraph replay, copy enqueued on the lane's own stream (a lane's copies overlap the other lane's kernels);
      "graph":  the two copies are nodes of the lane's hipGraph (one launch per batch)."""
    res = {}
    # --- copies enqueued around the replay
    outs = [torch.empty_like(r.out["mel"], device="cpu").pin_memory() for r, _, _ in lanes]
    it = [0]

    def once():
        i = it[0] % len(lanes)
        r, fn, st = lanes[i]
        it[0] += 1
        with torch.cuda.stream(st):
            r.g["dev_blob"].copy_(r.g["pin_blob"], non_blocking=True)
            fn()
            outs[i].copy_(r.out["mel"], non_blocking=True)

    res["stream"] = lanes[0][0].timed(once, steps, warmup, barrier)
    ok = all(torch.equal(o, r.out["mel"].cpu()) for o, (r, _, _) in zip(outs, lanes))
    # --- copies as graph nodes
    try:
        fns = [r.capture(with_copies=True) for r, _, _ in lanes]
        it[0] = 0

        def once_g():
            i = it[0] % len(lanes)
            it[0] += 1
            with torch.cuda.stream(lanes[i][2]):
                fns[i]()
        for r, _, _ in lanes:
            r.out_host.zero_()
        res["graph"] = lanes[0][0].timed(once_g, steps, warmup, barrier)
        ok = ok and all(torch.equal(r.out_host, r.out["mel"].cpu()) for r, _, _ in lanes)
    except Exception as e:                               # (capture of pinned copies not supported: keep the stream form)
        res["graph_error"] = repr(e)[:200]
    # the copies on their own (one lane, events on its stream)
    r, fn, st = lanes[0]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.cuda.stream(st):
        ev[0].record()
        r.g["dev_blob"].copy_(r.g["pin_blob"], non_blocking=True)
        ev[1].record()
        fn()
        ev[2].record()
        outs[0].copy_(r.out["mel"], non_blocking=True)
        ev[3].record()
    torch.cuda.synchronize()
    nb_in, nb_out = r.g["in_bytes"], outs[0].numel() * 4
    h2d, d2h = ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])
    form = min((k for k in ("stream", "graph") if k in res), key=lambda k: res[k])
    dt = res[form]
    return dict(ms_per_step_including_transfers=dt / steps * 1e3, elapsed_s=dt, form=form, copies_verified=ok,
                ms_per_step_by_form={k: v / steps * 1e3 for k, v in res.items() if isinstance(v, float)}, graph_error=res.get("graph_error"),
                h2d_ms=h2d, d2h_ms=d2h, h2d_bytes=nb_in, d2h_bytes=nb_out, h2d_gb_per_s=nb_in / h2d / 1e6, d2h_gb_per_s=nb_out / d2h / 1e6,
                note="pinned host buffers; per lane and step: one H2D copy of the batch's inputs -> the step -> D2H of the mel, on the lane's "
                     "own stream (a lane's copies overlap the other lane's kernels)")


def bench_native_lanes(net, batches, firsts, steps, warmup):
    """The same K steps through the LIBRARY's own lanes (as_lanes, csrc/lanes.hip: serial plans, streams, workspaces and hipGraphs kept by
    the C++ runtime; one as_lanes_submit per batch) instead of this script's graphs and streams: the arrangement of the headline number
    as a C-ABI object.  Results are compared bitwise with the lanes' own first eager steps."""
    from artspeech_amd import models
    n = len(batches)
    lanes = models.Lanes(net, n)
    outs = [None] * n

    def submit(i):
        g = batches[i % n]
        _, outs[i % n] = lanes.submit(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], forced=g["forced"],
                                      frames=g["frames"], out=outs[i % n])
    for i in range(max(warmup, 3 * n)):                       # (eager, captured, replayed)
        submit(i)
    lanes.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        submit(i)
    lanes.wait()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = all(torch.equal(o["mel"], f) for o, f in zip(outs, firsts))
    lanes.close()
    return dict(ms_per_step=el / steps * 1e3, lanes=n, results_bitwise_equal=ok,
                note="as_lanes_submit per batch: the library keeps the serial plans, streams, workspaces and hipGraphs")


def merge_hosts(hosts):
    """several make_inputs host dicts as one (utterances in order)"""
    return {k: [v for h in hosts for v in h[k]] for k in hosts[0]}


def bench_coalesced(net, hosts, k, n_lanes, steps, warmup, barrier=lambda: None):
    """The K timed steps as K submissions of ONE 32-utterance batch each through the library's lanes with as_lanes_set_coalesce(k): a lane's k
    consecutive batches live in adjacent column ranges of one block, so the lane launches them as ONE as_forward_test call over 32 k
    utterances, without a copy (wider conv GEMM launches, one fetch of the weights for k batches; models.py:361-362 processes one utterance
    at a time: any grouping is legal).  hosts: k * n_lanes make_inputs host dicts (one per batch).  Every batch's mel is compared with the
    same batch run ALONE as one merged chain (not bitwise: k times the columns, another GEMM tile)."""
    from artspeech_amd import models
    dev = net.rt.device
    per = len(hosts[0]["frames"])
    chain = net.replica()
    chain.rt.set_serial(True)
    blocks, alone = [], []
    for i in range(n_lanes):
        hs = hosts[i * k:(i + 1) * k]
        g = pack_inputs(merge_hosts(hs), list(range(per * k)), dev)
        subs, mel_all = adjacent_submissions(g, per)
        blocks.append((g, subs, mel_all))
        for h in hs:
            gj = pack_inputs(h, list(range(per)), dev)
            alone.append(Runner(chain, gj).step()["mel"].clone())
    torch.cuda.synchronize()
    lanes = models.Lanes(net, n_lanes)
    lanes.set_coalesce(k)
    order = [sub for (_, subs, _) in blocks for sub in subs]

    def submit(i):
        sub = order[i % len(order)]
        lanes.submit(sub["tok"], sub["tok_lens"], sub["mel"], sub["f0"], sub["ema"], sub["ref_lens"], forced=sub["forced"], frames=sub["frames"],
                     out=sub["out"])
    for i in range(max(warmup // len(order) + 1, 4) * len(order)):   # whole rounds: eager, eager (graph plan), captured, replayed
        submit(i)
    lanes.wait()
    els = []
    for _ in range(REPEATS):                                      # the K steps, REPEATS times over (the first is the line's figure)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            submit(i)
        lanes.wait()                                              # (launches a group that is still short of its k)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        els.append(time.perf_counter() - t0)
        for i in range(steps, (steps // len(order) + 1) * len(order)):   # finish the round: every output block is of ONE launch generation
            submit(i)
        lanes.wait()
    el = els[0]
    torch.cuda.synchronize()
    worst = max(float((sub["out"]["mel"] - a).abs().max()) for sub, a in zip(order, alone))
    merged = sum(lanes.merged_calls(i) for i in range(n_lanes))
    st = lanes.stats(0)
    first = order[0]["out"]["mel"].clone()
    lanes.close()
    # The SAME K submissions with HOST buffers at the boundary (/root/reference/test.py:96-113 moves tokens / mel to the device
    # inside `synthesis`): as_lanes_submit_host -- per submission the pinned host -> device copies into the lane's own block, the group's
    # launch, the device -> host copy of every submission's mel, all issued by the library (the copies on its two copy streams, ordered against
    # the lane's kernels by events; two device blocks per lane, alternating).  Same lanes, same coalescing,
    # same batches; the host results must be the BITS of the device-buffer run above (same groups, same graphs' kernels).
    pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
    hb = []
    for h in hosts:
        cat = lambda key, ax: np.concatenate([h[key][b] for b in range(per)], ax)
        hb.append(dict(tok=pin(cat("tokens", 0).astype(np.int32)), mel=pin(cat("mel", 1)), f0=pin(cat("f0", 1).reshape(-1)), ema=pin(cat("ema", 1)),
                       forced=pin(cat("forced", 0).astype(np.int32)), tok_lens=h["tok_lens"], ref_lens=h["ref_lens"], frames=h["frames"],
                       out=torch.zeros((net.rt.cfg.n_mels, 2 * sum(h["frames"])), dtype=torch.float32).pin_memory()))
    lanes = models.Lanes(net, n_lanes)
    lanes.set_coalesce(k)

    def submit_h(i):
        b = hb[i % len(hb)]
        lanes.submit_host(b["tok"], b["tok_lens"], b["mel"], b["f0"], b["ema"], b["ref_lens"], b["forced"], b["frames"], b["out"])
    for i in range(max(warmup // len(hb) + 1, 4) * len(hb)):
        submit_h(i)
    lanes.wait()
    els_h = []
    for _ in range(REPEATS):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            submit_h(i)
        lanes.wait()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        els_h.append(time.perf_counter() - t0)
        for i in range(steps, (steps // len(hb) + 1) * len(hb)):
            submit_h(i)
        lanes.wait()
    copies_ok = all(torch.equal(b["out"], sub["out"]["mel"].cpu()) for b, sub in zip(hb, order))
    in_bytes = sum(hb[0][key].numel() * hb[0][key].element_size() for key in ("tok", "mel", "f0", "ema", "forced"))
    host_boundary = dict(elapsed_s=els_h[0], ms_per_step_including_transfers=els_h[0] / steps * 1e3,
                         ms_per_step_repeats=[e / steps * 1e3 for e in els_h],
                         # (median of the REPEATS passes of either form: one 20-step pass is 75 ms of measurement)
                         vs_resident_inputs=sorted(els_h)[len(els_h) // 2] / sorted(els)[len(els) // 2], copies_verified=bool(copies_ok),
                         h2d_bytes_per_step=in_bytes, d2h_bytes_per_step=hb[0]["out"].numel() * 4, graph_launches_lane0=lanes.stats(0)["graph_launches"],
                         arrangement="as the line's value",
                         note="as_lanes_submit_host: per 32-utterance submission five pinned host -> device copies (tokens, forced durations, f0, the EMA "
                              "and mel rows) into the lane's own device block, the group's launch, one device -> host copy of the submission's mel -- "
                              "all issued by the library (copies on its two copy streams, events against the lane's kernels); results bitwise equal to the "
                              "device-buffer submissions")
    lanes.close()
    return dict(elapsed_s=el, ms_per_step=el / steps * 1e3, ms_per_step_repeats=[e / steps * 1e3 for e in els], coalesce=k, lanes=n_lanes,
                utterances_per_call=per * k, merged_calls=merged,
                graph_launches_lane0=st["graph_launches"], max_abs_vs_each_batch_alone=worst, results_verified=bool(worst <= 3e-5),
                host_boundary=host_boundary,
                note="as_lanes_submit per 32-utterance batch; a lane launches its k adjacent batches as one as_forward_test call"), first


def bench_predicted(net, hosts, k, n_lanes, steps, warmup, barrier=lambda: None):
    """The same K submissions as `forward(step="test")` is actually called (/root/reference/models.py:361-368, test.py:113): durations
    PREDICTED by the model, no forced durations, no frame counts from the host.  as_forward_io.frame_cap: every submission names the
    half-rate frames it has room for (here 1.25 x what the predictor yields on these inputs, measured once through the read-back path),
    the call is sized by that capacity and finds the utterances' extents on the device -- so the lanes replay it from hipGraphs and
    coalesce k adjacent submissions into one call exactly as with known counts.  Every submission's mel and frame offsets are held
    against the same submission run alone through the read-back path."""
    from artspeech_amd import models
    dev = net.rt.device
    per = len(hosts[0]["frames"])
    chain = net.replica()
    chain.rt.set_serial(True)
    order, known = [], []
    for i in range(n_lanes):
        hs = hosts[i * k:(i + 1) * k]
        g = pack_inputs(merge_hosts(hs), list(range(per * k)), dev)
        subs, _ = adjacent_submissions(g, per)
        durs, frames = [], []
        for h, sub in zip(hs, subs):
            gj = pack_inputs(h, list(range(per)), dev)
            ref = chain.forward_packed(gj["tok"], gj["tok_lens"], gj["mel"], gj["f0"], gj["ema"], gj["ref_lens"])   # predicted durations, read back
            sub["want_off"] = ref["frame_off"].cpu()
            sub["want_mel"] = ref["mel"].clone()
            sub["total"] = int(sub["want_off"][-1])
            sub["cap"] = int(1.25 * sub["total"]) + 8
            sub["res"] = None
            order.append(sub)
            durs.append(ref["dur_i"].clone())
            frames += [int(sub["want_off"][b + 1] - sub["want_off"][b]) for b in range(per)]
        # the SAME durations handed over as host-known counts (forced = what the predictor said): what the capacity mechanism itself costs
        g2 = dict(g)
        g2["forced"], g2["frames"] = torch.cat(durs).contiguous(), frames
        known += adjacent_submissions(g2, per)[0]
    torch.cuda.synchronize()
    lanes = models.Lanes(net, n_lanes)
    lanes.set_coalesce(k)

    def submit(i):
        sub = order[i % len(order)]
        _, sub["res"] = lanes.submit(sub["tok"], sub["tok_lens"], sub["mel"], sub["f0"], sub["ema"], sub["ref_lens"], frame_cap=sub["cap"],
                                     out=sub["res"])
    for i in range(max(warmup // len(order) + 1, 4) * len(order)):
        submit(i)
    lanes.wait()
    els = []
    for _ in range(REPEATS):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            submit(i)
        lanes.wait()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        els.append(time.perf_counter() - t0)
        for i in range(steps, (steps // len(order) + 1) * len(order)):
            submit(i)
        lanes.wait()
    torch.cuda.synchronize()
    worst = max(float((sub["res"]["mel"][:, : 2 * sub["total"]] - sub["want_mel"]).abs().max()) for sub in order)
    offs_ok = all(torch.equal(sub["res"]["frame_off"].cpu(), sub["want_off"]) for sub in order)
    st = lanes.stats(0)
    merged = sum(lanes.merged_calls(i) for i in range(n_lanes))
    lanes.close()
    frames_step = 2.0 * sum(order[i % len(order)]["total"] for i in range(steps)) / steps            # mel frames of an average timed step
    # the same batches with the same durations as KNOWN counts through the same lanes (the headline's path on this ragged workload)
    lanes = models.Lanes(net, n_lanes)
    lanes.set_coalesce(k)

    def submit_k(i):
        sub = known[i % len(known)]
        lanes.submit(sub["tok"], sub["tok_lens"], sub["mel"], sub["f0"], sub["ema"], sub["ref_lens"], forced=sub["forced"], frames=sub["frames"],
                     out=sub["out"])
    for i in range(max(warmup // len(known) + 1, 4) * len(known)):
        submit_k(i)
    lanes.wait()
    els_k = []
    for _ in range(REPEATS):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            submit_k(i)
        lanes.wait()
        torch.cuda.synchronize()
        els_k.append(time.perf_counter() - t0)
        for i in range(steps, (steps // len(known) + 1) * len(known)):
            submit_k(i)
        lanes.wait()
    torch.cuda.synchronize()
    worst_k = max(float((sk["out"]["mel"] - so["want_mel"]).abs().max()) for sk, so in zip(known, order))
    lanes.close()
    # ... and the boundary of the reference as it is called (/root/reference/test.py:96-113): HOST arrays in, durations predicted on the
    # device, the mel and the frame offsets back on the host (as_lanes_submit_host with a frame capacity)
    pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
    hb = []
    for h, sub in zip(hosts, order):
        cat = lambda key, ax: np.concatenate([h[key][b] for b in range(per)], ax)
        hb.append(dict(tok=pin(cat("tokens", 0).astype(np.int32)), mel=pin(cat("mel", 1)), f0=pin(cat("f0", 1).reshape(-1)), ema=pin(cat("ema", 1)),
                       tok_lens=h["tok_lens"], ref_lens=h["ref_lens"], cap=sub["cap"], off=torch.zeros(per + 1, dtype=torch.int32).pin_memory(),
                       out=torch.zeros((net.rt.cfg.n_mels, 2 * sub["cap"]), dtype=torch.float32).pin_memory()))
    lanes = models.Lanes(net, n_lanes)
    lanes.set_coalesce(k)

    def submit_h(i):
        b = hb[i % len(hb)]
        lanes.submit_host(b["tok"], b["tok_lens"], b["mel"], b["f0"], b["ema"], b["ref_lens"], None, None, b["out"], frame_cap=b["cap"], frame_off=b["off"])
    for i in range(max(warmup // len(hb) + 1, 4) * len(hb)):
        submit_h(i)
    lanes.wait()
    els_h = []
    for _ in range(REPEATS):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            submit_h(i)
        lanes.wait()
        torch.cuda.synchronize()
        els_h.append(time.perf_counter() - t0)
        for i in range(steps, (steps // len(hb) + 1) * len(hb)):
            submit_h(i)
        lanes.wait()
    host_ok = all(torch.equal(b["off"], sub["want_off"]) and
                  torch.equal(b["out"][:, : 2 * sub["total"]], sub["res"]["mel"][:, : 2 * sub["total"]].cpu()) for b, sub in zip(hb, order))
    lanes.close()
    med = lambda v: sorted(v)[len(v) // 2]
    return dict(known_counts_same_durations=dict(ms_per_step=els_k[0] / steps * 1e3, ms_per_step_repeats=[e / steps * 1e3 for e in els_k],
                                                 max_abs_vs_read_back_path=worst_k,
                                                 note="the same batches, the predictor's integer durations handed over as forced durations with "
                                                      "host-known frame counts: the headline's path on this ragged workload"),
                capacity_overhead=med(els) / med(els_k),
                host_boundary=dict(ms_per_step_including_transfers=els_h[0] / steps * 1e3, ms_per_step_repeats=[e / steps * 1e3 for e in els_h],
                                   vs_resident_inputs=med(els_h) / med(els), copies_verified=bool(host_ok),
                                   note="as_lanes_submit_host with a frame capacity: host arrays in, the whole capacity slot of the mel and the "
                                        "frame offsets back; bitwise equal to the device-buffer submissions"),
                elapsed_s=els[0], ms_per_step=els[0] / steps * 1e3, ms_per_step_repeats=[e / steps * 1e3 for e in els], coalesce=k, lanes=n_lanes,
                mel_frames_per_step=frames_step, frames_per_s=frames_step * steps / els[0],
                frame_cap_per_submission=[sub["cap"] for sub in order], frames_predicted_per_submission=[sub["total"] for sub in order],
                merged_calls=merged, graph_launches_lane0=st["graph_launches"], eager_calls_lane0=st["eager_calls"],
                max_abs_vs_read_back_path=worst, frame_offsets_equal=bool(offs_ok), results_verified=bool(worst <= 3e-5 and offs_ok),
                note="as_lanes_submit with as_forward_io.frame_cap: durations predicted on the device, no host read-back, hipGraph replay, "
                     "k adjacent submissions per call (as_segments)")


def c4_check(net, host, mel_mine, mine, world, rank, dev, dist, dump=None):
    """merge the ranks' shards on rank 0 and compare with rank 0 running the whole global batch alone; `dump`: an .npz that gets the
    merged mel of three utterances (shortest, median, longest) for the caller to hold against the oracle (tests/test_multirank_gpu.py)"""
    from artspeech_amd import shard
    mels = [m.copy() for m in shard.split_utterances(mel_mine.cpu().numpy(), [2 * host["frames"][i] for i in mine])]
    if world > 1:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object((mine, mels), gathered, dst=0)
    else:
        gathered = [(mine, mels)]
    if rank != 0:
        return None
    merged = shard.merge_shards(gathered, len(host["frames"]))
    g_all = pack_inputs(host, list(range(len(host["frames"]))), dev)
    whole = Runner(net, g_all).step()["mel"].cpu().numpy()
    ref = shard.split_utterances(whole, [2 * f for f in host["frames"]])
    worst = max(float(np.abs(a - b).max()) for a, b in zip(merged, ref))
    if dump:
        order = sorted(range(len(merged)), key=lambda i: host["frames"][i])
        pick = [order[0], order[len(order) // 2], order[-1]]
        np.savez(dump, idx=np.asarray(pick), **{f"mel_{i}": merged[i] for i in pick})
    return dict(utterances=len(ref), shards=world, max_abs_sharded_vs_single_rank=worst,
                note="not bitwise: the GEMM's tile choice depends on a shard's total column count (bound 5e-5)")


def self_launch(n):
    """`python3 bench.py --gpus N` without a launcher around it: start `python -m torch.distributed.run --nproc-per-node N bench.py <the
    same arguments>` as a child process (rendezvous on 127.0.0.1, a free port), pass rank 0's JSON line through, return the launcher's
    exit code.  Called before anything in this process has initialised the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (the pool's driver supports dmabuf IPC only)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for ln in p.stdout:                                      # rank 0 prints the one line; the launcher's own chatter goes to stderr
        sys.stdout.write(ln)
        sys.stdout.flush()
    return p.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cpu-utts", type=int, default=96, help="utterances in the CPU-baseline sample, capped at ~20 s (0 = skip)")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-concurrency", action="store_true", help="run the independent branches back to back (profiling)")
    ap.add_argument("--no-extras", action="store_true", help="skip the MAS / C2 / C5 / transfer / surface lines (profiling runs)")
    ap.add_argument("--surface-only", action="store_true", help="print only the `surface` object (the test.py chain around the path): profiling runs")
    ap.add_argument("--config", default="C3", choices=["C3", "C5", "C2"], help="workload of the timed region (profiling runs; the headline is C3)")
    ap.add_argument("--global-batch", type=int, default=0, help="C4: ONE ragged batch of this many utterances sharded over the ranks")
    ap.add_argument("--c4-dump", default=None, help="with --global-batch: .npz for the merged mel of three utterances (tests hold them against the oracle)")
    ap.add_argument("--in-flight", type=int, default=4, help="batches in flight per GPU: consecutive steps are replayed on this many HIP streams "
                    "(each with its own plan and workspaces on the same weights); 1 = one step at a time (the step's latency)")
    ap.add_argument("--call-batches", type=int, default=1, help="profiling runs (C3): a step = ONE as_forward_test call over this many batches of 32 "
                    "(what a coalescing lane launches: --call-batches 2 = the headline arrangement's call)")
    ap.add_argument("--coalesce", type=int, default=2, help="batches of 32 a lane of the library launches as ONE call (as_lanes_set_coalesce: "
                    "adjacent buffers, no copy); with --in-flight F the library runs F / coalesce lanes.  1 = off")
    ap.add_argument("--lane-branches", action="store_true", help="lanes in flight keep their step's branches on side streams (default: a "
                    "lane in flight is one chain on one stream)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python3 bench.py --gpus N` on its own: this process has not touched the GPU (importing torch does not), so it starts the N
        # ranks as a fresh CHILD process group and relays rank 0's line and the launcher's exit code (never an exec from here on)
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dist = None
    if world > 1:                                       # the process group first, before anything touches the GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (AS_BENCH_TEST_ONE_GPU=1, testing only: every rank on GPU 0 with gloo -- exercises this launch path on a 1-GPU box)
        one_gpu = os.environ.get("AS_BENCH_TEST_ONE_GPU") == "1"
        if one_gpu:
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from artspeech_amd import models, shard, synth
    from artspeech_amd.weights import DEFAULT_STATS, load_distribution

    sd = synth.synth_state_dict(512, 64, seed=WEIGHT_SEED)
    model = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80), None, "second",
                               load_distribution(DEFAULT_STATS), dev)
    models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
    net = model.ArtsSpeech
    if args.no_concurrency:
        net.rt.set_serial(True)
    barrier = (lambda: dist.barrier()) if world > 1 else (lambda: None)
    if args.surface_only:
        print(json.dumps({"surface": bench_surface(net, dev)}), flush=True)
        return

    c4, mine = None, None
    if args.global_batch:
        # C4: one global ragged batch, length-sorted round-robin shards, merged on rank 0 and compared with rank 0 running it all
        host, _ = make_inputs(None, args.global_batch, vary=True)
        mine = shard.shard_indices(host["frames"], world, rank)
        g = pack_inputs(host, mine, dev)
        frames_total = 2 * sum(host["frames"])
        workload = f"C4: ONE ragged batch of {args.global_batch} utterances sharded over {world} GPUs (length-sorted round robin)"
    elif args.config == "C5":
        host, g = make_inputs(dev, 8, 1024, 1024, 200, seed0=DATA_SEED + 1000)
        frames_total = world * 2 * sum(host["frames"])
        workload = "C5: long form, batch 8 per GPU, 1024 tokens -> 2048 mel frames per utterance, T_ref=200 (profiling run)"
    elif args.config == "C2":
        host, g = make_inputs(dev, 1, 30, 75, 150, seed0=DATA_SEED + 1000)
        frames_total = world * 2 * sum(host["frames"])
        workload = "C2: one utterance, 30 tokens -> 150 mel frames, T_ref=150 (profiling run)"
    elif args.call_batches > 1:
        host = merge_hosts([make_inputs(None, seed0=DATA_SEED + 100 * i)[0] for i in range(args.call_batches)])
        g = pack_inputs(host, list(range(len(host["frames"]))), dev)
        frames_total = world * 2 * sum(host["frames"])
        workload = (f"C3 as a coalescing lane launches it: ONE call over {args.call_batches} batches of 32 utterances (40 tokens -> 200 mel frames each, "
                    "T_ref=200), forced integer durations (profiling run: a step here is one such call)")
    else:
        host, g = make_inputs(dev)
        frames_total = world * 2 * sum(host["frames"])
        workload = ("C3: LibriTTS-like batch=32 per GPU, 40 tokens -> 200 mel frames per utterance, T_ref=200, "
                    "full predictor+decoder path, forced integer durations, synthetic weights seed 3407")
    runner = Runner(net, g)
    mel_first = runner.step()["mel"].clone()
    torch.cuda.synchronize()
    run = runner.step if args.no_graph else runner.capture()
    elapsed = runner.timed(run, args.steps, args.warmup, barrier)
    single_ms = elapsed / args.steps * 1e3                               # one step at a time: the step's latency
    if not args.no_graph:
        torch.cuda.synchronize()
        assert torch.equal(runner.out["mel"], mel_first), "graph replay changed the result"
    # Consecutive steps are independent batches: with two in flight (a second plan + workspaces on the same weights, its own stream) the
    # tail round of one batch's kernels is filled by the other batch's -- what a server does; the K timed steps alternate between them.
    # Every lane has its OWN batch (other seeds, same geometry) in its own buffers: nothing a lane reads is warm from the other's pass.
    one_chain_ms, chain_vs_side, rep_fl = None, None, []
    n_fl = 1 if (args.no_graph or args.global_batch or args.call_batches > 1) else max(1, args.in_flight)
    in_flight_note = None
    lanes = [(runner, run, torch.cuda.Stream())]
    if n_fl > 1:
        firsts = []
        lanes = []
        for i in range(n_fl):
            if i == 0:
                gi = g
            elif args.config == "C5":
                _, gi = make_inputs(dev, 8, 1024, 1024, 200, seed0=DATA_SEED + 1000 + 100 * i)
            elif args.config == "C2":
                _, gi = make_inputs(dev, 1, 30, 75, 150, seed0=DATA_SEED + 1000 + 100 * i)
            else:
                _, gi = make_inputs(dev, seed0=DATA_SEED + 100 * i)
            twin = net.replica()
            if not args.lane_branches:
                twin.rt.set_serial(True)                     # a lane in flight: one chain on one stream
            r2 = Runner(twin, gi)
            firsts.append(r2.step()["mel"].clone())
            lanes.append((r2, r2.capture(), torch.cuda.Stream()))
        # (a chain's conv GEMMs share launches across branches -- as_plan_set_merge -- and may then run on another tile shape than alone:
        # the same arithmetic in another order of partial sums)
        chain_vs_side = float((firsts[0] - mel_first).abs().max())
        assert chain_vs_side <= 3e-5, f"a step as one chain and the step with its branches on side streams differ by {chain_vs_side}"
        # one chain ALONE (the latency of a batch on a merging serial plan, beside single_ms: the same batch with its branches on side streams)
        with torch.cuda.stream(lanes[0][2]):
            one_chain_ms = lanes[0][0].timed(lanes[0][1], max(args.steps // 4, 5), 3, lambda: None) / max(args.steps // 4, 5) * 1e3
        it = [0]

        def run_lanes():
            _, fn, st = lanes[it[0] % n_fl]
            it[0] += 1
            with torch.cuda.stream(st):
                fn()
        rep_fl = []
        elapsed_fl = runner.timed(run_lanes, args.steps, args.warmup, barrier, repeats=rep_fl)
        # every lane must still produce the bits of its own first eager step; if not, the in-flight number is discarded and the line
        # reports the one-at-a-time run (and says so)
        lanes_ok = all(torch.equal(r2.out["mel"], f) for (r2, _, _), f in zip(lanes, firsts))
        if world > 1:                                        # ONE decision for all ranks: what follows (bench_coalesced, transfers) holds collective barriers
            ok_t = torch.tensor([1 if lanes_ok else 0], dtype=torch.int32, device="cpu" if dist.get_backend() == "gloo" else dev)
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
            lanes_ok = bool(int(ok_t.item()))
        if lanes_ok:
            elapsed = elapsed_fl
        else:
            in_flight_note = "results of the batches in flight differed from the one-at-a-time result: in-flight timing discarded"
            n_fl = 1
            lanes = lanes[:1]
    # The same K steps as K submissions through the LIBRARY's lanes with coalescing: a lane launches `--coalesce` adjacent batches of 32 as one
    # call.  Every rank; the faster verified arrangement is the line's `value` (both are in the line).
    coal, coal_first, elapsed_lanes32 = None, None, None
    if n_fl > 1 and args.coalesce > 1 and args.config == "C3" and not args.global_batch and not args.no_graph and not args.lane_branches:
        kc = args.coalesce
        ncl = max(1, n_fl // kc)
        hosts = [host] + [make_inputs(None, seed0=DATA_SEED + 100 * i)[0] for i in range(1, kc * ncl)]
        coal, coal_first = bench_coalesced(net, hosts, kc, ncl, args.steps, args.warmup, barrier)
        c_el, l_el, c_ok = coal["elapsed_s"], elapsed, coal["results_verified"]
        if world > 1:                                        # ONE decision for all ranks (what follows holds collectives): the slowest rank's times
            cpu = dist.get_backend() == "gloo"
            t = torch.tensor([c_el, l_el, 0.0 if c_ok else 1.0], dtype=torch.float64, device="cpu" if cpu else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c_el, l_el, c_ok = float(t[0].item()), float(t[1].item()), float(t[2].item()) == 0.0
        coal["adopted_as_value"] = bool(c_ok and c_el < l_el)
        if coal["adopted_as_value"]:
            elapsed_lanes32 = elapsed
            elapsed = coal["elapsed_s"]
    predicted = None
    if coal is not None and rank == 0:
        try:
            predicted = bench_predicted(net, hosts, kc, ncl, args.steps, args.warmup)
            # per mel frame against the line's value (the predicted durations of these inputs do not add up to the forced 100 frames per utterance)
            predicted["ms_per_mel_frame_vs_value"] = (predicted["ms_per_step"] / predicted["mel_frames_per_step"]) / (elapsed / args.steps * 1e3 / (frames_total / world))
        except Exception as e:                               # (an extra: never costs the headline line)
            predicted = {"error": repr(e)[:300]}
    native = None
    if n_fl > 1 and rank == 0 and not args.no_extras:
        native = bench_native_lanes(net, [r.g for r, _, _ in lanes], firsts, args.steps, args.warmup)
    # the same K steps with the host <-> device copies inside the timed region (every rank; max over ranks like the headline)
    transfers = None
    if not args.no_graph and not args.global_batch and not args.no_extras:
        transfers = bench_transfers(lanes, args.steps, min(args.warmup, 10), barrier)
    if world > 1:
        t = torch.tensor([elapsed, transfers["elapsed_s"] if transfers else 0.0], dtype=torch.float64,
                         device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        if transfers:
            transfers["elapsed_s"] = float(t[1].item())
            transfers["ms_per_step_including_transfers"] = transfers["elapsed_s"] / args.steps * 1e3
    if transfers:
        transfers["frames_per_s_including_transfers"] = frames_total * args.steps / transfers["elapsed_s"]
        # (the copies are timed around the replays of 32-utterance graphs on the script's own streams: compared with THAT arrangement's
        #  resident-input time, which is `elapsed` unless the coalesced lanes' figure became the line's value)
        transfers["vs_resident_inputs"] = transfers["elapsed_s"] / (elapsed_lanes32 or elapsed)
        transfers["arrangement"] = "batches of 32, one hipGraph replay each (ms_per_step_lanes_of_32)" if elapsed_lanes32 else "as the line's value"
    if coal and coal["adopted_as_value"]:
        # the line's value is the coalesced arrangement: its copy-inclusive figure is the library's own host boundary (as_lanes_submit_host),
        # the K steps with the copies inside the timed region; the 4 x 32 arrangement's figure stays beside it
        hbnd = dict(coal["host_boundary"])
        if world > 1:
            t = torch.tensor([hbnd["elapsed_s"]], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            hbnd["elapsed_s"] = float(t[0].item())
            hbnd["ms_per_step_including_transfers"] = hbnd["elapsed_s"] / args.steps * 1e3
            hbnd["vs_resident_inputs"] = hbnd["elapsed_s"] / elapsed                 # (the slowest rank's first passes)
        hbnd["frames_per_s_including_transfers"] = frames_total * args.steps / hbnd["elapsed_s"]
        hbnd["lanes_of_32"] = transfers
        transfers = hbnd
    if args.global_batch:
        c4 = c4_check(net, host, runner.out["mel"], mine, world, rank, dev, dist, dump=args.c4_dump)

    # ---- phase times of one eager step with the branches concurrent (HIP events between the phases, on the calling stream)
    prunner = Runner(net, g)
    phase = [net.rt.phase_ms(prunner.step) for _ in range(3)][-1]
    # the kernel classes AS THE TIMED REGION LAUNCHES THEM: with the coalesced arrangement adopted, one call over its 32 k utterances
    per_call = args.call_batches
    g_roof = g
    if coal and coal["adopted_as_value"]:
        per_call = coal["coalesce"]
        g_roof = pack_inputs(merge_hosts(hosts[:per_call]), list(range(per_call * len(host["frames"]))), dev)
    kern = profile_classes(net, Runner(net, g_roof))
    net.rt.set_serial(args.no_concurrency)
    gemm_tflops, gemm_peak = gemm_roofline(kern, 3)
    k0 = kern["conv_gemm"]
    hbm_ms = sum(kern[c]["ms_per_step"] for c in HBM_GROUP if c in kern)
    hbm_gb = sum(kern[c]["gbyte_per_step"] for c in HBM_GROUP if c in kern)

    # HBM traffic of the dominant kernel cannot be read live (PMC needs rocprofv3): it comes from the committed counter pass of this
    # same command (profiles/latest_pmc_traffic.json, scripts/gpu_profile.sh), and only when that pass was taken from THIS build
    traffic, traffic_src = None, "no PMC profile of this build (profiles/latest_pmc_traffic.json was taken from other kernel sources)"
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc_traffic.json")))
        if pj.get("source_id") == source_id() and pj.get("batches_per_call", 1) == per_call:
            traffic, traffic_src = pj["conv_gemm_hbm_bytes_per_launch"], pj["source"]
    except Exception:
        pass

    sus_reg, sus_lds = mfma_sustained()
    ms_per_step = elapsed / args.steps * 1e3
    # The roofline fractions are reported from the committed rocprofv3 kernel trace of THIS build when there is one
    # (profiles/latest_trace_classes.json: sum of kernel durations per class and step, scripts/make_profiles.py) -- a trace counts a kernel's
    # own duration, nothing of the brackets around it -- and from this run's HIP events otherwise; the event figures stay beside them.
    trace_cls, trace_src = None, "no kernel trace of this build under profiles/ (HIP events of this run, bracket cost removed)"
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "latest_trace_classes.json")))
        if tj.get("source_id") == source_id() and args.config == "C3" and not args.global_batch and tj.get("batches_per_call", 1) == per_call:
            trace_cls, trace_src = tj["classes"], f"{tj['file']} ({tj['command']}; {tj['steps_in_trace']} steps in the trace)"
    except Exception:
        pass
    gemm_tflops_ev = gemm_tflops
    hbm_ms_ev = hbm_ms
    tc_ms = None
    if trace_cls:                                                  # (both sides per CALL of per_call batches)
        try:
            tc_ms = lambda c: trace_cls[c].get("ms_per_call", trace_cls[c]["ms_per_step"] * per_call)   # (files of rounds 1-4: per step = per call)
            gemm_tflops = k0["gflop_per_step"] / tc_ms("conv_gemm")
            hbm_ms = sum(tc_ms(c) for c in HBM_GROUP if c in trace_cls)
        except Exception:                                          # (a committed summary this script cannot read never costs the line)
            trace_cls, trace_src, gemm_tflops, hbm_ms = None, "the committed kernel-trace summary is unreadable (HIP events of this run)", gemm_tflops_ev, hbm_ms_ev
    value = frames_total * args.steps / elapsed
    # the K steps of the arrangement that became `value`, timed REPEATS times back to back inside this run (rank-local; `ms_per_step` is
    # the first of them, max over ranks): min / median / max say what a 20-step measurement is worth
    reps_ms = (coal["ms_per_step_repeats"] if (coal and coal["adopted_as_value"]) else [e / args.steps * 1e3 for e in rep_fl]) or [ms_per_step]
    rep_line = dict(values=reps_ms, min=min(reps_ms), median=sorted(reps_ms)[len(reps_ms) // 2], max=max(reps_ms))
    line = {
        "metric": "mel frames/sec (whole job; per-GPU = value / n_gpus), acoustic-model inference path, batch 32 x 200-frame utterances",
        "value": value, "unit": "mel frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "ms_per_step_repeats": rep_line, "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None,
        "dtype": "f32 (f16x3 split-operand MFMA, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": workload, "global_batch": args.global_batch or len(g["frames"]) * world, "frames_per_step": frames_total,
                   "parallelism": f"batch-shard x{world}, no collectives",
                   "launch": "eager (C ABI as_forward_test)" if args.no_graph else "hipGraph replay of one as_forward_test call",
                   "in_flight": n_fl, "lane_streams": "branches on side streams" if (args.lane_branches or n_fl == 1) else "one chain per lane",
                   "arrangement": (f"coalesced {coal['coalesce']}x32: steps are submissions of ONE batch of 32 (as_lanes_submit); each of the "
                                   f"library's {coal['lanes']} lanes launches {coal['coalesce']} adjacent batches as one as_forward_test call")
                                  if (coal and coal["adopted_as_value"]) else
                                  (f"{n_fl} batches of 32 in flight, one hipGraph replay of one as_forward_test call each" if n_fl > 1 else "one batch at a time")},
        "ms_per_step_lanes_of_32": (elapsed_lanes32 / args.steps * 1e3) if elapsed_lanes32 else None,   # (the 4 x 32 arrangement, when `value` is the coalesced one's)
        "coalesced": coal,
        "predicted": predicted,
        "ms_per_step_one_in_flight": single_ms,
        "ms_per_step_one_chain_alone": one_chain_ms,
        "chain_vs_side_streams_max_abs": chain_vs_side,
        "in_flight_note": in_flight_note,
        "rtf": (elapsed / args.steps) / (frames_total * FRAME_SEC),
        "x_realtime_per_gpu": (value / world) * FRAME_SEC,
        "roofline": {"bound": "mfma",
                     "kernel": "conv_gemm_h3_kernel (implicit-GEMM conv, fp16 matrix cores, two-way split operands h + l, 3 products per fp32 "
                               "product, fp32 accumulate; both operands staged by LDS-DMA from producer-written images)",
                     "achieved": gemm_tflops, "peak": gemm_peak, "unit": "TFLOP/s", "frac": gemm_tflops / gemm_peak,
                     # where `achieved` / `frac` come from: "committed_trace" = the rocprofv3 kernel trace of THIS build committed under
                     # profiles/ (matched by the kernel sources' id; not measured by this run), "events_this_run" = HIP events of this run
                     "frac_source": "committed_trace" if trace_cls else "events_this_run",
                     "time_source": trace_src,
                     "achieved_by_events": gemm_tflops_ev, "frac_by_events": gemm_tflops_ev / gemm_peak,      # (always this run's)
                     # the launches the fraction is about: one as_forward_test call over `batches_per_call` batches of 32 (the timed region's)
                     "batches_per_call": per_call, "utterances_per_call": len(g_roof["frames"]),
                     "gemm_ms_per_step_by_trace": tc_ms("conv_gemm") / per_call if trace_cls else None,
                     "gemm_ms_per_step_by_events": k0["ms_per_step"] / per_call,
                     "peak_basis": "dense fp16 MFMA 2516.6 TFLOP/s / 3 matrix-core products per fp32 product (achieved = algorithmic fp32 flop); "
                                   "round 1 ran six bf16 products per fp32 product (ceiling 419.4)",
                     # measured live on this device: a bare loop of the kernel's own MFMA pattern on random operands (no memory traffic, no
                     # barriers).  The chip lowers its clock under matrix-core load on non-trivial data: `peak` above is the data sheet's.
                     "sustained_on_random_operands": {
                         "mfma_f16_tflops_operands_in_registers": sus_reg, "mfma_f16_tflops_operands_from_lds": sus_lds,
                         "as_f16x3_tflops_from_lds": sus_lds / 3.0, "frac_of_nominal_peak": sus_lds / PEAK_F16_MFMA_TFLOPS,
                         "kernel_frac_of_sustained": gemm_tflops / (sus_lds / 3.0)},
                     "frac_of_fp32_mfma_peak": gemm_tflops / PEAK_F32_MFMA_TFLOPS,
                     "frac_of_round1_bf16x6_ceiling": gemm_tflops / PEAK_X6_TFLOPS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "avg_launch_ms": k0["ms_per_step"] / max(k0["launches_per_step"], 1), "launches_per_step": k0["launches_per_step"],
                     "event_bracket_overhead_us": bracket_overhead_ms() * 1e3,
                     "achieved_with_bracket_overhead": k0["gflop_per_step"] / k0["ms_per_step_bracketed"],
                     "algorithmic_gflop_per_step": k0["gflop_per_step"] / per_call},
        "roofline_hbm": {"bound": "hbm", "kernels": "AdaIN / LayerNorm operand-image writers, pooling, expansion, im2col, reference features "
                                                    "(classes adain + layernorm + other of the event profiler)",
                         "achieved": hbm_gb / hbm_ms if hbm_ms else None, "peak": PEAK_HBM_TBS, "unit": "TB/s",
                         "frac": hbm_gb / hbm_ms / PEAK_HBM_TBS if hbm_ms else None,
                         "frac_source": "committed_trace" if trace_cls else "events_this_run",
                         "batches_per_call": per_call,
                         "algorithmic_gbyte_per_step": hbm_gb / per_call, "ms_per_step": hbm_ms / per_call, "time_source": trace_src,
                         "ms_per_step_by_events": hbm_ms_ev / per_call, "frac_by_events": hbm_gb / hbm_ms_ev / PEAK_HBM_TBS if hbm_ms_ev else None,
                         "launches_per_step": sum(kern[c]["launches_per_step"] for c in HBM_GROUP if c in kern)},
        "launches_per_step": {c: kern[c]["launches_per_step"] for c in kern},       # (per call of `roofline.batches_per_call` batches, like kernel_classes)
        "forward_calls_in_process": None,
        "kernel_classes": kern,
        "phase_ms_eager": dict(zip(["features", "encoders_towers_duration", "predictors", "decoder"], [round(v, 3) for v in phase])),
    }
    if c4 is not None:
        line["c4_shard_check"] = c4
    extras = rank == 0 and not args.no_extras and not args.global_batch and args.config == "C3"
    if transfers:
        line["transfers"] = transfers
    if native:
        line["lanes_native"] = native
    if extras:
        line["mas"] = bench_mas(dev)
        line["configs"] = {
            "C2": bench_config(net, dev, "C2 (LJSpeech-like latency)", 1, 30, 75, 150, 50, n_prod_modes=(3, 1), lanes=4),
            "C5": bench_config(net, dev, "C5 (long form)", 8, 1024, 1024, 200, 10, lanes=4),
        }
        try:
            line["surface"] = bench_surface(net, dev)
        except Exception as e:                          # (an extra: never costs the headline line)
            line["surface"] = {"error": repr(e)[:300]}
    if rank == 0 and args.cpu_utts > 0 and not args.global_batch and args.config == "C3":
        cb, outs = cpu_baseline(host, sd, args.cpu_utts)
        line["cpu_baseline"] = cb
        fo = np.concatenate([[0], np.cumsum([2 * f for f in g["frames"]])])
        err = max(float((mel_first[:, fo[b]:fo[b + 1]].cpu() - outs[b]["mel"]).abs().max()) for b in range(min(len(outs), B)))
        line["parity_mel_max_abs_vs_oracle"] = err
        if coal_first is not None:                       # the same batch as it came out of a coalesced call
            line["parity_coalesced_mel_max_abs_vs_oracle"] = max(
                float((coal_first[:, fo[b]:fo[b + 1]].cpu() - outs[b]["mel"]).abs().max()) for b in range(min(len(outs), B)))
    line["forward_calls_in_process"] = FORWARD_CALLS[0] if args.no_graph else None      # (what a kernel trace of an eager run holds)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()                                  # the other ranks stay until rank 0 has finished its extra lines
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
