#!/usr/bin/env python3
"""Headline benchmark: mel frames/s of the acoustic-model inference path (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path -- text encoder + articulatory encoder + style towers +
duration predictor + integer alignment expansion + F0/energy/TV predictors + AdaIN mel decoder -- over one
batch of 32 synthetic utterances (config C3 of SURVEY.md section 8: N = 40 tokens, forced integer durations
summing to M = 100 => 200 mel frames per utterance, T_ref = 200; full-size model, seeded synthetic weights).
Inputs are resident in HBM before the timed region.  With N > 1 every rank runs the same workload on its own
GPU (utterance batches shard embarrassingly; no data-path collective) => weak scaling; the only
torch.distributed use is the barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" for the dominant kernel (the implicit-GEMM conv on the
matrix cores -- bf16x6 split-operand arithmetic, fp32-accurate, or fp32 MFMAs with AS_GEMM_IMPL=f32 -- timed with
HIP events on its launch stream in an instrumented pass of the same step) and "cpu_baseline"
(the oracle's CPU restatement timed on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

B, N_TOK, M_HALF, T_REF = 32, 40, 100, 200          # config C3
FRAMES_PER_UTT = 2 * M_HALF
FRAME_SEC = 300.0 / 24000.0                          # hop 300 @ 24 kHz (test.py:40)
WEIGHT_SEED, DATA_SEED = 3407, 1234
PEAK_F32_MFMA_TFLOPS = 157.3                         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 (64 cycles per SIMD)
# f16x3: every fp32 product is three fp16 MFMA products; dense fp16/bf16 peak 256 CU x 4 SIMD x 1024 flop/clk x 2.4 GHz = 2516.6
PEAK_F16_MFMA_TFLOPS = 2516.6
PEAK_H3_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0          # fp32-equivalent ceiling of the two-way-split GEMM: 838.9
PEAK_X6_TFLOPS = PEAK_F16_MFMA_TFLOPS / 6.0          # round 1's arithmetic (bf16x6): 419.4 -- kept for comparison across rounds
CLASSES = ["conv_gemm", "adain", "layernorm", "attention", "lstm", "mas", "other"]


def make_inputs(dev):
    from artspeech_amd import synth
    from artspeech_amd.weights import DEFAULT_STATS
    toks, mels, f0s, emas = [], [], [], []
    for b in range(B):
        toks.append(synth.synth_tokens(N_TOK, DATA_SEED + b))
        mel, f0, ema = synth.synth_ref_features(T_REF, DATA_SEED + b)
        mels.append(mel)
        f0s.append(f0 * np.float32(DEFAULT_STATS["pitch"][3]) + np.float32(DEFAULT_STATS["pitch"][2]))
        emas.append(ema * np.asarray(DEFAULT_STATS["EMA"][3], np.float32)[:, None] + np.asarray(DEFAULT_STATS["EMA"][2], np.float32)[:, None])
    forced = np.full(N_TOK, 2, np.int32)
    forced[::2] = 3                                    # 20*3 + 20*2 = 100 half-rate frames
    assert forced.sum() == M_HALF
    host = dict(tokens=toks, mel=mels, f0=f0s, ema=emas, forced=forced)
    if dev is None:
        return host, None
    g = dict(
        tok=torch.from_numpy(np.concatenate(toks)).to(dev, torch.int32),
        tok_lens=[N_TOK] * B, ref_lens=[T_REF] * B,
        mel=torch.from_numpy(np.concatenate(mels, 1)).to(dev).contiguous(),
        f0=torch.from_numpy(np.concatenate(f0s, 1).astype(np.float32)).to(dev).contiguous(),
        ema=torch.from_numpy(np.concatenate(emas, 1).astype(np.float32)).to(dev).contiguous(),
        forced=torch.from_numpy(np.tile(forced, B)).to(dev, torch.int32),
        frames=[M_HALF] * B)
    return host, g


def cpu_baseline(host, sd, n_utt):
    """The oracle (CPU restatement of the reference, oracle/acoustic.py) on this box's host cores."""
    from artspeech_amd.weights import DEFAULT_STATS, fold_state_dict, load_distribution
    from oracle import acoustic
    cores = min(os.cpu_count() or 1, 16)              # small per-utterance ops: more threads only add sync cost
    torch.set_num_threads(cores)
    W = fold_state_dict(sd)
    dist = load_distribution(DEFAULT_STATS)
    args = [(torch.from_numpy(host["tokens"][b % B]), torch.from_numpy(host["mel"][b % B]), torch.from_numpy(host["f0"][b % B].astype(np.float32)),
             torch.from_numpy(host["ema"][b % B].astype(np.float32))) for b in range(n_utt)]
    acoustic.forward_test(W, *args[0], dist, forced_dur=host["forced"])          # warm-up
    t0 = time.perf_counter()
    outs = []
    for a in args:                                     # bounded: stop after ~20 s of CPU work
        outs.append(acoustic.forward_test(W, *a, dist, forced_dur=host["forced"]))
        if time.perf_counter() - t0 > 20.0:
            break
    dt = time.perf_counter() - t0
    n_utt = len(outs)
    return dict(value=n_utt * FRAMES_PER_UTT / dt, unit="mel frames/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n_utt} utterances of the same C3 workload, one at a time (the reference is batch-1), "
                       f"{dt:.1f} s of CPU work, torch {torch.__version__} fp32"), outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cpu-utts", type=int, default=96, help="utterances in the CPU-baseline sample, capped at ~20 s (0 = skip)")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-concurrency", action="store_true", help="run the independent branches back to back (profiling)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from artspeech_amd import _lib, models, synth
    from artspeech_amd.weights import DEFAULT_STATS, load_distribution

    sd = synth.synth_state_dict(512, 64, seed=WEIGHT_SEED)
    model = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80), None, "second",
                               load_distribution(DEFAULT_STATS), dev)
    models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
    net = model.ArtsSpeech
    host, g = make_inputs(dev)
    if args.no_concurrency:
        net.rt.set_serial(True)

    def step(out=None):
        return net.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], forced=g["forced"],
                                  frames_hint=g["frames"], out=out)

    out = step()                                            # also uploads the geometry tables (one blocking upload per new geometry)
    torch.cuda.synchronize()
    mel_first = out["mel"].clone()

    graph = None
    if not args.no_graph:
        # the step is a fixed sequence of launches on one stream with no host sync: capture it once
        graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            gout = step()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=s):
                step(out=gout)
        torch.cuda.current_stream().wait_stream(s)
        run = graph.replay
    else:
        gout = None
        run = step

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if graph is not None:
        assert torch.equal(gout["mel"], mel_first), "graph replay changed the result"

    # ---- phase times of one eager step with the branches concurrent (HIP events between the phases, on the calling stream)
    phase = [net.rt.phase_ms(step) for _ in range(3)][-1]

    # ---- instrumented pass: per-kernel-class time with HIP events on the launch stream (eager launches)
    # (branches that normally overlap on side streams run back to back here, so a kernel's event-bracketed duration is
    #  its own and not that of whatever shared the chip with it)
    L = _lib.lib()
    prof_steps = 3
    net.rt.set_serial(True)
    step()
    torch.cuda.synchronize()
    L.as_prof_enable(1)
    for _ in range(prof_steps):
        step()
    n = len(CLASSES)
    ms, fl, by = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)()
    cnt = (ctypes.c_int32 * n)()
    _lib.check(L.as_prof_collect(ms, fl, by, cnt, n), "as_prof_collect")
    L.as_prof_enable(0)
    net.rt.set_serial(args.no_concurrency)
    kern = {CLASSES[i]: dict(ms_per_step=ms[i] / prof_steps, launches_per_step=cnt[i] // prof_steps,
                             gflop_per_step=fl[i] / prof_steps / 1e9) for i in range(n) if cnt[i]}
    gemm_ms = ms[0] / max(cnt[0], 1)
    gemm_tflops = (fl[0] / max(cnt[0], 1)) / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0

    # HBM traffic of the dominant kernel cannot be read live (PMC needs rocprofv3): it comes from the committed
    # counter pass of this same command (profiles/latest_pmc_traffic.json, made by scripts/gpu_profile.sh)
    traffic, traffic_src = None, None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc_traffic.json")))
        traffic, traffic_src = pj["conv_gemm_hbm_bytes_per_launch"], pj["source"]
    except Exception:
        pass

    from artspeech_amd import ops as _ops
    gemm_impl = _ops.GEMM_IMPL
    n_prod = 1 if gemm_impl == "h1" else 3
    gemm_kernel = ("conv_gemm_h3_kernel (implicit-GEMM conv, fp16 matrix cores, two-way split operands h + l, %d product%s per fp32 "
                   "product, fp32 accumulate; both operands pre-split and staged by LDS-DMA)" % (n_prod, "" if n_prod == 1 else "s"))
    gemm_peak = PEAK_F16_MFMA_TFLOPS / n_prod
    gemm_peak_basis = f"dense fp16 MFMA 2516.6 TFLOP/s / {n_prod} matrix-core products per fp32 product (achieved = algorithmic fp32 flop)"
    frames_per_step = B * FRAMES_PER_UTT
    ms_per_step = elapsed / args.steps * 1e3
    value = world * frames_per_step * args.steps / elapsed
    line = {
        "metric": "mel frames/sec (whole job; per-GPU = value / n_gpus), acoustic-model inference path, batch 32 x 200-frame utterances",
        "value": value, "unit": "mel frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 operands, f32 accumulate (AS_GEMM_IMPL=h1)" if n_prod == 1 else "f32 (f16x3 split-operand MFMA, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": "C3: LibriTTS-like batch=32 per GPU, 40 tokens -> 200 mel frames per utterance, T_ref=200, "
                               "full predictor+decoder path, forced integer durations, synthetic weights seed 3407",
                   "global_batch": B * world, "frames_per_utt": FRAMES_PER_UTT, "parallelism": f"batch-shard x{world}, no collectives",
                   "launch": "eager" if graph is None else "hipGraph replay"},
        "rtf": (elapsed / args.steps) / (frames_per_step * FRAME_SEC),
        "x_realtime_per_gpu": (value / world) * FRAME_SEC,
        "roofline": {"bound": "mfma", "kernel": gemm_kernel, "achieved": gemm_tflops, "peak": gemm_peak, "unit": "TFLOP/s",
                     "frac": gemm_tflops / gemm_peak, "peak_basis": gemm_peak_basis,
                     "frac_of_fp32_mfma_peak": gemm_tflops / PEAK_F32_MFMA_TFLOPS,
                     "frac_of_round1_bf16x6_ceiling": gemm_tflops / PEAK_X6_TFLOPS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "avg_launch_ms": gemm_ms, "launches_per_step": int(cnt[0] // prof_steps),
                     "algorithmic_gflop_per_step": fl[0] / prof_steps / 1e9},
        "kernel_classes": kern,
        "phase_ms_eager": dict(zip(["features", "encoders_towers_duration", "predictors", "decoder"], [round(v, 3) for v in phase])),
    }
    if rank == 0 and args.cpu_utts > 0:
        cb, outs = cpu_baseline(host, sd, args.cpu_utts)
        line["cpu_baseline"] = cb
        err = max(float((mel_first[:, b * FRAMES_PER_UTT:(b + 1) * FRAMES_PER_UTT].cpu() - outs[b]["mel"]).abs().max())
                  for b in range(min(len(outs), B)))
        line["parity_mel_max_abs_vs_oracle"] = err
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
