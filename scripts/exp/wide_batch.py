#!/usr/bin/env python3
"""Round 5, VERDICT item 1 step 1: is ONE wide as_forward_test call over k x 32 utterances faster per 32 utterances than k calls of
32 in flight?  Every arrangement keeps its batches resident; time per 32 utterances = wall time of the replays / (k x chains x reps).

  arrangement (k, c): c merged serial chains in flight, each a hipGraph of ONE as_forward_test over 32 k utterances (C3 geometry).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from artspeech_amd import models, synth
    from artspeech_amd.weights import DEFAULT_STATS, load_distribution
    sd = synth.synth_state_dict(512, 64, seed=bench.WEIGHT_SEED)
    model = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80), None, "second",
                               load_distribution(DEFAULT_STATS), dev)
    models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
    net = model.ArtsSpeech
    arrangements = [tuple(int(v) for v in a.split("x")) for a in os.environ.get("ARR", "1x4,2x1,2x2,4x1,4x2,2x3,8x1,1x1").split(",")]
    reps = int(os.environ.get("REPS", "40"))
    res = []
    if os.environ.get("CLASSES"):
        # per-class kernel time (HIP events, one chain) per 32 utterances at 32 k utterances per call: where does a wide call spend it?
        for k in (int(v) for v in os.environ["CLASSES"].split(",")):
            _, g = bench.make_inputs(dev, 32 * k)
            twin = net.replica()
            kern = bench.profile_classes(twin, bench.Runner(twin, g))
            row = {c: round(v["ms_per_step"] / k, 4) for c, v in kern.items()}
            row["gemm_tflops"] = round(kern["conv_gemm"]["gflop_per_step"] / kern["conv_gemm"]["ms_per_step"], 1)
            row["launches"] = sum(v["launches_per_step"] for v in kern.values())
            print("classes per 32 utt at", 32 * k, "per call:", row, flush=True)
            del twin
            torch.cuda.empty_cache()
    if os.environ.get("NATIVE"):
        # the same arrangements through the library's own lanes (as_lanes: c streams created back to back -- consecutive HIP streams land on
        # different hardware queues; two torch streams of a pool may share one, and then nothing of the two chains overlaps)
        for k, c in arrangements:
            batches = [bench.make_inputs(dev, 32 * k, seed0=bench.DATA_SEED + 100 * i)[1] for i in range(c)]
            chain = net.replica()
            chain.rt.set_serial(True)
            firsts = [bench.Runner(chain, b).step()["mel"].clone() for b in batches]
            best = None
            for _ in range(2):
                nl = bench.bench_native_lanes(net, batches, firsts, reps * c, 0)
                best = nl["ms_per_step"] if best is None else min(best, nl["ms_per_step"])
            print(dict(utt_per_call=32 * k, chains=c, ms_per_32_utt=round(best / k, 4), native_lanes=True, bitwise=nl["results_bitwise_equal"]), flush=True)
            del chain, batches, firsts
            torch.cuda.empty_cache()
        return
    for k, c in arrangements:
        lanes = []
        for i in range(c):
            _, g = bench.make_inputs(dev, 32 * k, seed0=bench.DATA_SEED + 100 * i)
            twin = net.replica()
            twin.rt.set_serial(True)
            r = bench.Runner(twin, g)
            r.step()
            lanes.append((r, r.capture(), torch.cuda.Stream()))
        torch.cuda.synchronize()

        def once(n):
            for j in range(n):
                _, fn, st = lanes[j % c]
                with torch.cuda.stream(st):
                    fn()
        best = None
        for _ in range(3):
            once(2 * c)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            once(reps * c)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (reps * c * k) * 1e3
            best = dt if best is None else min(best, dt)
        res.append(dict(utt_per_call=32 * k, chains=c, ms_per_32_utt=round(best, 4)))
        print(res[-1], flush=True)
        del lanes
        torch.cuda.empty_cache()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
