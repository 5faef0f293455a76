"""GPU: as_lanes (csrc/lanes.hip) -- the throughput arrangement as a piece of the library: batches submitted round robin to N serial plans
on their own streams (eager, captured, then graph-replayed) produce the bits of the same batches run one at a time through
as_forward_test; predicted durations (frame counts read back, workspace re-sized on AS_ENOSPC) included."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _alone(net):
    """as_forward_test run ALONE on a plan like a lane's (serial, merging): what a lane's results are held against bit for bit"""
    twin = net.replica()
    twin.rt.set_serial(True)
    return twin


def _net(dev):
    import bench
    from artspeech_amd import models, synth
    from artspeech_amd.weights import DEFAULT_STATS, load_distribution
    sd = synth.synth_state_dict(512, 64, seed=bench.WEIGHT_SEED)
    model = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80), None, "second", load_distribution(DEFAULT_STATS), dev)
    models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
    return model.ArtsSpeech


def test_lanes_forced_durations_graph_replay_bitwise():
    import bench
    from artspeech_amd import models
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    net = _net(dev)
    ref = _alone(net)
    n_lanes, rounds = 4, 4
    gs, wants = [], []
    for i in range(n_lanes):
        _, g = bench.make_inputs(dev, 8, 24, 60, 100, seed0=bench.DATA_SEED + 10 * i, vary=True)
        gs.append(g)
        side = net.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], forced=g["forced"], frames_hint=g["frames"])["mel"]
        assert float((side - ref.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], forced=g["forced"],
                                                frames_hint=g["frames"])["mel"]).abs().max()) <= 3e-5
        wants.append(ref.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], forced=g["forced"],
                                        frames_hint=g["frames"])["mel"].clone())
    torch.cuda.synchronize()
    lanes = models.Lanes(net, n_lanes)
    outs = [None] * n_lanes
    rounds += 1
    for r in range(rounds):                                   # rounds 0, 1 eager (eager plan, graph plan), round 2 captured + launched, then replayed
        for i, g in enumerate(gs):
            lane, outs[i] = lanes.submit(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], forced=g["forced"],
                                         frames=g["frames"], out=outs[i])
            assert lane == i
        lanes.wait()
        for i in range(n_lanes):
            assert torch.equal(outs[i]["mel"], wants[i]), (r, i)
            outs[i]["mel"].zero_()                            # the next round must write it again
    # another geometry on the same lanes (workspaces grow, graphs are dropped), then the first one again
    _, big = bench.make_inputs(dev, 12, 30, 80, 120, seed0=bench.DATA_SEED + 77, vary=True)
    want_big = ref.forward_packed(big["tok"], big["tok_lens"], big["mel"], big["f0"], big["ema"], big["ref_lens"], forced=big["forced"],
                                  frames_hint=big["frames"])["mel"].clone()
    ob = None
    for r in range(4 * n_lanes):
        _, ob = lanes.submit(big["tok"], big["tok_lens"], big["mel"], big["f0"], big["ema"], big["ref_lens"], forced=big["forced"],
                             frames=big["frames"], out=ob)
        lanes.wait()
        assert torch.equal(ob["mel"], want_big), r
    lane, o0 = lanes.submit(gs[0]["tok"], gs[0]["tok_lens"], gs[0]["mel"], gs[0]["f0"], gs[0]["ema"], gs[0]["ref_lens"],
                            forced=gs[0]["forced"], frames=gs[0]["frames"], out=outs[0])
    lanes.wait(lane)
    assert torch.equal(o0["mel"], wants[0])
    lanes.close()


def test_lanes_predicted_durations():
    import bench
    from artspeech_amd import models
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    net = _net(dev)
    _, g = bench.make_inputs(dev, 6, 20, 50, 90, seed0=bench.DATA_SEED + 5, vary=True)
    ref = _alone(net).forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"])          # predicted durations
    frames = ref["frames"]
    lanes = models.Lanes(net, 2)
    cap = 2 * sum(frames)
    for _ in range(3):
        lane, o = lanes.submit(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], capacity=cap)
        lanes.wait(lane)
        assert o["frames"] == frames
        assert torch.equal(o["mel"][:, :cap], ref["mel"][:, :cap])
    # an output buffer that cannot hold the result: AS_ENOSPC, and the frame counts say what is needed
    from artspeech_amd._lib import HipLibraryError
    with pytest.raises(HipLibraryError):
        lanes.submit(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], capacity=cap // 2)
    lanes.wait()
    lanes.close()


def _tiny_net(dev):
    from artspeech_amd import models, synth
    from artspeech_amd.weights import DEFAULT_STATS, load_distribution
    sd = synth.synth_state_dict(64, 8, seed=11)
    model = models.build_model(models.Munch(hidden_dim=64, dim_in=8, style_dim=256, n_mels=80), None, "second", load_distribution(DEFAULT_STATS), dev)
    models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
    return model.ArtsSpeech


def test_lanes_survive_layout_flushes_and_graph_eviction():
    """VERDICT r3 / ADVICE r3 (high): a lane's graphs hold geometry-table addresses.  The eager plan's layout cache is flushed again and
    again here (cap 64, > 300 ragged geometries with known frame counts through 4 lanes) while hot geometries keep being replayed from
    their graphs, and the graph cache itself is driven past its cap: every result stays bitwise equal to as_forward_test run alone."""
    import itertools
    import bench
    from artspeech_amd import _lib, models
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    net = _tiny_net(dev)
    solo = _alone(net)
    host, _ = bench.make_inputs(None, 14, 20, 44, 96, seed0=bench.DATA_SEED + 300, vary=True)
    subsets = [c for k in (2, 3) for c in itertools.combinations(range(14), k)]     # 91 + 364 distinct ragged geometries
    n_lanes = 4

    def alone(g):
        return solo.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], forced=g["forced"],
                                  frames_hint=g["frames"])["mel"].clone()

    def submit(lanes, g, out=None):
        return lanes.submit(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], forced=g["forced"], frames=g["frames"], out=out)

    lanes = models.Lanes(net, n_lanes)
    lanes.set_layout_cap(64)
    lanes.set_graph_cap(2)
    # three hot sets of one geometry per lane: with a cap of two graphs per lane the third set evicts
    hot = []
    for k in range(3 * n_lanes):
        g = bench.pack_inputs(host, list(subsets[k]), dev)
        hot.append(dict(g=g, want=alone(g), out=None))
    torch.cuda.synchronize()

    def hot_round(k):                                          # set k: lane i gets hot[k * n_lanes + i]
        for i in range(n_lanes):
            h = hot[k * n_lanes + i]
            lane, h["out"] = submit(lanes, h["g"], h["out"])
            assert lane == i
        lanes.wait()
        for i in range(n_lanes):
            h = hot[k * n_lanes + i]
            assert torch.equal(h["out"]["mel"], h["want"]), (k, i)
            h["out"]["mel"].zero_()

    # workspaces at their final size from the start (a workspace that moves drops the lane's graphs -- not what this test is after; and a
    # smaller batch can need MORE room than a larger one: split-K slabs exist only where the tile grid is small)
    lanes.reserve(256 << 20, 256 << 20)
    for _ in range(5):                                         # eager, eager (graph plan), captured, replayed, replayed
        hot_round(0)
    st = lanes.stats(0)
    assert st["graphs"] == 1 and st["captures"] == 1 and st["graph_launches"] >= 3, st
    drops0 = [lanes.stats(i)["graph_drops"] for i in range(n_lanes)]
    # > 300 new ragged geometries, each once (they never reach the graph plan), the hot set replayed in between
    n_new = 0
    for r in range(80):
        fresh = []
        for i in range(n_lanes):
            g = bench.pack_inputs(host, list(subsets[3 * n_lanes + n_new]), dev)
            n_new += 1
            lane, o = submit(lanes, g)
            fresh.append((g, o))
        lanes.wait()
        for g, o in fresh:
            assert torch.equal(o["mel"], alone(g)), r
        if r % 8 == 7:
            hot_round(0)
    assert n_new > 300
    for i in range(n_lanes):
        st = lanes.stats(i)
        assert st["layout_flushes"] >= 3, st                  # the eager plan was flushed under the graphs ...
        assert st["captures"] == 1 and st["graph_drops"] == drops0[i], st   # ... which were never rebuilt
    hot_round(0)
    # drive the graph cache past its cap: sets 1 and 2 come in, set 0 is revisited after the eviction
    for k in (1, 2, 0, 1, 2, 0):
        for _ in range(4):
            hot_round(k)
    st = lanes.stats(0)
    assert st["graph_drops"] >= drops0[0] + 2 and st["graphs"] <= 2, st
    assert _lib.lib().as_device_status(0) == 0
    lanes.close()


def test_lanes_clustered_lstm_soak():
    """four lanes' clustered H = 256 recurrences share the chip.  Batch-1 requests (C2's shape) and a longer batch, a few hundred
    steps in flight: no poll gives up (as_lanes_wait reports AS_EDEVICE if one did) and the results stay those of a step run alone."""
    import bench
    from artspeech_amd import models
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    net = _net(dev)
    solo = _alone(net)
    for n_utt, n_tok, m_half, rounds in ((1, 30, 75, 60), (8, 96, 200, 12)):
        gs, wants, outs = [], [], []
        for i in range(4):
            _, g = bench.make_inputs(dev, n_utt, n_tok, m_half, 150, seed0=bench.DATA_SEED + 900 + 10 * i, vary=n_utt > 1)
            gs.append(g)
            wants.append(solo.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], forced=g["forced"],
                                             frames_hint=g["frames"])["mel"].clone())
            outs.append(None)
        torch.cuda.synchronize()
        lanes = models.Lanes(net, 4)
        for r in range(rounds):
            for i, g in enumerate(gs):
                _, outs[i] = lanes.submit(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], forced=g["forced"],
                                          frames=g["frames"], out=outs[i])
            if r % 10 == 9 or r == rounds - 1:
                lanes.wait()                                   # raises on AS_EDEVICE (a timed-out poll)
                for i in range(4):
                    assert torch.equal(outs[i]["mel"], wants[i]), (n_utt, r, i)
        lanes.close()


def test_lanes_coalesce_adjacent_submissions():
    """as_lanes_set_coalesce(2): submissions whose tensors are adjacent column ranges of one block go out two at a time as ONE
    as_forward_test call (models.py:361-362 is per utterance: any grouping is legal).  Every submission's mel against the same 32 ragged
    utterances run alone (not bitwise: twice the columns, another GEMM tile); the replayed graph of a group gives the bits of its eager
    launches; a submission that is NOT a neighbour sends the waiting one out alone -- and then they are the bits of the alone run --;
    as_lanes_wait launches what still waits."""
    import bench
    from artspeech_amd import _lib, models
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    net = _net(dev)
    solo = _alone(net)
    per, k, n_lanes = 16, 2, 2
    blocks, wants = [], []
    for i in range(n_lanes):
        host, g = bench.make_inputs(dev, per * k, 30, 70, 120, seed0=bench.DATA_SEED + 40 * i, vary=True)
        subs, mel_all = bench.adjacent_submissions(g, per)
        blocks.append((g, subs, mel_all))
        for j in range(k):
            gj = bench.pack_inputs(host, list(range(j * per, (j + 1) * per)), dev)
            wants.append(solo.forward_packed(gj["tok"], gj["tok_lens"], gj["mel"], gj["f0"], gj["ema"], gj["ref_lens"], forced=gj["forced"],
                                             frames_hint=gj["frames"])["mel"].clone())
    torch.cuda.synchronize()
    lanes = models.Lanes(net, n_lanes)
    lanes.set_coalesce(k)

    def submit(sub):
        lane, _ = lanes.submit(sub["tok"], sub["tok_lens"], sub["mel"], sub["f0"], sub["ema"], sub["ref_lens"], forced=sub["forced"],
                               frames=sub["frames"], out=sub["out"])
        return lane
    first = None
    for r in range(5):                                         # eager, eager (graph plan), captured, replayed, replayed
        for i, (g, subs, mel_all) in enumerate(blocks):
            mel_all.zero_()
            assert [submit(sub) for sub in subs] == [i] * k    # a group fills ONE lane, then the turn passes on
        lanes.wait()
        got = [sub["out"]["mel"].clone() for (_, subs, _) in blocks for sub in subs]
        for a, w in zip(got, wants):
            assert a.shape == w.shape and float((a - w).abs().max()) <= 3e-5
        if first is None:
            first = got
        else:
            for a, b in zip(got, first):
                assert torch.equal(a, b), r
    assert all(lanes.merged_calls(i) == 5 for i in range(n_lanes))
    assert lanes.stats(0)["graph_launches"] >= 2
    # not neighbours: the first halves of the two blocks one after the other -- each goes out alone, with the bits of the alone run
    for (_, subs, mel_all) in blocks:
        mel_all.zero_()
    la = submit(blocks[0][1][0])
    lb = submit(blocks[1][1][0])                               # (sends the one that waits on lane `la` out first)
    assert lb == (la + 1) % n_lanes
    lanes.wait()                                               # launches the one still waiting
    assert torch.equal(blocks[0][1][0]["out"]["mel"], wants[0]) and torch.equal(blocks[1][1][0]["out"]["mel"], wants[k])
    assert all(lanes.merged_calls(i) == 5 for i in range(n_lanes))
    # back to one submission per call
    lanes.set_coalesce(1)
    sub = blocks[0][1][1]
    sub["out"]["mel"].zero_()
    lane, o = lanes.submit(sub["tok"], sub["tok_lens"], sub["mel"], sub["f0"], sub["ema"], sub["ref_lens"], forced=sub["forced"],
                           frames=sub["frames"], out=sub["out"])
    lanes.wait(lane)
    assert torch.equal(o["mel"], wants[1])
    assert _lib.lib().as_device_status(0) == 0
    lanes.close()


def test_c3_coalesced_lanes_vs_oracle():
    """BASELINE config C3 through the coalescing lanes: two ragged batches of 32 utterances, adjacent in one block, launched by as_lanes as
    ONE as_forward_test call over 64 utterances (from its replayed hipGraph) -- every one of the 64 utterances against the oracle's batch-1
    run of the same utterance with the same forced integer durations: 1e-4, the north-star bound."""
    import os
    import bench
    from artspeech_amd import _lib, models, synth
    from artspeech_amd.weights import DEFAULT_STATS, fold_state_dict, load_distribution
    from oracle import acoustic
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    net = _net(dev)
    host, g = bench.make_inputs(dev, 64, vary=True, seed0=bench.DATA_SEED + 555)
    subs, mel_all = bench.adjacent_submissions(g, 32)
    lanes = models.Lanes(net, 1)
    lanes.set_coalesce(2)
    for r in range(4):                                         # the last round is a graph replay
        mel_all.zero_()
        for sub in subs:
            lanes.submit(sub["tok"], sub["tok_lens"], sub["mel"], sub["f0"], sub["ema"], sub["ref_lens"], forced=sub["forced"],
                         frames=sub["frames"], out=sub["out"])
        lanes.wait()
    assert lanes.merged_calls(0) == 4 and lanes.stats(0)["graph_launches"] >= 2
    W = fold_state_dict(synth.synth_state_dict(512, 64, seed=bench.WEIGHT_SEED))
    dist = load_distribution(DEFAULT_STATS)
    got = mel_all.cpu()
    worst, o = 0.0, 0
    for b in range(64):
        ref = acoustic.forward_test(W, torch.from_numpy(host["tokens"][b]), torch.from_numpy(host["mel"][b]), torch.from_numpy(host["f0"][b]),
                                    torch.from_numpy(host["ema"][b]), dist, forced_dur=host["forced"][b])["mel"]
        n2 = 2 * host["frames"][b]
        assert ref.shape[1] == n2
        d = float((got[:, o:o + n2] - ref).abs().max())
        worst = max(worst, d)
        assert d <= 1e-4, (b, d)
        o += n2
    print("C3 coalesced 2 x 32 through as_lanes: worst mel max-abs vs oracle", worst)
    assert _lib.lib().as_device_status(0) == 0
    lanes.close()

