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
