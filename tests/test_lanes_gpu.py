"""GPU: as_lanes (csrc/lanes.hip) -- the throughput arrangement as a piece of the library: batches submitted round robin to N serial plans
on their own streams (eager, captured, then graph-replayed) produce the bits of the same batches run one at a time through
as_forward_test; predicted durations (frame counts read back, workspace re-sized on AS_ENOSPC) included."""
import pytest
import torch

pytestmark = pytest.mark.gpu


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
    n_lanes, rounds = 4, 4
    gs, wants = [], []
    for i in range(n_lanes):
        _, g = bench.make_inputs(dev, 8, 24, 60, 100, seed0=bench.DATA_SEED + 10 * i, vary=True)
        gs.append(g)
        wants.append(net.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"], forced=g["forced"],
                                        frames_hint=g["frames"])["mel"].clone())
    torch.cuda.synchronize()
    lanes = models.Lanes(net, n_lanes)
    outs = [None] * n_lanes
    for r in range(rounds):                                   # round 0 eager, round 1 captured + launched, rounds 2.. replayed
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
    want_big = net.forward_packed(big["tok"], big["tok_lens"], big["mel"], big["f0"], big["ema"], big["ref_lens"], forced=big["forced"],
                                  frames_hint=big["frames"])["mel"].clone()
    ob = None
    for r in range(3 * n_lanes):
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
    ref = net.forward_packed(g["tok"], g["tok_lens"], g["mel"], g["f0"], g["ema"], g["ref_lens"])          # predicted durations
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
