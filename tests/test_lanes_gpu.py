"""GPU: as_lanes (csrc/lanes.hip) -- the throughput arrangement as a piece of the library: batches submitted round robin to N serial plans
on their own streams (eager, captured, then graph-replayed) produce the bits of the same batches run one at a time through
as_forward_test; predicted durations (frame counts read back, workspace re-sized on AS_ENOSPC) included."""
import numpy as np
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
            torch.cuda.synchronize()                          # (the memset ran on torch's stream, the lanes launch on their own: order them)
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
            torch.cuda.synchronize()                          # (the memset ran on torch's stream, the lanes launch on their own: order them)

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
            torch.cuda.synchronize()                          # (the memset ran on torch's stream, the lanes launch on their own: order them)
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
    torch.cuda.synchronize()                              # (torch's stream -> the lanes' streams)
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
    torch.cuda.synchronize()                          # (the memset ran on torch's stream, the lanes launch on their own: order them)
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
        torch.cuda.synchronize()                          # (the memset ran on torch's stream, the lanes launch on their own: order them)
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



def _sub_args(sub):
    return (sub["tok"], sub["tok_lens"], sub["mel"], sub["f0"], sub["ema"], sub["ref_lens"])


def test_lanes_debug_mode_reports_buffers_overwritten_while_pending():
    """The buffer rule of as_lanes_set_coalesce (a held-back submission's device buffers are read when its GROUP is launched) is the
    caller's to keep; as_lanes_set_debug makes a breach loud: the inputs are checksummed at submit and at the group's launch, a difference
    raises AS_STATUS_BAD_LAYOUT and the launching call fails.  A clean round under debug mode gives the results of the plain lanes."""
    import bench
    from artspeech_amd import _lib, models
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    net = _net(dev)
    solo = _alone(net)
    host, g = bench.make_inputs(dev, 16, 24, 60, 100, seed0=bench.DATA_SEED + 901, vary=True)
    subs, mel_all = bench.adjacent_submissions(g, 8)
    wants = []
    for j in range(2):
        gj = bench.pack_inputs(host, list(range(8 * j, 8 * j + 8)), dev)
        wants.append(solo.forward_packed(gj["tok"], gj["tok_lens"], gj["mel"], gj["f0"], gj["ema"], gj["ref_lens"], forced=gj["forced"],
                                         frames_hint=gj["frames"])["mel"].clone())
    torch.cuda.synchronize()
    lanes = models.Lanes(net, 1)
    lanes.set_coalesce(2)
    lanes.set_debug(True)
    # a clean round
    for sub in subs:
        lanes.submit(*_sub_args(sub), forced=sub["forced"], frames=sub["frames"], out=sub["out"])
    lanes.wait()
    for sub, w in zip(subs, wants):
        assert float((sub["out"]["mel"] - w).abs().max()) <= 3e-5
    assert _lib.lib().as_device_status(0) == 0
    # the first submission's tokens are overwritten while it waits for its neighbour: the group's launch says so
    lanes.submit(*_sub_args(subs[0]), forced=subs[0]["forced"], frames=subs[0]["frames"], out=subs[0]["out"])
    keep = subs[0]["tok"].clone()
    subs[0]["tok"].copy_(torch.flip(keep, dims=[0]))
    torch.cuda.synchronize()
    with pytest.raises(_lib.HipLibraryError, match="as_device_status|kernel reported"):
        lanes.submit(*_sub_args(subs[1]), forced=subs[1]["forced"], frames=subs[1]["frames"], out=subs[1]["out"])
        lanes.wait()
    assert _lib.lib().as_device_status(0) & (1 << 4)               # AS_STATUS_BAD_LAYOUT
    assert _lib.lib().as_device_status(1) != 0                     # (clear)
    subs[0]["tok"].copy_(keep)
    torch.cuda.synchronize()
    # ... and the same with a mel row (a strided 2-D input), caught at as_lanes_wait's flush
    lanes.submit(*_sub_args(subs[0]), forced=subs[0]["forced"], frames=subs[0]["frames"], out=subs[0]["out"])
    row = subs[0]["mel"][37].clone()
    subs[0]["mel"][37].add_(1.0)
    torch.cuda.synchronize()
    with pytest.raises(_lib.HipLibraryError):
        lanes.wait()
    assert _lib.lib().as_device_status(1) & (1 << 4)
    subs[0]["mel"][37].copy_(row)
    torch.cuda.synchronize()
    # healthy again
    for sub in subs:
        lanes.submit(*_sub_args(sub), forced=sub["forced"], frames=sub["frames"], out=sub["out"])
    lanes.wait()
    for sub, w in zip(subs, wants):
        assert float((sub["out"]["mel"] - w).abs().max()) <= 3e-5
    lanes.close()


def test_lanes_failed_group_surfaces_at_wait_and_short_last_group():
    """k = 3: two submissions and as_lanes_wait -- the short group goes out as one call of two; then a group whose tokens hold an id the
    embedding cannot look up: its submits return (nothing is enqueued until the group is full), the failure of the group's kernels is what
    as_lanes_wait returns (AS_EDEVICE, AS_STATUS_BAD_TOKEN), and after the clear the lanes serve again."""
    import bench
    from artspeech_amd import _lib, models
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    net = _net(dev)
    solo = _alone(net)
    host, g = bench.make_inputs(dev, 24, 24, 60, 100, seed0=bench.DATA_SEED + 902, vary=True)
    subs, mel_all = bench.adjacent_submissions(g, 8)
    wants = []
    for j in range(3):
        gj = bench.pack_inputs(host, list(range(8 * j, 8 * j + 8)), dev)
        wants.append(solo.forward_packed(gj["tok"], gj["tok_lens"], gj["mel"], gj["f0"], gj["ema"], gj["ref_lens"], forced=gj["forced"],
                                         frames_hint=gj["frames"])["mel"].clone())
    torch.cuda.synchronize()
    lanes = models.Lanes(net, 2)
    lanes.set_coalesce(3)
    for r in range(4):                                             # full groups of three: eager, eager, captured, replayed
        for sub in subs:
            lanes.submit(*_sub_args(sub), forced=sub["forced"], frames=sub["frames"], out=sub["out"])
    lanes.wait()
    for sub, w in zip(subs, wants):
        assert float((sub["out"]["mel"] - w).abs().max()) <= 3e-5
    merged0 = sum(lanes.merged_calls(i) for i in range(2))
    assert merged0 == 4
    mel_all.zero_()
    torch.cuda.synchronize()
    la = [lanes.submit(*_sub_args(sub), forced=sub["forced"], frames=sub["frames"], out=sub["out"])[0] for sub in subs[:2]]
    assert la[0] == la[1]
    assert float(mel_all.abs().max()) == 0.0                       # (held back: nothing has run)
    lanes.wait()                                                   # the short group: ONE call over the two
    assert sum(lanes.merged_calls(i) for i in range(2)) == merged0 + 1
    for sub, w in zip(subs[:2], wants[:2]):
        assert float((sub["out"]["mel"] - w).abs().max()) <= 3e-5
    assert float(subs[2]["out"]["mel"].abs().max()) == 0.0
    # a bad token inside the second submission of a group
    keep = subs[1]["tok"].clone()
    subs[1]["tok"][3] = 100000
    torch.cuda.synchronize()
    for sub in subs[:2]:
        lanes.submit(*_sub_args(sub), forced=sub["forced"], frames=sub["frames"], out=sub["out"])   # (returns: the group is not out yet)
    with pytest.raises(_lib.HipLibraryError):
        lanes.submit(*_sub_args(subs[2]), forced=subs[2]["forced"], frames=subs[2]["frames"], out=subs[2]["out"])   # launches the group
        lanes.wait()                                               # ... whose kernels raise the bit
    assert _lib.lib().as_device_status(1) & (1 << 2)               # AS_STATUS_BAD_TOKEN; cleared
    subs[1]["tok"].copy_(keep)
    torch.cuda.synchronize()
    for sub in subs:
        lanes.submit(*_sub_args(sub), forced=sub["forced"], frames=sub["frames"], out=sub["out"])
    lanes.wait()
    for sub, w in zip(subs, wants):
        assert float((sub["out"]["mel"] - w).abs().max()) <= 3e-5
    lanes.close()


def test_lanes_submit_host_blocks_refilled_behind_groups_in_flight():
    """as_lanes_submit_host: host arrays in, host arrays out, the lane owns the device block.  Twelve DIFFERENT batches go through two
    coalescing lanes back to back without a wait in between -- every lane's block is refilled (by the submit's copies) while the block's
    previous group and the other lane's group are in flight; stream order on the lane's stream is all that keeps them apart -- and every
    utterance of every batch is held against the same batch run alone from device buffers (<= 3e-5: a group is another GEMM tile).  Then
    the same batches again (graphs by now): the bits of the first pass."""
    import bench
    from artspeech_amd import _lib, models
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    net = _net(dev)
    solo = _alone(net)
    n_b, per = 16, 8
    pin = lambda a: torch.from_numpy(a).pin_memory()
    batches, wants = [], []
    for i in range(n_b):
        # (three geometries in rotation: groups of equal geometry come back and are replayed from graphs; the lengths inside a batch vary)
        host, g = bench.make_inputs(dev, per, 24, 60, 100, seed0=bench.DATA_SEED + 3000 + (i % 3), vary=True)
        host2, _ = bench.make_inputs(None, per, 24, 60, 100, seed0=bench.DATA_SEED + 3000 + (i % 3), vary=True)
        # same geometry, other values: the reference features and tokens of batch i
        rng = np.random.default_rng(7000 + i)
        for b in range(per):
            host2["mel"][b] = (host["mel"][b] + 0.1 * rng.standard_normal(host["mel"][b].shape)).astype(np.float32)
            host2["tokens"][b] = np.concatenate([[0], rng.integers(1, 178, size=len(host["tokens"][b]) - 2), [0]]).astype(host["tokens"][b].dtype)
        gi = bench.pack_inputs(host2, list(range(per)), dev)
        wants.append(solo.forward_packed(gi["tok"], gi["tok_lens"], gi["mel"], gi["f0"], gi["ema"], gi["ref_lens"], forced=gi["forced"],
                                         frames_hint=gi["frames"])["mel"].cpu())
        cat = lambda key, ax: np.ascontiguousarray(np.concatenate([host2[key][b] for b in range(per)], ax))
        n2 = 2 * sum(host2["frames"])
        batches.append(dict(tok=pin(cat("tokens", 0).astype(np.int32)), mel=pin(cat("mel", 1)), f0=pin(cat("f0", 1).reshape(-1)),
                            ema=pin(cat("ema", 1)), forced=pin(cat("forced", 0).astype(np.int32)), tok_lens=host2["tok_lens"],
                            ref_lens=host2["ref_lens"], frames=host2["frames"], out=torch.zeros((80, n2), dtype=torch.float32).pin_memory()))
    torch.cuda.synchronize()
    lanes = models.Lanes(net, 2)
    lanes.set_coalesce(2)
    first = None
    for r in range(4):
        for b in batches:
            b["out"].zero_()
        for b in batches:                                          # no wait in between: blocks are refilled behind groups in flight
            lanes.submit_host(b["tok"], b["tok_lens"], b["mel"], b["f0"], b["ema"], b["ref_lens"], b["forced"], b["frames"], b["out"])
        lanes.wait()
        for i, (b, w) in enumerate(zip(batches, wants)):
            assert b["out"].shape == w.shape
            d = float((b["out"] - w).abs().max())
            assert d <= 3e-5, (r, i, d)
        if r == 2:
            first = [b["out"].clone() for b in batches]
        if r == 3:
            for b, f in zip(batches, first):
                assert torch.equal(b["out"], f)
    assert sum(lanes.merged_calls(i) for i in range(2)) == 4 * n_b // 2
    assert lanes.stats(0)["graph_launches"] >= 2
    # a lone host submission with coalescing off: a group of one
    lanes.set_coalesce(1)
    b = batches[5]
    b["out"].zero_()
    lane = lanes.submit_host(b["tok"], b["tok_lens"], b["mel"], b["f0"], b["ema"], b["ref_lens"], b["forced"], b["frames"], b["out"])
    lanes.wait(lane)
    assert float((b["out"] - wants[5]).abs().max()) <= 3e-5
    assert _lib.lib().as_device_status(0) == 0
    # predicted durations at the host boundary (test.py:96-113 as it is called): host arrays in, a frame capacity, the mel and the frame
    # offsets back -- coalesced two at a time; against the read-back path of every batch run alone from device buffers
    lanes.set_coalesce(2)
    caps, offs, outs2, want2 = [], [], [], []
    for i, b in enumerate(batches[:8]):
        dv = lambda t: t.to(dev)
        ref = solo.forward_packed(dv(b["tok"]), b["tok_lens"], dv(b["mel"]), dv(b["f0"]).reshape(1, -1), dv(b["ema"]), b["ref_lens"])
        off = ref["frame_off"].cpu()
        cap = int(1.2 * int(off[-1])) + 3
        caps.append(cap)
        want2.append((off, ref["mel"].cpu()))
        offs.append(torch.zeros(per + 1, dtype=torch.int32).pin_memory())
        outs2.append(torch.zeros((80, 2 * cap), dtype=torch.float32).pin_memory())
    for r in range(4):
        for o in outs2:
            o.zero_()
        for i, b in enumerate(batches[:8]):
            lanes.submit_host(b["tok"], b["tok_lens"], b["mel"], b["f0"], b["ema"], b["ref_lens"], None, None, outs2[i], frame_cap=caps[i],
                              frame_off=offs[i])
        lanes.wait()
        for i in range(8):
            off, mel = want2[i]
            assert torch.equal(offs[i], off), (r, i)
            n2 = 2 * int(off[-1])
            assert float((outs2[i][:, :n2] - mel).abs().max()) <= 3e-5, (r, i)
    assert _lib.lib().as_device_status(0) == 0
    lanes.close()


def test_lanes_predicted_durations_under_a_frame_capacity_graphs_and_coalescing():
    """as_forward_io.frame_cap through as_lanes: submissions with PREDICTED durations are replayed from hipGraphs (nothing in the call
    waits for the host) and coalesced like submissions with known counts -- two adjacent submissions of 16 ragged utterances go out as
    ONE call whose mel is dealt out to the submissions' own output buffers (as_segments), each with its own frame_off from 0.  Every
    utterance against the read-back path of its own submission run alone (<= 3e-5: another GEMM tile), frame offsets equal, replays bit
    for bit; then uncoalesced (k = 1) submissions of the same kind: eager, captured, replayed."""
    import bench
    from artspeech_amd import _lib, models
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    net = _net(dev)
    solo = _alone(net)
    per, k, n_lanes = 16, 2, 2
    blocks = []
    for i in range(n_lanes):
        host, g = bench.make_inputs(dev, per * k, 30, 70, 120, seed0=bench.DATA_SEED + 60 * i, vary=True)
        subs, _ = bench.adjacent_submissions(g, per)
        for j, sub in enumerate(subs):
            gj = bench.pack_inputs(host, list(range(j * per, (j + 1) * per)), dev)
            ref = solo.forward_packed(gj["tok"], gj["tok_lens"], gj["mel"], gj["f0"], gj["ema"], gj["ref_lens"])       # predicted, read-back path
            sub["want_off"] = ref["frame_off"].cpu().numpy().copy()
            sub["want_mel"] = ref["mel"].clone()
            sub["cap"] = int(1.25 * int(sub["want_off"][-1])) + 5
            sub["res"] = None
        blocks.append(subs)
    torch.cuda.synchronize()
    lanes = models.Lanes(net, n_lanes)
    lanes.set_coalesce(k)

    def run_round():
        for i, subs in enumerate(blocks):
            got = []
            for sub in subs:
                lane, sub["res"] = lanes.submit(sub["tok"], sub["tok_lens"], sub["mel"], sub["f0"], sub["ema"], sub["ref_lens"],
                                                frame_cap=sub["cap"], out=sub["res"])
                got.append(lane)
            assert got == [i] * k
        lanes.wait()
    first = None
    for r in range(5):                                             # eager, eager (graph plan), captured, replayed, replayed
        for subs in blocks:
            for sub in subs:
                if sub["res"] is not None:
                    sub["res"]["mel"].zero_()
                    sub["res"]["frame_off"].zero_()
        torch.cuda.synchronize()
        run_round()
        assert _lib.lib().as_device_status(0) == 0, _lib.device_status()
        got = []
        for subs in blocks:
            for sub in subs:
                off = sub["res"]["frame_off"].cpu().numpy()
                assert np.array_equal(off, sub["want_off"]), (r, off, sub["want_off"])
                n2 = 2 * int(off[-1])
                assert sub["res"]["mel"].shape[1] == 2 * sub["cap"]
                d = float((sub["res"]["mel"][:, :n2] - sub["want_mel"]).abs().max())
                assert d <= 3e-5, (r, d)
                got.append(sub["res"]["mel"][:, :n2].clone())
        if r == 3:
            first = got
        if r == 4:
            for a, b in zip(got, first):
                assert torch.equal(a, b)
    assert all(lanes.merged_calls(i) == 5 for i in range(n_lanes))
    assert lanes.stats(0)["graph_launches"] >= 2 and lanes.stats(0)["eager_calls"] == 2
    # one submission per call
    lanes.set_coalesce(1)
    sub = blocks[0][1]
    res = None
    for r in range(4):
        lane, res = lanes.submit(sub["tok"], sub["tok_lens"], sub["mel"], sub["f0"], sub["ema"], sub["ref_lens"], frame_cap=sub["cap"], out=res)
        lanes.wait(lane)
        n2 = 2 * int(sub["want_off"][-1])
        assert np.array_equal(res["frame_off"].cpu().numpy(), sub["want_off"])
        assert np.array_equal(res["dur_i"].cpu().numpy().shape, (sum(sub["tok_lens"]),))
        assert float((res["mel"][:, :n2] - sub["want_mel"]).abs().max()) <= 3e-5
    assert _lib.lib().as_device_status(0) == 0
    lanes.close()


def test_lanes_frame_capacity_three_unequal_submissions_and_one_too_small():
    """Coalescing under a frame capacity with k = 3 and submissions of DIFFERENT sizes (8, 5 and 11 utterances: only the inputs have to be
    adjacent, every submission has its own output buffer and frame offsets); a short last group sent out by as_lanes_wait; and one
    submission whose capacity is too small for what the predictor says: AS_STATUS_CAPACITY (its slot is never written past), and after the
    clear the lanes serve again."""
    import bench
    from artspeech_amd import _lib, models
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    net = _net(dev)
    solo = _alone(net)
    sizes = [8, 5, 11]
    host, g = bench.make_inputs(dev, sum(sizes), 28, 66, 110, seed0=bench.DATA_SEED + 4242, vary=True)
    subs, t0, r0, u0 = [], 0, 0, 0
    for n in sizes:
        nt, nr = sum(g["tok_lens"][u0:u0 + n]), sum(g["ref_lens"][u0:u0 + n])
        sub = dict(tok=g["tok"][t0:t0 + nt], mel=g["mel"][:, r0:r0 + nr], f0=g["f0"][:, r0:r0 + nr], ema=g["ema"][:, r0:r0 + nr],
                   tok_lens=g["tok_lens"][u0:u0 + n], ref_lens=g["ref_lens"][u0:u0 + n], res=None)
        gj = bench.pack_inputs(host, list(range(u0, u0 + n)), dev)
        ref = solo.forward_packed(gj["tok"], gj["tok_lens"], gj["mel"], gj["f0"], gj["ema"], gj["ref_lens"])
        sub["want_off"], sub["want_mel"] = ref["frame_off"].cpu(), ref["mel"].clone()
        sub["cap"] = int(1.2 * int(sub["want_off"][-1])) + 3
        subs.append(sub)
        t0, r0, u0 = t0 + nt, r0 + nr, u0 + n
    torch.cuda.synchronize()
    lanes = models.Lanes(net, 1)
    lanes.set_coalesce(3)

    def check(which):
        for sub in which:
            off = sub["res"]["frame_off"].cpu()
            assert torch.equal(off, sub["want_off"])
            n2 = 2 * int(off[-1])
            assert float((sub["res"]["mel"][:, :n2] - sub["want_mel"]).abs().max()) <= 3e-5
    for r in range(4):                                             # eager, eager, captured, replayed
        for sub in subs:
            _, sub["res"] = lanes.submit(sub["tok"], sub["tok_lens"], sub["mel"], sub["f0"], sub["ema"], sub["ref_lens"], frame_cap=sub["cap"], out=sub["res"])
        lanes.wait()
        check(subs)
    assert lanes.merged_calls(0) == 4 and lanes.stats(0)["graph_launches"] >= 2
    # a short group: two of the three, sent out by the wait
    for sub in subs[:2]:
        sub["res"]["mel"].zero_()
    torch.cuda.synchronize()
    for sub in subs[:2]:
        lanes.submit(sub["tok"], sub["tok_lens"], sub["mel"], sub["f0"], sub["ema"], sub["ref_lens"], frame_cap=sub["cap"], out=sub["res"])
    lanes.wait()
    check(subs[:2])
    assert lanes.merged_calls(0) == 5
    # the middle submission without enough room: loud, and its neighbours' buffers end where they end
    small = int(subs[1]["want_off"][-1]) - 20
    guard = torch.full((80, 2 * small + 32), 7.0, device=dev)
    res_small = {"mel": guard[:, : 2 * small]}
    outs = [subs[0]["res"], res_small, subs[2]["res"]]
    caps = [subs[0]["cap"], small, subs[2]["cap"]]
    try:
        for sub, o, c in zip(subs, outs, caps):
            lanes.submit(sub["tok"], sub["tok_lens"], sub["mel"], sub["f0"], sub["ema"], sub["ref_lens"], frame_cap=c, out=o)
        with pytest.raises(_lib.HipLibraryError):
            lanes.wait()
        assert _lib.lib().as_device_status(0) & (1 << 5)               # AS_STATUS_CAPACITY
        assert float((guard[:, 2 * small:] - 7.0).abs().max()) == 0.0  # nothing past the slot
    finally:
        _lib.lib().as_device_status(1)
    for sub in subs:
        _, sub["res"] = lanes.submit(sub["tok"], sub["tok_lens"], sub["mel"], sub["f0"], sub["ema"], sub["ref_lens"], frame_cap=sub["cap"], out=sub["res"])
    lanes.wait()
    check(subs)
    lanes.close()
