"""GPU: two batches in flight (bench.py's default mode) -- two plans on the same weights, each replaying its own hipGraph on its own
stream, must keep producing the bits of a step run alone (kernels of different batches share CUs, LDS and caches)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_batches_in_flight_are_bit_identical():
    import bench
    from artspeech_amd import models, synth
    from artspeech_amd.weights import DEFAULT_STATS, load_distribution
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    sd = synth.synth_state_dict(512, 64, seed=bench.WEIGHT_SEED)
    model = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80), None, "second", load_distribution(DEFAULT_STATS), dev)
    models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
    net = model.ArtsSpeech
    _, g = bench.make_inputs(dev)
    a = bench.Runner(net, g)
    want = a.step()["mel"].clone()
    torch.cuda.synchronize()
    run_a = a.capture()
    b = bench.Runner(net.replica(), g)
    run_b = b.capture()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    bad = 0
    for i in range(120):
        with torch.cuda.stream(sa):
            run_a()
        with torch.cuda.stream(sb):
            run_b()
        if i % 8 == 7:                                       # (checked while the lanes' last steps overlapped each other)
            torch.cuda.synchronize()
            bad += int(not torch.equal(a.out["mel"], want)) + int(not torch.equal(b.out["mel"], want))
    assert bad == 0, f"{bad} of 30 checks differed from the step run alone"


def test_four_single_stream_batches_in_flight_are_bit_identical():
    """bench.py's default arrangement since the end of round 3: four batches in flight, each ONE chain on one stream (as_plan_set_serial:
    the step's branches on the one stream, their ready conv GEMMs sharing launches: as_plan_set_merge).  Without merging a chain produces
    the bits of the step with its branches on side streams; with it (the default) a conv that shares a launch may run on another tile
    shape -- same arithmetic, another order of the partial sums: within 3e-5 of the side-stream step -- and four chains replayed side by
    side keep producing the bits of the chain run alone."""
    import bench
    from artspeech_amd import models, synth
    from artspeech_amd.weights import DEFAULT_STATS, load_distribution
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    sd = synth.synth_state_dict(512, 64, seed=bench.WEIGHT_SEED)
    model = models.build_model(models.Munch(hidden_dim=512, dim_in=64, style_dim=256, n_mels=80), None, "second", load_distribution(DEFAULT_STATS), dev)
    models.load_checkpoint(model, None, {"net": {"ArtsSpeech": sd}})
    net = model.ArtsSpeech
    lanes, wants = [], []
    for i in range(4):
        _, g = bench.make_inputs(dev, seed0=bench.DATA_SEED + 100 * i)
        side = bench.Runner(net, g).step()["mel"].clone()          # branches on side streams, alone
        plain = net.replica()
        plain.rt.set_serial(True)
        plain.rt.set_merge(False)
        assert torch.equal(bench.Runner(plain, g).step()["mel"], side), "a step as one unmerged chain differs from the step with its branches on side streams"
        twin = net.replica()
        twin.rt.set_serial(True)
        r = bench.Runner(twin, g)
        want = r.step()["mel"].clone()                             # the merged chain, alone
        assert float((want - side).abs().max()) <= 3e-5        # (each is within ~1e-5 of the fp32 oracle: test_c3_full_config_ragged_batch_vs_oracle)
        lanes.append((r, r.capture(), torch.cuda.Stream()))
        wants.append(want)
    torch.cuda.synchronize()
    bad = 0
    for i in range(160):
        r, run, st = lanes[i % 4]
        with torch.cuda.stream(st):
            run()
        if i % 16 == 15:
            torch.cuda.synchronize()
            bad += sum(int(not torch.equal(r.out["mel"], w)) for (r, _, _), w in zip(lanes, wants))
    assert bad == 0, f"{bad} of 40 checks differed from the step run alone"
