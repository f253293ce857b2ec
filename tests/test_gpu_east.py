"""EAST network on HIP vs the oracle's CPU fp32 restatement (same seeded synthetic weights)."""
import numpy as np
import pytest
import torch

from manuscript_ocr_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import east_model as oem
    sd = synth.east_state_dict(seed=20260128)
    net = oem.EASTNet()
    net.load_state_dict(sd)
    net.eval()
    return sd, net


def _page(seed, H, W):
    from manuscript_ocr_amd import synth
    return synth.synth_page(seed, H, W)[0]


@pytest.mark.parametrize("hw", [(128, 160), (256, 192)])
def test_east_forward_f32_matches_oracle(setup, hw):
    """fp32 parity mode: tolerance 1e-4 absolute on the sigmoid score map, 2.5e-6 of max|geo| on the geometry (conftest.GEO_RTOL =
    2x the measured error); summation order differs from oneDNN, BN is folded."""
    from conftest import assert_maps_close
    from manuscript_ocr_amd.detectors._east.net import EastNet
    from oracle import imgproc
    sd, ref_net = setup
    H, W = hw
    pages = np.stack([_page(11, H, W), _page(12, H, W)])
    x = torch.from_numpy(np.concatenate([imgproc.east_preprocess(p, W, H) for p in pages]))
    with torch.no_grad():
        ref = ref_net(x)
    net = EastNet(sd, torch.float32)
    score, geo = net.forward(torch.from_numpy(pages).cuda())
    torch.cuda.synchronize()
    rs, rg = ref["score"][:, 0].numpy(), ref["geometry"].permute(0, 2, 3, 1).numpy()
    assert score.shape == rs.shape and geo.shape == rg.shape
    assert rs.std() > 0.01, "degenerate score map: weights do not exercise the network"
    es, eg = assert_maps_close(score.cpu().numpy(), geo.cpu().numpy(), rs, rg, f"east {hw}")
    # identical candidate set at the default threshold
    assert np.array_equal(score.cpu().numpy() > np.float32(0.6), rs > np.float32(0.6)) or es < 1e-5


def test_east_forward_bf16_close(setup):
    """bf16 throughput mode: stated tolerance 0.05 absolute on score, 5 % of max on geometry."""
    from manuscript_ocr_amd.detectors._east.net import EastNet
    from oracle import imgproc
    sd, ref_net = setup
    H, W = 128, 160
    pages = np.stack([_page(13, H, W)])
    x = torch.from_numpy(np.concatenate([imgproc.east_preprocess(p, W, H) for p in pages]))
    with torch.no_grad():
        ref = ref_net(x)
    net = EastNet(sd, torch.bfloat16)
    score, geo = net.forward(torch.from_numpy(pages).cuda())
    rs, rg = ref["score"][:, 0].numpy(), ref["geometry"].permute(0, 2, 3, 1).numpy()
    es = np.abs(score.cpu().numpy() - rs).max()
    eg = np.abs(geo.cpu().numpy() - rg).max() / max(np.abs(rg).max(), 1.0)
    assert es < 0.05 and eg < 0.05, (es, eg)


def test_hipgraph_replay_equals_eager(setup):
    """EAST(use_graphs=True): the captured detect sequence (network + decode + LANMS), replayed on fresh page bytes, returns the
    same maps and boxes as the eager launches; two handles in flight use two graph instances."""
    from manuscript_ocr_amd.detectors import EAST
    sd, _ = setup
    H, W = 128, 160
    eager = EAST(state_dict=sd, target_size=(W, H), device="cuda", score_thresh=0.5)
    graph = EAST(state_dict=sd, target_size=(W, H), device="cuda", score_thresh=0.5, use_graphs=True)
    for rnd in range(4):  # 1st call warm-up (eager), 2nd captures + replays, later ones replay
        pages = [_page(10 * rnd + k, H, W) for k in range(2)]
        a = eager.predict_batch(pages, return_maps=True)
        b = graph.predict_batch(pages, return_maps=True)
        for ra, rb in zip(a, b):
            assert np.array_equal(ra["score_map"], rb["score_map"]) and np.array_equal(ra["geo_map"], rb["geo_map"])
            assert [w.polygon for w in ra["page"].blocks[0].words] == [w.polygon for w in rb["page"].blocks[0].words]
    pool = next(iter(graph._graphs.values()))
    assert pool["warm"] and len(pool["inst"]) == 2 and not any(i["busy"] for i in pool["inst"])
    dev = torch.from_numpy(np.stack([_page(k, H, W) for k in range(2)])).cuda()
    h1, h2 = graph.detect_start(dev), graph.detect_start(dev)   # two batches in flight -> two instances
    assert h1[-1] is not None and h2[-1] is not None and h1[-1] is not h2[-1] and len(pool["inst"]) == 2
    r1 = graph.detect_finish(h1, [_page(0, H, W)] * 2)
    r2 = graph.detect_finish(h2, [_page(0, H, W)] * 2)
    assert [w.polygon for w in r1[0]["page"].blocks[0].words] == [w.polygon for w in r2[0]["page"].blocks[0].words]
    assert not any(i["busy"] for i in pool["inst"])
