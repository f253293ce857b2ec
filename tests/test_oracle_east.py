"""Oracle EAST restatement vs golden vectors generated from the reference files."""
import os

import numpy as np
import torch

from oracle import east_model as oem
from oracle import east_post as P


def _biteq(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def test_decode_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "east_post.npz"))
    for q in (1, 2, 4):
        out = P.decode_quads_from_maps(g["score"], g["geo"], 0.6, 4.0, q)
        assert _biteq(out, g[f"decoded_q{q}"]), q
    assert _biteq(P.decode_quads_from_maps(g["score"], g["geo"], 0.9, 4.0, 2), g["decoded_thr09_q2"])
    e = P.decode_quads_from_maps(np.zeros_like(g["score"]), g["geo"], 0.6, 4.0, 2)
    assert e.shape == (0, 9) and e.dtype == np.float32


def test_expand_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "east_post.npz"))
    assert _biteq(P.expand_boxes(g["lanms_q2"], 0.9, 0.9), g["expanded"])
    assert _biteq(P.expand_boxes(g["lanms_q2"], 0.5, 0.0), g["expanded_w05_h0"])
    assert _biteq(P.expand_boxes(g["deg_in"], 0.9, 0.9), g["deg_expanded"])
    z = np.zeros((0, 9), np.float32)
    assert P.expand_boxes(z, 0.9, 0.9) is z


def test_decoder_head_golden(golden_dir):
    import sys
    sys.path.insert(0, golden_dir)
    from gen_golden import perturb_bn  # data-only helper (no reference import at module import time)
    g = np.load(os.path.join(golden_dir, "east_decoder_head.npz"))
    seed = int(g["seed"])
    torch.manual_seed(seed)
    dec, head = oem.FeatureMergingBranchResNet(), oem.OutputHead()
    dec.load_state_dict(perturb_bn(dec.state_dict(), seed + 1))
    dec.eval(), head.eval()
    gen = torch.Generator().manual_seed(seed + 2)
    feats = {
        "res1": torch.randn(1, 256, 16, 24, generator=gen),
        "res2": torch.randn(1, 512, 8, 12, generator=gen),
        "res3": torch.randn(1, 1024, 4, 6, generator=gen),
        "res4": torch.randn(1, 2048, 2, 3, generator=gen),
    }
    with torch.no_grad():
        h1 = dec(feats)
        score, geo = head(h1)
    np.testing.assert_allclose(h1.numpy(), g["h1"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(score.numpy(), g["score"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(geo.numpy(), g["geo"], rtol=0, atol=1e-6)


def test_resnet50_macs_and_keys():
    """Unpinned backbone: structural self-check — 4.087 GMAC conv MACs at 224^2 up to
    layer4 (published 4.09 incl. fc) and torchvision's state_dict key layout."""
    net = oem.EASTNet()
    keys = set(net.state_dict())
    for k in ("backbone.extractor.conv1.weight", "backbone.extractor.layer1.0.downsample.1.running_var",
              "backbone.extractor.layer3.5.conv3.weight", "backbone.extractor.layer4.2.bn3.bias",
              "decoder.block2.conv1x1.0.bias", "decoder.block4.conv3x3.1.running_mean",
              "output_head.score_map.bias", "output_head.geo_map.weight"):
        assert k in keys, k
    macs = [0]

    def hook(m, i, o):
        macs[0] += o.numel() * m.in_channels * m.kernel_size[0] * m.kernel_size[1] // m.groups

    hs = [m.register_forward_hook(hook) for m in net.backbone.modules() if isinstance(m, torch.nn.Conv2d)]
    with torch.no_grad():
        net.backbone(torch.zeros(1, 3, 224, 224))
    for h in hs:
        h.remove()
    assert abs(macs[0] / 1e9 - 4.087) < 0.01, macs[0]
    n_params = sum(p.numel() for p in net.backbone.parameters())
    assert n_params == 23508032  # resnet50 without fc


def test_post_filters_small():
    quads = np.array([
        [0, 0, 100, 0, 100, 40, 0, 40, 0.9],
        [10, 10, 50, 10, 50, 30, 10, 30, 0.8],   # inside the first -> removed
        [200, 0, 300, 0, 300, 40, 200, 40, 0.7],
    ], dtype=np.float32)
    out = P.remove_fully_contained_boxes(quads)
    assert out.shape[0] == 2 and out[0, 8] == np.float32(0.9) and out[1, 8] == np.float32(0.7)
    aa = P.convert_to_axis_aligned(np.array([[5, 1, 9, 2, 8, 7, 4, 6, 0.5]], dtype=np.float32))
    assert aa[0, :8].tolist() == [4, 1, 9, 1, 9, 7, 4, 7]
    assert P.point_polygon_test_sign(quads[0, :8].reshape(4, 2), (50, 20)) == 1
    assert P.point_polygon_test_sign(quads[0, :8].reshape(4, 2), (100, 20)) == 0
    assert P.point_polygon_test_sign(quads[0, :8].reshape(4, 2), (101, 20)) == -1
    many = np.tile(quads[2:3], (40, 1)).astype(np.float32)
    many[:, 0:8:2] += np.arange(40, dtype=np.float32)[:, None] * 200
    many[0, [2, 4]] += 5000  # one huge box
    many[0, [5, 7]] += 5000
    kept = P.remove_area_anomalies(many, True, 5.0, 30)
    assert kept.shape[0] == 39
