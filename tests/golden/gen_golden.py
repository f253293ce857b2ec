"""Generate golden vectors by running the REFERENCE's own source files.

Run once in the build container (needs /root/reference):
    python tests/golden/gen_golden.py
Outputs small fixtures next to this file.  Only DATA is stored (inputs, seeds,
expected outputs) — never reference source.  See _refload.py for how single
reference files are loaded behind inert stubs.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _refload  # noqa: E402
from manuscript_ocr_amd import synth  # noqa: E402
from oracle import east_model as oem  # noqa: E402
from oracle import trba_model as otm  # noqa: E402


def perturb_bn(sd, seed):
    """Deterministic non-trivial BN statistics/affine (applied to reference AND oracle)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        v = v.clone()
        if k.endswith("running_var"):
            v = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith("running_mean"):
            v = 0.2 * torch.randn(v.shape, generator=g)
        out[k] = v
    # BN affine: identify by the sibling running_mean key
    for k in list(out):
        if k.endswith("running_mean"):
            base = k[: -len("running_mean")]
            out[base + "weight"] = 0.7 + 0.6 * torch.rand(out[base + "weight"].shape, generator=g)
            out[base + "bias"] = 0.2 * torch.randn(out[base + "bias"].shape, generator=g)
    return out


def gen_lanms():
    ref = _refload.ref_lanms()
    out = {}
    rng = np.random.default_rng(11)
    cases = {}
    # (a) candidates decoded from injected maps (realistic: many near-duplicates per word)
    utils = _refload.ref_east_utils()
    for name, (H, W, seed) in {"page_small": (256, 384, 3), "page_mid": (512, 768, 4)}.items():
        rects = synth.synth_layout(seed, H, W)
        score, geo = synth.synth_maps(rects, (H, W), (H // 4, W // 4), seed)
        cand = utils.decode_quads_from_maps(score, geo, 0.6, 4.0, 2)
        cases[name] = cand
        out[f"{name}_score"], out[f"{name}_geo"] = score, geo
        out[f"{name}_decoded"] = cand
    # (b) random rotated quads, tie-free
    def rot_quads(n, span):
        c = rng.uniform(20, span, size=(n, 2))
        wh = rng.uniform(8, 60, size=(n, 2))
        th = rng.uniform(-0.6, 0.6, size=n)
        base = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=np.float64) * 0.5
        q = np.empty((n, 9), dtype=np.float32)
        for i in range(n):
            R = np.array([[np.cos(th[i]), -np.sin(th[i])], [np.sin(th[i]), np.cos(th[i])]])
            pts = (base * wh[i]) @ R.T + c[i]
            k = rng.integers(0, 4)  # random vertex rotation; keep winding
            pts = np.roll(pts, k, axis=0)
            q[i, :8] = pts.reshape(-1)
            q[i, 8] = rng.uniform(0.6, 1.0)
        assert len(np.unique(q[:, 0])) == n and len(np.unique(q[:, 8])) == n
        return q
    cases["rot_1"] = rot_quads(1, 100)
    cases["rot_50"] = rot_quads(50, 200)
    cases["rot_600"] = rot_quads(600, 700)
    # (c) reversed winding (clockwise in image coords flipped) -> exercises empty intersections
    rv = rot_quads(40, 150)
    rv[:, :8] = rv[:, :8].reshape(-1, 4, 2)[:, ::-1].reshape(-1, 8)
    cases["rot_rev_40"] = rv
    for name, boxes in cases.items():
        for thr in (0.2,):
            res = ref.locality_aware_nms(boxes, thr)
            out[f"{name}_in"] = boxes.astype(np.float32)
            out[f"{name}_out"] = res.astype(np.float32)
            print("lanms", name, boxes.shape, "->", res.shape)
    np.savez_compressed(os.path.join(HERE, "lanms.npz"), **out)


def gen_east_post():
    utils = _refload.ref_east_utils()
    out = {}
    H, W, seed = 192, 256, 21
    rects = synth.synth_layout(seed, H * 4, W * 4)
    score, geo = synth.synth_maps(rects, (H * 4, W * 4), (H, W), seed)
    # sprinkle isolated above-threshold pixels incl. last row/col cells and sub-threshold cell centres
    rng = np.random.default_rng(5)
    ys, xs = rng.integers(0, H, 300), rng.integers(0, W, 300)
    score[ys, xs] = rng.uniform(0.55, 0.99, 300).astype(np.float32)
    out["score"], out["geo"] = score, geo
    for q in (1, 2, 4):
        out[f"decoded_q{q}"] = utils.decode_quads_from_maps(score, geo, 0.6, 4.0, q)
    out["decoded_thr09_q2"] = utils.decode_quads_from_maps(score, geo, 0.9, 4.0, 2)
    out["decoded_empty"] = utils.decode_quads_from_maps(np.zeros_like(score), geo, 0.6, 4.0, 2)
    lan = _refload.ref_lanms().locality_aware_nms(out["decoded_q2"], 0.2)
    out["lanms_q2"] = lan
    out["expanded"] = utils.expand_boxes(lan, 0.9, 0.9)
    out["expanded_w05_h0"] = utils.expand_boxes(lan, 0.5, 0.0)
    # degenerate quads for expand: zero-area, repeated vertices, reversed winding
    deg = np.array([
        [0, 0, 10, 0, 10, 0, 0, 0, 0.7],
        [5, 5, 5, 5, 5, 5, 5, 5, 0.8],
        [0, 0, 0, 10, 10, 10, 10, 0, 0.9],
    ], dtype=np.float32)
    out["deg_in"] = deg
    out["deg_expanded"] = utils.expand_boxes(deg, 0.9, 0.9)
    np.savez_compressed(os.path.join(HERE, "east_post.npz"), **out)
    print("east_post", {k: v.shape for k, v in out.items()})


def gen_east_decoder_head():
    rm = _refload.ref_east_model()
    seed = 1234
    torch.manual_seed(seed)
    ref_dec, ref_head = rm.FeatureMergingBranchResNet(), rm.OutputHead()
    torch.manual_seed(seed)
    my_dec, my_head = oem.FeatureMergingBranchResNet(), oem.OutputHead()
    for a, b in ((ref_dec, my_dec), (ref_head, my_head)):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb)
        assert all(torch.equal(sa[k], sb[k]) for k in sa), "seeded construction differs from reference"
    sd = perturb_bn(ref_dec.state_dict(), seed + 1)
    ref_dec.load_state_dict(sd)
    ref_dec.eval(), ref_head.eval()
    g = torch.Generator().manual_seed(seed + 2)
    feats = {
        "res1": torch.randn(1, 256, 16, 24, generator=g),
        "res2": torch.randn(1, 512, 8, 12, generator=g),
        "res3": torch.randn(1, 1024, 4, 6, generator=g),
        "res4": torch.randn(1, 2048, 2, 3, generator=g),
    }
    with torch.no_grad():
        h1 = ref_dec(feats)
        score, geo = ref_head(h1)
    np.savez_compressed(os.path.join(HERE, "east_decoder_head.npz"), seed=seed,
                        h1=h1.numpy(), score=score.numpy(), geo=geo.numpy())
    print("east_decoder_head", h1.shape, float(h1.abs().mean()), float(score.mean()))


def gen_trba():
    rt = _refload.ref_trba_model()
    out = {}
    seed = 4321
    for tag, (B, h, w, max_len) in {"b4_32x100": (4, 32, 100, 25), "b2_64x256": (2, 64, 256, 25)}.items():
        torch.manual_seed(seed)
        ref = rt.TRBAModel(num_classes=194, blank_id=None)
        torch.manual_seed(seed)
        mine = otm.TRBANet(194, 256)
        sa, sb = ref.state_dict(), mine.state_dict()
        assert list(sa) == list(sb), [k for k in sa if k not in sb]
        assert all(torch.equal(sa[k], sb[k]) for k in sa), "seeded construction differs from reference"
        ref.load_state_dict(synth.trba_state_dict(194, 256, seed=seed), strict=True)
        ref.eval()
        crops = synth.synth_crops(seed + 2, B, h, w)
        x = torch.from_numpy(((crops.astype(np.float32) - 127.5) * np.float32(1 / 127.5)).transpose(0, 3, 1, 2).copy())
        with torch.no_grad():
            f = ref.cnn(x)
            enc = ref.encode(x)
            gl, gi = ref(x, is_train=False, batch_max_length=max_len, mode="greedy")
            bl, bi = ref(x, is_train=False, batch_max_length=max_len, mode="beam", beam_size=8, alpha=0.9, temperature=1.7)
            bl5, bi5 = ref(x, is_train=False, batch_max_length=max_len, mode="beam", beam_size=5, alpha=0.0, temperature=1.0)
        out[f"{tag}_cnn"] = f.numpy()
        out[f"{tag}_enc"] = enc.numpy()
        out[f"{tag}_greedy_logits"], out[f"{tag}_greedy_ids"] = gl.numpy(), gi.numpy()
        out[f"{tag}_beam_logits"], out[f"{tag}_beam_ids"] = bl.numpy(), bi.numpy()
        out[f"{tag}_beam5_logits"], out[f"{tag}_beam5_ids"] = bl5.numpy(), bi5.numpy()
        print("trba", tag, f.shape, enc.shape, gl.shape, bl.shape, "enc|mean|", float(enc.abs().mean()),
              "greedy ids", gi[0, :8].tolist(), "beam ids", bi[0, :8].tolist())
    out["seed"] = seed
    np.savez_compressed(os.path.join(HERE, "trba.npz"), **out)


def gen_pipeline_glue():
    utils = _refload.ref_east_utils()
    rng = np.random.default_rng(77)
    cases = []
    fixed = [
        [(10, 10, 50, 30), (60, 10, 100, 30), (10, 50, 50, 70)],
        [(10, 10, 55, 30), (50, 10, 100, 30)],
        [(0, 0, 100, 100), (10, 10, 90, 90), (20, 20, 80, 80)],
        [],
        [(5, 5, 6, 6)],
        [(10, 10, 60, 40), (10, 10, 60, 40), (70, 12, 120, 38)],
    ]
    for boxes in fixed:
        cases.append(boxes)
    for n in (8, 40, 150):
        rects = synth.synth_layout(int(rng.integers(1 << 30)), 600, 900)
        idx = rng.permutation(len(rects))[:n]
        b = []
        for r in rects[idx]:
            j = rng.integers(-6, 7, size=4)
            b.append(tuple(int(v) for v in (r + j)))
        cases.append(b)
    res = []
    for boxes in cases:
        boxes_np = [tuple(np.int32(v) for v in b) for b in boxes]  # the pipeline passes np.int32 tuples
        res.append({
            "boxes": [list(map(int, b)) for b in boxes],
            "resolved": [list(map(int, b)) for b in utils.resolve_intersections(boxes_np)],
            "sorted": [list(map(int, b)) for b in utils.sort_boxes_reading_order(boxes_np)],
            "sorted_res": [list(map(int, b)) for b in utils.sort_boxes_reading_order_with_resolutions(boxes_np)],
        })
    with open(os.path.join(HERE, "pipeline_glue.json"), "w") as f:
        json.dump(res, f)
    print("pipeline_glue", len(res), "cases")


if __name__ == "__main__":
    which = sys.argv[1:] or ["lanms", "east_post", "east_decoder_head", "trba", "pipeline_glue"]
    for w in which:
        globals()["gen_" + w]()
