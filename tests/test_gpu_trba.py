"""TRBA on HIP vs the oracle (CPU fp32 restatement, pinned bit-exactly to the reference by
tests/golden/trba.npz) and vs the golden vectors themselves.  Floating point (fp32 mode; different
summation order, hoisted i2h, BN folded): CNN/encoder features within 2e-4 absolute, decoder logits
within 1e-3 of the largest |logit| (the synthetic weights amplify recurrent
state: rounding noise grows ~100x over the 25-26 dependent steps); ids / texts identical."""
import os

import numpy as np
import pytest
import torch

from manuscript_ocr_amd import synth

pytestmark = pytest.mark.gpu

CHARSET = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "manuscript_ocr_amd", "recognizers", "_trba",
                       "configs", "charset.txt")


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import trba_model as otm
    return otm


def _oracle_net(otm, seed):
    net = otm.TRBANet(194, 256)
    net.load_state_dict(synth.trba_state_dict(194, 256, seed=seed), strict=True)
    return net.eval()


def _x_from_canvases(c):
    return torch.from_numpy(((c.astype(np.float32) - 127.5) * np.float32(1 / 127.5)).transpose(0, 3, 1, 2).copy())


def test_se_residual_and_bilstm_ops(env):
    from manuscript_ocr_amd import ops
    g = torch.Generator().manual_seed(1)
    x, idt = torch.randn(3, 8, 25, 256, generator=g), torch.randn(3, 8, 25, 256, generator=g)
    w1, w2 = torch.randn(16, 256, generator=g) * 0.1, torch.randn(256, 16, generator=g) * 0.3
    gate = torch.sigmoid(torch.relu(x.mean((1, 2)) @ w1.t()) @ w2.t())
    ref = torch.relu(x * gate[:, None, None, :] + idt)
    out = ops.se_residual(x.cuda(), idt.cuda(), w1.cuda(), w2.cuda()).cpu()
    assert (out - ref).abs().max().item() < 1e-5
    outb = ops.se_residual(x.bfloat16().cuda(), idt.bfloat16().cuda(), w1.cuda(), w2.cuda()).float().cpu()
    refb = torch.relu(x.bfloat16().float() * torch.sigmoid(torch.relu(x.bfloat16().float().mean((1, 2)) @ w1.t()) @ w2.t())[:, None, None, :]
                      + idt.bfloat16().float())
    assert (outb - refb).abs().max().item() < 3e-2
    f = torch.randn(2, 3, 7, 64, generator=g)
    assert (ops.mean_over_h(f.cuda()).cpu() - f.mean(1)).abs().max().item() < 1e-6
    # BiLSTM recurrence vs nn.LSTM
    B, T, H, In = 11, 13, 256, 64
    lstm = torch.nn.LSTM(In, H, bidirectional=True, batch_first=True)
    xs = torch.randn(B, T, In, generator=g)
    with torch.no_grad():
        ref_h, _ = lstm(xs)
        sd = lstm.state_dict()
        w_ih = torch.cat([sd["weight_ih_l0"], sd["weight_ih_l0_reverse"]])
        bias = torch.cat([sd["bias_ih_l0"] + sd["bias_hh_l0"], sd["bias_ih_l0_reverse"] + sd["bias_hh_l0_reverse"]])
        xproj = (xs.reshape(B * T, In) @ w_ih.t() + bias).contiguous()
        il = lambda w: w.t().reshape(H, 4, H).permute(0, 2, 1).contiguous()  # [k][j][gate]
        whh_t = torch.stack([il(sd["weight_hh_l0"]), il(sd["weight_hh_l0_reverse"])])
    got = ops.bilstm_recurrent(xproj.cuda(), whh_t.cuda(), B, T, H).cpu()
    assert (got - ref_h).abs().max().item() < 2e-5
    # the matrix-core recurrence (csrc/bilstm_mfma.hip: 32 crops per workgroup, h W_hh^T in the split-operand form), same bound;
    # 70 crops = two full row blocks + a ragged one
    from manuscript_ocr_amd import _native as nat
    B2 = 70
    xs2 = torch.randn(B2, T, In, generator=g) * 2.0
    with torch.no_grad():
        ref2, _ = lstm(xs2)
        xproj2 = (xs2.reshape(B2 * T, In) @ w_ih.t() + bias).contiguous()
    n = nat.lib().msocr_attn_pack_split_elems(4 * H)
    packed = torch.empty((2, n), dtype=torch.int16)
    for d in (0, 1):
        assert nat.lib().msocr_attn_pack_split_host(whh_t[d].contiguous().data_ptr(), 4 * H, 1, packed[d].data_ptr()) == 0
    got2 = ops.bilstm_recurrent(xproj2.cuda(), whh_t.cuda(), B2, T, H, packed.cuda()).cpu()
    valu = ops.bilstm_recurrent(xproj2.cuda(), whh_t.cuda(), B2, T, H).cpu()
    assert (valu - ref2).abs().max().item() < 2e-5
    assert (got2 - ref2).abs().max().item() < 2e-5, (got2 - ref2).abs().max().item()


@pytest.mark.parametrize("tag,B,h,w", [("b4_32x100", 4, 32, 100), ("b2_64x256", 2, 64, 256)])
def test_trba_vs_reference_goldens(env, golden_dir, tag, B, h, w):
    """Same seeded weights and inputs as the fixtures produced by the reference's own model files."""
    from manuscript_ocr_amd import synth
    from manuscript_ocr_amd.recognizers._trba.net import TrbaNet
    otm = env
    g = np.load(os.path.join(golden_dir, "trba.npz"))
    seed = int(g["seed"])
    sd = synth.trba_state_dict(194, 256, seed=seed)
    net = TrbaNet(sd, 194, 256, torch.float32)
    canv = synth.synth_crops(seed + 2, B, h, w)
    cd = torch.from_numpy(canv).cuda()
    f = net.cnn(cd)
    ref_f = g[f"{tag}_cnn"].transpose(0, 2, 3, 1)
    assert np.abs(f.cpu().numpy() - ref_f).max() < 2e-4 * max(1.0, np.abs(ref_f).max())
    batch_H, proj_H = net.encode(cd)
    ref_e = g[f"{tag}_enc"]  # two BiLSTM layers with x6 recurrent weights: tolerance relative to the largest activation
    assert np.abs(batch_H.cpu().numpy() - ref_e).max() < 2e-4 * max(1.0, np.abs(ref_e).max())
    # greedy: reference ran until its batch-level early break
    gl, gi = net.greedy(batch_H, proj_H, 25, 1, 2, None)
    ref_i, ref_l = g[f"{tag}_greedy_ids"], g[f"{tag}_greedy_logits"]
    tr = ref_i.shape[1]
    assert np.array_equal(gi.cpu().numpy()[:, :tr], ref_i)
    assert np.abs(gl.cpu().numpy()[:, :tr] - ref_l).max() < 1e-3 * max(1.0, np.abs(ref_l).max())
    # beam (8, T=1.7, alpha=0.9) and (5, T=1, alpha=0)
    for key, K, alpha, temp in (("beam", 8, 0.9, 1.7), ("beam5", 5, 0.0, 1.0)):
        ws, fin, _ = net.beam(batch_H, proj_H, 25, K, alpha, temp, 1, 2, None)
        ref_i, ref_l = g[f"{tag}_{key}_ids"], g[f"{tag}_{key}_logits"]
        tr = ref_i.shape[1]
        fin_h = fin.cpu().numpy()
        assert min(int(fin_h.max()), 25) == tr, (fin_h, tr)
        trun = torch.full((B,), tr, dtype=torch.int32).cuda()
        bl, bi = net.beam_finalize(ws, B, 25, K, trun)
        assert np.array_equal(bi.cpu().numpy()[:, :tr], ref_i), key
        assert np.abs(bl.cpu().numpy()[:, :tr] - ref_l).max() < 1e-3 * max(1.0, np.abs(ref_l).max()), key


@pytest.mark.parametrize("mode", ["greedy", "beam"])
def test_trba_predict_matches_oracle_text_and_confidence(env, mode):
    """40 crops, batch_size=32 (two reference chunks with different run lengths): identical texts (conftest.compare_texts),
    confidences within 1e-4.  Planted-decoder weights: decisions carry margins like a trained checkpoint's."""
    from conftest import compare_texts
    from manuscript_ocr_amd import synth
    from manuscript_ocr_amd.recognizers import TRBA
    from oracle import imgproc
    otm = env
    seed = 20260128
    sd = synth.trba_state_dict_confident(194, 256, seed=seed)
    cfg = {"img_h": 32, "img_w": 100, "max_len": 25, "hidden_size": 256}
    rec = TRBA(state_dict=sd, config=cfg, device="cuda")
    rng = np.random.default_rng(3)
    crops = []
    for i, c in enumerate(synth.synth_crops(77, 40, 32, 100)):
        hh, ww = int(rng.integers(20, 60)), int(rng.integers(40, 220))  # exercise AREA and LINEAR resizes
        crops.append(imgproc.resize_linear_u8(c, ww, hh))
    got = rec.predict(crops, batch_size=32, mode=mode)
    ref_net = otm.TRBANet(194, 256)
    ref_net.load_state_dict(sd, strict=True)
    ref_net.eval()
    itos, _ = otm.load_charset(CHARSET)
    exp = []
    for c0 in range(0, 40, 32):
        x = torch.from_numpy(np.stack([imgproc.trba_preprocess(c, 32, 100) for c in crops[c0:c0 + 32]]))
        with torch.no_grad():
            if mode == "greedy":
                lg, ids = ref_net(x, max_len=25, mode="greedy")
            else:
                lg, ids = ref_net(x, max_len=25, mode="beam", beam_size=8, alpha=0.9, temperature=1.7)
        exp += otm.texts_and_confidences(lg, ids, itos, 0, 2, None)
    same = compare_texts([r["text"] for r in got], exp, itos)
    assert len({r["text"] for r in exp}) >= 8, "degenerate fixture: texts do not vary"
    np.testing.assert_allclose([got[i]["confidence"] for i in same], [exp[i]["confidence"] for i in same], atol=1e-4)


_ORACLE_DECODES = {}   # (seed, N, mode, max_len) -> (oracle net, rows, batch_H): the CPU decode of one case is the same for every device variant


def _random_weight_case(otm, seed, N, mode, rec, canv=None, max_len=25):
    """GPU decode (ids, run lengths, logits, confidences), the oracle's rows for N synthetic crops (all-random weights), and the
    CALIBRATED logit bounds of that case (oracle/decode_check.py::calibrated_logit_bounds): the device's encoder output is
    read back, its error against the oracle's is measured, and the oracle's own decoder is re-run on inputs perturbed by noise
    of that size — the device's logit-error distribution has to lie within 2x the oracle's own response."""
    from conftest import calibrated_logit_bounds, oracle_decode_chunks
    sd = synth.trba_state_dict(194, 256, seed=seed)
    key = (seed, N, mode, max_len) if canv is None else None
    if canv is None:
        canv = synth.synth_crops(seed % 1000 + 5, N, 32, 100)
    if key in _ORACLE_DECODES:
        ref_net, exp, keep = _ORACLE_DECODES[key]
    else:
        ref_net = otm.TRBANet(194, 256)
        ref_net.load_state_dict(sd, strict=True)
        ref_net.eval()
        keep = []
        exp = oracle_decode_chunks(ref_net, _x_from_canvases(canv), mode, max_len=max_len, keep_batch_H=keep)
        if key is not None:
            _ORACLE_DECODES[key] = (ref_net, exp, keep)
    canv_dev = torch.from_numpy(canv).cuda()
    dev_bH = rec.model.encode(canv_dev)[0].float().cpu().numpy()
    cal = calibrated_logit_bounds(ref_net, np.concatenate(keep), dev_bH, exp, mode, max_len=max_len)
    ids, trun, conf, lg = rec.recognize_canvases(canv_dev, batch_size=32, mode=mode, return_logits=True)
    return ids, trun, conf, lg, exp, cal


def _assert_near_tie_parity(rep, N, what, cal, got=None, mode=None):
    """(1) DECODER parity: the device's decode against the ORACLE's decoder started from the device's own encoder output
    (cal["rows_on_dev_H"]): every row identical up to near-ties of that decode (margin < TIE_TOL), logits within 1e-4 of
    max |logit| — both decoders see the same input, only the decoder's own f32 rounding differs.
    (2) END TO END against the oracle's decode of ITS encoder output: zero differences that are neither near-ties of the oracle's
    own decode nor rows explained by the encoder's in-tolerance error alone (identical to (1)'s reference; oracle/decode_check.py::
    admit_encoder_sensitive); those two kinds together capped at 3 % (+1); the per-row logit error within the calibrated bounds
    (median, 90th percentile and maximum <= 2x the oracle decoder's own response to an encoder-output perturbation of the
    measured size)."""
    from conftest import admit_encoder_sensitive, compare_decodes
    admitted = []
    decoder_identical = False
    if got is not None:
        ids, trun, lg = got
        rep_dec = compare_decodes(ids, trun, lg, cal["rows_on_dev_H"], mode, logit_rtol=DECODER_LOGIT_RTOL)
        msg_d = (f"{what} [decoder only]: {len(rep_dec['same'])}/{N} rows identical to the oracle decoder on the device's batch_H, ties "
                 f"{rep_dec['ties']}, hard {rep_dec['hard']}, max logit err {rep_dec['max_logit_err_rel']:.2e}")
        print(msg_d)
        assert not rep_dec["hard"] and len(rep_dec["ties"]) <= 1 + (3 * N) // 100, msg_d
        rep["hard"], admitted = admit_encoder_sensitive(rep, rep_dec)
        decoder_identical = len(rep_dec["same"]) == N
    msg = (f"{what}: {len(rep['same'])}/{N} rows identical, ties {rep['ties']}, encoder-sensitive rows {admitted}, run-length-only rows "
           f"{rep['run_length_only']}, hard {rep['hard']}, max logit err {rep['max_logit_err_rel']:.2e} of max|logit|")
    print(msg)
    assert not rep["hard"], msg
    assert len(rep["ties"]) + len(admitted) <= 1 + (3 * N) // 100, msg
    rep["encoder_sensitive"] = admitted
    err = np.array(rep["row_logit_err_rel"])
    q = lambda v, p: float(np.quantile(v, p))
    print(f"{what}: encoder error {cal['enc_err_rel']:.2e} of max|batch_H| (rms {cal['enc_rms_rel']:.2e}); logit error / max|logit| "
          f"per row  device: p50 {q(err, .5):.2e} p90 {q(err, .9):.2e} max {err.max():.2e} | oracle under noise of that size: "
          f"p50 {q(cal['oracle_noise'], .5):.2e} p90 {q(cal['oracle_noise'], .9):.2e} max {cal['oracle_noise'].max():.2e} | oracle "
          f"decoder on the device's batch_H: p50 {q(cal['oracle_on_dev_H'], .5):.2e} p90 {q(cal['oracle_on_dev_H'], .9):.2e} "
          f"max {cal['oracle_on_dev_H'].max():.2e}")
    assert cal["enc_err_rel"] < ENCODER_ERR_REL, (what, cal["enc_err_rel"])
    if got is not None:
        # per row, no statistics: |device - oracle| <= |device - oracle decoder on the device's batch_H| + |that decoder's output - oracle|
        # (triangle inequality; the first term is bounded by DECODER_LOGIT_RTOL above, the second is the reference decoder's own response
        # to the encoder's in-tolerance error)
        slack = err - (np.asarray(cal["oracle_on_dev_H"]) + DECODER_LOGIT_RTOL)
        assert slack.max() <= 0, (what, int(slack.argmax()), float(slack.max()))
    # the response itself against the noise model (Gaussian perturbation of the measured per-row size): the median always; the 90th
    # percentile and the maximum for every case in which the DECODER-ONLY comparison above has at least one differing row.  When
    # the device's decode is row-for-row identical to the oracle's decoder on the device's own batch_H, every bit of `err` is the
    # reference decoder's own response to the encoder's in-tolerance error (the per-row triangle bound above already holds it to
    # that response), and its upper quantiles against a Gaussian noise model carry nothing about the device — no gate on N.
    assert q(err, .5) <= cal["p50"], (what, q(err, .5), cal["p50"])
    if not decoder_identical:
        assert q(err, .9) <= cal["p90"] and err.max() <= cal["max"], (what, q(err, .9), err.max(), cal["p90"], cal["max"])


RANDOM_WEIGHT_SEED = 20260128
# The device's encoder output (batch_H) against the oracle's, relative to max |batch_H|: a stated bound on the f32 rounding
# difference of the SE-ResNet31 + 2 BiLSTMs (measured 2e-5 .. 6e-5 on these weights).  Everything downstream of it is bounded
# by calibration (calibrated_logit_bounds), not by a chosen constant.
ENCODER_ERR_REL = 2e-4
# Decoder alone (device decode against the oracle's decoder started from the device's own batch_H): stated bound on the f32 rounding
# difference of 25-41 chained attention / LSTM steps, relative to max |logit|: 2x the largest measured value (1.1e-5 .. 8.6e-5 over
# the ten cases of this file, gpurun_out/r3_trba3.log; ids 256/256, 40/40, 96/96 identical with no ties).
DECODER_LOGIT_RTOL = 2e-4


@pytest.mark.parametrize("mode", ["greedy", "beam"])
def test_trba_random_weights_decode_parity(env, mode):
    """ALL-RANDOM weights (synth.trba_state_dict: x6 recurrent gain, every character an arg-max over near-Gaussian logits —
    the most rounding-sensitive decoder we can build), 160 crops (256 until round 4: the driver's GPU tier allows the suite 900 s, and
    bench.py's cpu_baseline runs the same comparison on 256), the reference's 32-row chunks.  Every row must reproduce
    the oracle's ids at every generated step; the only admitted difference is at the FIRST differing step and only where the
    oracle's own decision margin there is below TIE_TOL (conftest.compare_decodes).  Logits up to that step: this decoder is
    chaotic (x6 recurrent gain, 25-26 chained steps), so a fixed tolerance would be "whatever passed"; the bound is CALIBRATED
    instead: the device's batch_H is read back, its error against the oracle's encoder output is measured
    (and bounded by ENCODER_ERR_REL), the oracle's own decoder is re-run on its batch_H perturbed by noise of that size, and
    the device's per-row logit-error distribution (median, p90, maximum) has to lie within 2x the oracle's own.  Confidences
    of identical rows: a mean of softmax probabilities whose logits carry that error — within 2x the same calibrated p90 /
    maximum (probabilities move by at most the logit error in absolute terms)."""
    from conftest import compare_decodes
    from manuscript_ocr_amd.recognizers import TRBA
    otm = env
    N = 160
    rec = TRBA(state_dict=synth.trba_state_dict(194, 256, seed=RANDOM_WEIGHT_SEED),
               config={"img_h": 32, "img_w": 100, "max_len": 25, "hidden_size": 256}, device="cuda")
    ids, trun, conf, lg, exp, cal = _random_weight_case(otm, RANDOM_WEIGHT_SEED, N, mode, rec)
    rep = compare_decodes(ids, trun, lg, exp, mode, logit_rtol=cal["max"])
    _assert_near_tie_parity(rep, N, f"random weights / {mode}", cal, (ids, trun, lg), mode)
    rep["chunks_with_ties"] |= {exp[h[0]]["chunk"] for h in rep["encoder_sensitive"]}
    assert len({tuple(e["ids"].tolist()) for e in exp}) > N // 2, "degenerate fixture: decodes do not vary"
    itos, _ = otm.load_charset(CHARSET)
    by_chunk = {}
    for i, e in enumerate(exp):
        by_chunk.setdefault(e["chunk"], []).append(i)
    for ch, rows in by_chunk.items():  # confidences depend on the chunk's run length: compare chunks without a tie row
        if ch in rep["chunks_with_ties"]:
            continue
        lg_c = torch.from_numpy(np.stack([exp[i]["logits"] for i in rows]))
        ids_c = torch.from_numpy(np.stack([exp[i]["ids"] for i in rows]))
        ref = otm.texts_and_confidences(lg_c, ids_c, itos, 0, 2, None)
        got_texts = rec.texts(ids[rows], trun[rows])
        assert got_texts == [r["text"] for r in ref]
        dconf = np.abs(conf[rows] - np.array([r["confidence"] for r in ref]))
        scale = max(1.0, max(float(np.abs(exp[i]["logits"]).max()) for i in rows))  # logit error bounds are relative to max|logit|
        assert np.quantile(dconf, 0.9) <= 2 * cal["p90"] * scale and dconf.max() <= 2 * cal["max"] * scale, (ch, dconf.max(), cal["max"], scale)


@pytest.mark.parametrize("mode", ["greedy", "beam"])
def test_trba_random_weights_recorded_round1_case(env, mode):
    """The input that was red in round 1 (gpurun_out/s2_tests.log: 40 resized crops of synth_crops(77), all-random weights,
    texts compared by equality; crop 29 diverged at character 21) under the first-differing-step rule: every differing row
    must be a near-tie of the oracle's own decode, and its margin is printed for the record."""
    from conftest import compare_decodes, oracle_decode_chunks
    from manuscript_ocr_amd.recognizers import TRBA
    from oracle import imgproc
    otm = env
    sd = synth.trba_state_dict(194, 256, seed=RANDOM_WEIGHT_SEED)
    rec = TRBA(state_dict=sd, config={"img_h": 32, "img_w": 100, "max_len": 25, "hidden_size": 256}, device="cuda")
    rng = np.random.default_rng(3)
    crops = []
    for c in synth.synth_crops(77, 40, 32, 100):
        hh, ww = int(rng.integers(20, 60)), int(rng.integers(40, 220))
        crops.append(imgproc.resize_linear_u8(c, ww, hh))
    canv = np.stack([imgproc.resize_and_pad(c, 32, 100) for c in crops])
    ids, trun, conf, lg, exp, cal = _random_weight_case(otm, RANDOM_WEIGHT_SEED, 40, mode, rec, canv=canv)
    rep = compare_decodes(ids, trun, lg, exp, mode, logit_rtol=cal["max"])
    _assert_near_tie_parity(rep, 40, f"round-1 recorded case / {mode}", cal, (ids, trun, lg), mode)


@pytest.mark.parametrize("mode", ["greedy", "beam"])
def test_trba_shipped_config_32x128_maxlen40(env, mode):
    """The configuration the reference ships (recognizers/_trba/configs/config.json: img_h 32, img_w 128, max_len 40,
    hidden_size 256): T_enc = 17, 40 / 41 decode steps.  All-random weights, 96 crops, near-tie rule against the oracle."""
    from conftest import compare_decodes, oracle_decode_chunks
    from manuscript_ocr_amd.recognizers import TRBA
    otm = env
    N = 96
    sd = synth.trba_state_dict(194, 256, seed=RANDOM_WEIGHT_SEED)
    rec = TRBA(state_dict=sd, config={"img_h": 32, "img_w": 128, "max_len": 40, "hidden_size": 256}, device="cuda")
    canv = synth.synth_crops(123, N, 32, 128)
    ids, trun, conf, lg, exp, cal = _random_weight_case(otm, RANDOM_WEIGHT_SEED, N, mode, rec, canv=canv, max_len=40)
    assert ids.shape[1] == (41 if mode == "greedy" else 40)
    rep = compare_decodes(ids, trun, lg, exp, mode, logit_rtol=cal["max"])
    _assert_near_tie_parity(rep, N, f"shipped config 32x128 / max_len 40 / {mode}", cal, (ids, trun, lg), mode)


def test_trba_random_weights_three_way(env, monkeypatch):
    """The same 96 all-random-weight crops (three reference chunks; 256 in rounds 1-3 — the driver's GPU tier allows the whole suite
    900 s and each variant re-runs the CPU oracle's calibration) through (a) the default path (Winograd 3x3 layers, matrix-core beam kernel),
    (b) direct convolutions only (MSOCR_WINOGRAD_MIN_CIN=0), (c) the VALU beam kernel (MSOCR_BEAM_MFMA=0): each against the
    oracle under the near-tie rule, and pairwise: rows that are identical to the oracle in two variants are identical to
    each other, so the variants can only differ on the oracle's near-tie rows."""
    from conftest import compare_decodes
    from manuscript_ocr_amd import ops
    from manuscript_ocr_amd.recognizers import TRBA
    otm = env
    N, cfg = 96, {"img_h": 32, "img_w": 100, "max_len": 25, "hidden_size": 256}
    sd = synth.trba_state_dict(194, 256, seed=RANDOM_WEIGHT_SEED)
    results = {}
    rec_w = TRBA(state_dict=sd, config=cfg, device="cuda")
    results["winograd+mfma-beam"] = _random_weight_case(otm, RANDOM_WEIGHT_SEED, N, "beam", rec_w)
    monkeypatch.setenv("MSOCR_BEAM_MFMA", "0")
    results["winograd+valu-beam"] = _random_weight_case(otm, RANDOM_WEIGHT_SEED, N, "beam", rec_w)
    monkeypatch.delenv("MSOCR_BEAM_MFMA")
    monkeypatch.setattr(ops, "WINOGRAD_MIN_CIN", 0)
    rec_d = TRBA(state_dict=sd, config=cfg, device="cuda")
    assert not any(hasattr(w, "_msocr_wino") for w, _ in rec_d.model.P.values())
    results["direct+mfma-beam"] = _random_weight_case(otm, RANDOM_WEIGHT_SEED, N, "beam", rec_d)
    same = {}
    for name, (ids, trun, conf, lg, exp, cal) in results.items():
        rep = compare_decodes(ids, trun, lg, exp, "beam", logit_rtol=cal["max"])
        _assert_near_tie_parity(rep, N, name, cal, (ids, trun, lg), "beam")
        same[name] = set(rep["same"])
    names = list(results)
    for a in names:
        for b in names:
            if a < b:
                for i in same[a] & same[b]:
                    ta = int(results[a][1][i])
                    assert ta == int(results[b][1][i]) and np.array_equal(results[a][0][i][:ta], results[b][0][i][:ta]), (a, b, i)


def test_trba_bf16_cnn_close(env):
    """bf16 CNN (f32 accumulate; recurrent/attention stay f32).  Stated tolerance: CNN features within
    3 % of their max; the encoder output, which the x6-scaled synthetic LSTM weights amplify, within 15 %."""
    from manuscript_ocr_amd import synth
    from manuscript_ocr_amd.recognizers._trba.net import TrbaNet
    otm = env
    sd = synth.trba_state_dict(194, 256, seed=5)
    canv = synth.synth_crops(9, 8, 32, 100)
    ref_net = _oracle_net(otm, 5)
    with torch.no_grad():
        x = _x_from_canvases(canv)
        ref_f = ref_net.cnn(x).permute(0, 2, 3, 1).numpy()
        ref = ref_net.encode(x).numpy()
    net = TrbaNet(sd, 194, 256, torch.bfloat16)
    cd = torch.from_numpy(canv).cuda()
    f = net.cnn(cd).float().cpu().numpy()
    assert np.abs(f - ref_f).max() < 0.03 * np.abs(ref_f).max()
    batch_H, _ = net.encode(cd)
    err = np.abs(batch_H.cpu().numpy() - ref).max()
    assert err < 0.15 * max(1.0, np.abs(ref).max()), err


def test_greedy_matrix_core_kernel_matches_the_valu_kernel_and_the_oracle(env, monkeypatch):
    """mode="greedy" on the matrix cores (attn_greedy_mfma_kernel, 32 crops per workgroup, the default since round 4) against the
    round-1 VALU kernel (MSOCR_GREEDY_MFMA=0) and against the oracle's greedy decode of the DEVICE's encoder output: 70 crops (two
    full row blocks + a partial one), planted decoder — ids identical at every one of the 26 steps (the rows run on after their EOS,
    as model.py:254 does), logits within 1e-3 of the largest logit."""
    from manuscript_ocr_amd import synth
    from manuscript_ocr_amd.recognizers._trba.net import TrbaNet
    otm = env
    sd = synth.trba_state_dict_confident(194, 256, seed=11)
    net = TrbaNet(sd, 194, 256, torch.float32)
    B, max_len = 70, 25
    cd = torch.from_numpy(synth.synth_crops(21, B, 32, 100)).cuda()
    batch_H, proj_H = net.encode(cd)
    lg_m, ids_m = net.greedy(batch_H, proj_H, max_len, 1, 2, None)
    monkeypatch.setenv("MSOCR_GREEDY_MFMA", "0")
    lg_v, ids_v = net.greedy(batch_H, proj_H, max_len, 1, 2, None)
    monkeypatch.delenv("MSOCR_GREEDY_MFMA")
    lg_m, ids_m, lg_v, ids_v = lg_m.cpu().numpy(), ids_m.cpu().numpy(), lg_v.cpu().numpy(), ids_v.cpu().numpy()
    assert ids_m.shape == (B, max_len + 1) and np.array_equal(ids_m, ids_v)
    assert np.abs(lg_m - lg_v).max() < 1e-3 * np.abs(lg_v).max()
    ref_net = otm.TRBANet(194, 256)
    ref_net.load_state_dict(sd, strict=True)
    ref_net.eval()
    with torch.no_grad():
        rl, ri = ref_net.attn.greedy(batch_H.cpu(), max_len=max_len)
    t_run = ri.shape[1]  # the oracle stops where the reference does; the device ran all steps
    assert np.array_equal(ids_m[:, :t_run], ri.numpy())
    assert np.abs(lg_m[:, :t_run] - rl.numpy()).max() < 1e-3 * np.abs(rl.numpy()).max()


def test_beam_kernels_agree_and_early_exit_changes_nothing(env, monkeypatch):
    """Matrix-core beam kernel (default) vs the VALU kernel (MSOCR_BEAM_MFMA=0): same ids / finish steps, logits within 1e-3 of
    the largest logit; and the chunk-level early exit (rows grouped like the reference's batch_size chunks) leaves every
    output the finalize step reads (t < chunk run length) bit-identical to the run over all steps."""
    from manuscript_ocr_amd import synth
    from manuscript_ocr_amd.recognizers._trba.net import TrbaNet
    net = TrbaNet(synth.trba_state_dict_confident(194, 256, seed=11), 194, 256, torch.float32)
    B, K, steps = 70, 8, 25
    cd = torch.from_numpy(synth.synth_crops(21, B, 32, 100)).cuda()
    batch_H, proj_H = net.encode(cd)

    def run(chunks=None):
        ws, fin, _ = net.beam(batch_H, proj_H, steps, K, 0.9, 1.7, 1, 2, None, chunks)
        return ws, fin.cpu().numpy()

    def finalize(ws, fin_h, sizes):
        trun, o = np.empty(B, dtype=np.int32), 0
        for sz in sizes:
            trun[o:o + sz] = fin_h[o:o + sz].max()
            o += sz
        lg, ids = net.beam_finalize(ws, B, steps, K, torch.from_numpy(trun).cuda())
        return trun, lg.cpu().numpy(), ids.cpu().numpy()

    sizes = [32, 32, 6]
    ws_m, fin_m = run()
    trun, lg_m, ids_m = finalize(ws_m, fin_m, sizes)
    assert trun.max() < steps, "fixture must finish early for the early-exit check to mean anything"
    # (1) VALU kernel
    monkeypatch.setenv("MSOCR_BEAM_MFMA", "0")
    ws_v, fin_v = run()
    monkeypatch.delenv("MSOCR_BEAM_MFMA")
    _, lg_v, ids_v = finalize(ws_v, fin_v, sizes)
    assert np.array_equal(fin_m, fin_v) and np.array_equal(ids_m, ids_v)
    for b in range(B):
        t = trun[b]
        assert np.abs(lg_m[b, :t] - lg_v[b, :t]).max() < 1e-3 * max(1.0, np.abs(lg_v[b, :t]).max())
    # (2) early exit
    ids_c = torch.from_numpy(np.repeat(np.arange(3), sizes).astype(np.int32)).cuda()
    size_c = torch.tensor(sizes, dtype=torch.int32).cuda()
    state = torch.zeros(6, dtype=torch.int32).cuda()
    ws_e, fin_e = run((ids_c, size_c, state))
    _, lg_e, ids_e = finalize(ws_e, fin_e, sizes)
    st = state.cpu().numpy()
    assert np.array_equal(fin_e, fin_m) and st[0::2].tolist() == sizes and st[1::2].tolist() == [int(trun[0]), int(trun[32]), int(trun[64])]
    assert np.array_equal(ids_e, ids_m)
    for b in range(B):
        assert np.array_equal(lg_e[b, :trun[b]], lg_m[b, :trun[b]])


def test_split_operand_beam_kernel_matches_the_exact_one_on_every_beam(env, monkeypatch):
    """Guard for the split-operand form of the matrix-core beam kernel (three bf16 terms per f32 operand, csrc/attn_beam_mfma.hip):
    EVERY beam's logits of the first two steps — not only the best path the finalize step returns — against the exact-f32 MFMA form
    (MSOCR_BEAM_SPLIT=0) on 1920 crops, three launches.  Steps 0 and 1 come before any near-tie can reorder beams, so the bound is
    tight: 2e-5 of the largest logit (measured 3e-6).  A packed-f32 code shape in the hoisted context sum once produced wrong gate
    pre-activations in ~0.5 % of the rows of one crop slot (see the note at add_np / fmac_np); this comparison is what shows it."""
    from manuscript_ocr_amd import synth
    from manuscript_ocr_amd.recognizers._trba.net import TrbaNet
    B, V, S, K = 1920, 194, 4, 8
    net = TrbaNet(synth.trba_state_dict(V, 256, seed=1), V, 256, torch.float32)
    assert net._asw is not None, "precision fp32 must carry the split decoder weights"
    g = torch.Generator(device="cpu").manual_seed(0)
    bH = torch.randn(B, 13, 256, generator=g).cuda()
    pH = torch.randn(B, 13, 256, generator=g).cuda()

    def run():
        ws, _, _ = net.beam(bH, pH, S, K, 0.9, 1.7, 1, 2, None)
        torch.cuda.synchronize()
        n = B * S * K * V
        lg = ws[: 4 * n].view(torch.float32).view(B, S, K, V).clone()
        bt = ws[4 * n: 4 * n + 8 * B * S * K].view(torch.int32).view(2, B, S, K).clone()
        return lg, bt

    monkeypatch.setenv("MSOCR_BEAM_SPLIT", "0")
    ref_lg, ref_bt = run()
    monkeypatch.setenv("MSOCR_BEAM_SPLIT", "1")
    scale = float(ref_lg[:, :2].abs().max())
    for rep in range(3):
        lg, bt = run()
        assert torch.equal(bt[:, :, 0], ref_bt[:, :, 0]), "step-0 back-pointers / tokens differ"
        d = (lg[:, :2] - ref_lg[:, :2]).abs().amax(dim=-1)      # [B][2][K]
        bad = (d > 2e-5 * scale).nonzero().tolist()
        assert not bad, f"launch {rep}: {len(bad)} (crop, step, beam) rows off, first {bad[:8]}, worst {float(d.max()):.3e} of {scale:.2f}"


def test_small_device_batches_give_the_same_results(env):
    """Launch splitting (device_batch) is invisible: 120 crops in launches of <= 50 rows (aligned to the reference's 32-row
    chunks: 32 + 32 + 32 + 24) == one launch, in both modes."""
    from manuscript_ocr_amd import synth
    from manuscript_ocr_amd.recognizers import TRBA
    sd = synth.trba_state_dict_confident(194, 256, seed=5)
    cfg = {"img_h": 32, "img_w": 100, "max_len": 25, "hidden_size": 256}
    one = TRBA(state_dict=sd, config=cfg, device="cuda")
    many = TRBA(state_dict=sd, config=cfg, device="cuda", device_batch=50)
    crops = list(synth.synth_crops(31, 120, 32, 100))
    for mode in ("beam", "greedy"):
        a, b = one.predict(crops, mode=mode), many.predict(crops, mode=mode)
        assert [r["text"] for r in a] == [r["text"] for r in b]
        np.testing.assert_allclose([r["confidence"] for r in a], [r["confidence"] for r in b], rtol=0, atol=1e-6)
    assert many._device_batches(120, [(0, 120)], 32)[0] == [(0, 32), (32, 64), (64, 96), (96, 120)]


def test_random_span_layouts_match_the_full_length_decode(env, monkeypatch):
    """Randomised page layouts (spans of 1..90 crops, so chunks of 1..32 rows, ragged workgroups, launches of <= 64 rows):
    ids, run lengths and confidences with the chunk-level early exit == the VALU kernel running all steps."""
    from manuscript_ocr_amd import synth
    from manuscript_ocr_amd.recognizers import TRBA
    sd = synth.trba_state_dict_confident(194, 256, seed=13)
    rec = TRBA(state_dict=sd, config={"img_h": 32, "img_w": 100, "max_len": 25, "hidden_size": 256}, device="cuda", device_batch=64)
    rng = np.random.default_rng(99)
    for trial in range(4):
        counts = [int(c) for c in rng.choice([1, 2, 3, 5, 31, 32, 33, 64, 90], size=int(rng.integers(1, 5)))]
        N = sum(counts)
        spans, o = [], 0
        for c in counts:
            spans.append((o, c))
            o += c
        canv = torch.from_numpy(synth.synth_crops(1000 + trial, N, 32, 100)).cuda()
        ids_a, trun_a, conf_a = rec.recognize_canvases(canv, spans=spans)
        monkeypatch.setenv("MSOCR_BEAM_MFMA", "0")
        ids_b, trun_b, conf_b = rec.recognize_canvases(canv, spans=spans)
        monkeypatch.delenv("MSOCR_BEAM_MFMA")
        assert np.array_equal(trun_a, trun_b), (counts, trun_a, trun_b)
        assert np.array_equal(ids_a, ids_b), counts
        np.testing.assert_allclose(conf_a, conf_b, rtol=0, atol=1e-5)


@pytest.mark.parametrize("hidden,V,beam,mode", [
    (128, 194, 8, "greedy"), (128, 194, 8, "beam"),   # hidden_size 128
    (512, 194, 8, "beam"), (512, 194, 1, "greedy"),   # hidden_size 512
    (256, 400, 8, "beam"), (256, 400, 1, "greedy"),   # a charset above 256 tokens
    (256, 194, 12, "beam"),                           # a beam width above 8 (the reference's Optuna script sweeps 2..12)
    (256, 194, 3, "beam"),                            # a narrow beam on the matrix-core kernel (unused beam slots)
])
def test_trba_shapes_beyond_the_default_kernels(env, hidden, V, beam, mode, tmp_path):
    """Recogniser shapes the reference accepts and the round-2 kernels refused (missing 3): hidden_size from the
    checkpoint's config (recognizers/_trba/__init__.py:142-151), charset size, beam_size of TRBA.predict (:295-299).  They run on
    csrc/attn_general.hip + the templated BiLSTM kernel.  All-random weights, 48 crops; checked like the default shapes: decoder
    parity against the oracle's decoder on the device's own encoder output (ids identical up to near-ties), end-to-end under the
    near-tie / encoder-sensitive rule with calibrated logit bounds."""
    from conftest import calibrated_logit_bounds, compare_decodes, oracle_decode_chunks
    from manuscript_ocr_amd.recognizers import TRBA
    otm = env
    N = 48
    sd = synth.trba_state_dict(V, hidden, seed=77 + hidden + V)
    cfg = {"img_h": 32, "img_w": 100, "max_len": 25, "hidden_size": hidden}
    charset = None
    if V != 194:  # a charset file of V tokens: <PAD> <SOS> <EOS> + V - 3 symbols
        charset = tmp_path / "charset.txt"
        charset.write_text("\n".join(["<PAD>", "<SOS>", "<EOS>"] + [chr(0x4E00 + i) for i in range(V - 3)]) + "\n", encoding="utf-8")
    rec = TRBA(state_dict=sd, config=cfg, device="cuda", charset_path=None if charset is None else str(charset))
    assert rec.hidden_size == hidden and rec.model.V == V
    canv = synth.synth_crops(hidden + V + beam, N, 32, 100)
    ref_net = otm.TRBANet(V, hidden)
    ref_net.load_state_dict(sd, strict=True)
    ref_net.eval()
    keep = []
    exp = oracle_decode_chunks(ref_net, _x_from_canvases(canv), mode, keep_batch_H=keep, beam_size=beam)
    canv_dev = torch.from_numpy(canv).cuda()
    dev_bH = rec.model.encode(canv_dev)[0].float().cpu().numpy()
    cal = calibrated_logit_bounds(ref_net, np.concatenate(keep), dev_bH, exp, mode, beam_size=beam)
    ids, trun, conf, lg = rec.recognize_canvases(canv_dev, batch_size=32, mode=mode, beam_size=beam, return_logits=True)
    rep = compare_decodes(ids, trun, lg, exp, mode, logit_rtol=cal["max"])
    _assert_near_tie_parity(rep, N, f"hidden {hidden} / V {V} / beam {beam} / {mode}", cal, (ids, trun, lg), mode)
    assert len({tuple(e["ids"].tolist()) for e in exp}) > N // 3, "degenerate fixture: decodes do not vary"


def test_trba_shape_limits_raise_like_bad_arguments(env):
    """Beyond the kernels' shapes the constructor / predict raise ValueError (nothing is silently narrowed)."""
    from manuscript_ocr_amd.recognizers import TRBA
    cfg = {"img_h": 32, "img_w": 100, "max_len": 25, "hidden_size": 100}
    with pytest.raises(ValueError, match="hidden_size"):
        TRBA(state_dict=synth.trba_state_dict(194, 64, seed=1), config=cfg, device="cuda")
    rec = TRBA(state_dict=synth.trba_state_dict(194, 512, seed=1), config={**cfg, "hidden_size": 512}, device="cuda")
    with pytest.raises(ValueError, match="beam_size"):
        rec.predict([synth.synth_crops(1, 1, 32, 100)[0]], beam_size=12)  # 12 x 512 > 4096
