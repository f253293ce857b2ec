"""Decode-parity checker (TEST INFRASTRUCTURE): row-by-row comparison of a device TRBA decode with the CPU oracle's on ANY
weights, with the first-differing-step near-tie rule.  Used by tests/ (through tests/conftest.py) and by bench.py's
cpu_baseline leg; never by the product.

Reference decisions it audits: Attention._greedy_decode arg-max (recognizers/_trba/model/model.py:245-247),
Attention._beam_decode top-k (model.py:161-163) and final arg-max (model.py:218); TRBA.predict's per-chunk model calls
(recognizers/_trba/__init__.py:374-412).
"""

# Largest margin of an oracle decision (temperature-scaled logits, |logit| ~ 5..40) that counts as a tie between two f32
# implementations of the same network.
TIE_TOL = 5e-3


def oracle_decode_chunks(trba_net, x_all, mode, chunk=32, max_len=25, batch_H=None, keep_batch_H=None, beam_size=8):
    """The reference's TRBA.predict loop (recognizers/_trba/__init__.py:374-412) on the CPU oracle: one model call per
    `chunk` rows.  Returns a list with one dict per row: ids [T_run], logits [T_run, V], and the margins of the decisions the
    oracle's own decode took for that row (oracle/trba_model.py `diag`).
    `batch_H` (optional, [N, T, H]): decode from THIS encoder output instead of encoding x_all (TRBANet.forward is
    encode + attn, so with batch_H = the oracle's own chunk-wise encode the result is the same); `keep_batch_H` (a list)
    receives the oracle's chunk-wise encoder outputs."""
    import numpy as np
    import torch
    rows = []
    n = x_all.shape[0] if batch_H is None else batch_H.shape[0]
    for c0 in range(0, n, chunk):
        d = {}
        with torch.no_grad():
            enc = trba_net.encode(x_all[c0:c0 + chunk]) if batch_H is None else torch.as_tensor(batch_H[c0:c0 + chunk])
            if keep_batch_H is not None:
                keep_batch_H.append(enc.numpy().copy())
            if mode == "greedy":
                lg, ids = trba_net.attn.greedy(enc, max_len, d)
            else:
                lg, ids = trba_net.attn.beam(enc, max_len, beam_size, 0.9, 1.7, d)
        lg, ids = lg.numpy(), ids.numpy()
        for j in range(ids.shape[0]):
            r = {"ids": ids[j], "logits": lg[j], "chunk": c0 // chunk}
            if mode == "greedy":
                r["top2_margin"] = d["top2_margin"][j]
            else:
                r["boundary_gap"], r["beam_scores"], r["beam_tokens"] = d["boundary_gap"][j], d["beam_scores"][j], d["beam_tokens"][j]
            rows.append(r)
    return rows


def compare_decodes(got_ids, got_trun, got_logits, exp_rows, mode, tie_tol=TIE_TOL, logit_rtol=1e-3):
    """Row-by-row comparison of a device decode with the oracle's on ANY weights (all-random included).

    A row must reproduce the oracle's ids at every generated step.  The one admitted exception is a near-tie: at the FIRST
    differing step (same prefix on both sides, so both sides evaluated the same decision) the oracle's OWN margin for that
    decision must be below `tie_tol` —
      greedy: logit[oracle's token] - logit[device's token] at that step (model.py:245-247 arg-max);
      beam:   the device's hypothesis is one of the oracle's K final hypotheses and its score is within tie_tol of the
              best one (model.py:218 arg-max), or some top-k boundary (K-th vs (K+1)-th candidate, model.py:161) up to the
              end of the oracle's decode was within tie_tol (a hypothesis was kept / dropped on a rounding-level gap).
    Logits are compared at every step up to and including the first differing one (|diff| <= logit_rtol * max|logit|).
    Returns a report dict; the caller asserts on `hard` (differences that are NOT near-ties) and caps `ties`."""
    import numpy as np
    rep = {"rows": len(exp_rows), "same": [], "ties": [], "hard": [], "max_logit_err_rel": 0.0, "chunks_with_ties": set(),
           "row_logit_err_rel": []}
    runlen = []
    for i, e in enumerate(exp_rows):
        T = min(int(got_trun[i]), len(e["ids"]))
        gi, ei = np.asarray(got_ids[i][:T]), np.asarray(e["ids"][:T])
        neq = np.flatnonzero(gi != ei)
        upto = T if len(neq) == 0 else int(neq[0]) + 1
        scale = max(1.0, float(np.abs(e["logits"][:upto]).max()))
        err = float(np.abs(got_logits[i][:upto] - e["logits"][:upto]).max()) / scale
        rep["max_logit_err_rel"] = max(rep["max_logit_err_rel"], err)
        rep["row_logit_err_rel"].append(err)
        if len(neq) == 0 and int(got_trun[i]) == len(e["ids"]):
            if err > logit_rtol:
                rep["hard"].append((i, "logits", err))
            else:
                rep["same"].append(i)
            continue
        if len(neq) == 0:  # same ids over the common steps but another run length: legitimate only when a tie row of the
            runlen.append((i, e["chunk"]))  # same chunk moved the chunk's stopping step (model.py:215,254)
            continue
        t = int(neq[0])
        if mode == "greedy":
            margin = float(e["logits"][t][ei[t]] - e["logits"][t][gi[t]])
            kind = "argmax"
        else:
            margin, kind = None, None
            Tr = len(e["ids"])
            for k in range(e["beam_tokens"].shape[0]):
                if int(got_trun[i]) == Tr and np.array_equal(e["beam_tokens"][k][:Tr], np.asarray(got_ids[i][:Tr])):
                    margin, kind = float(e["beam_scores"].max() - e["beam_scores"][k]), "final-argmax"
                    break
            if margin is None:
                margin, kind = float(e["boundary_gap"].min()), "topk-boundary"
        ok = margin < tie_tol and err <= logit_rtol
        (rep["ties"] if ok else rep["hard"]).append((i, t, kind, margin))
        if ok:
            rep["chunks_with_ties"].add(e["chunk"])
    for i, ch in runlen:
        if ch not in rep["chunks_with_ties"]:
            rep["hard"].append((i, -1, "run-length", None))
    rep["run_length_only"] = [i for i, ch in runlen if ch in rep["chunks_with_ties"]]
    return rep


def row_logit_errors(rows_a, rows_b):
    """Per-row max |logit_a - logit_b| / max(1, max|logit_b|) over the steps up to and including the first step where the ids
    differ — the quantity compare_decodes bounds (rep["row_logit_err_rel"]), here between two ORACLE decodes."""
    import numpy as np
    out = []
    for a, b in zip(rows_a, rows_b):
        T = min(len(a["ids"]), len(b["ids"]))
        neq = np.flatnonzero(np.asarray(a["ids"][:T]) != np.asarray(b["ids"][:T]))
        upto = T if len(neq) == 0 else int(neq[0]) + 1
        scale = max(1.0, float(np.abs(b["logits"][:upto]).max()))
        out.append(float(np.abs(a["logits"][:upto] - b["logits"][:upto]).max()) / scale)
    return np.array(out)


def calibrated_logit_bounds(trba_net, ref_batch_H, dev_batch_H, exp_rows, mode, chunk=32, max_len=25, draws=4, seed=0, beam_size=8):
    """Where the device's decoder-logit error may lie, DERIVED instead of chosen: the device's encoder output differs from the
    oracle's by delta = dev_batch_H - ref_batch_H (measured, f32 rounding of a different summation order).  The oracle's own
    decoder is re-run on ref_batch_H + Gaussian noise whose per-row RMS equals that row's measured RMS of delta (`draws`
    independent draws) and on dev_batch_H itself; the per-row logit error of those runs against the unperturbed oracle decode
    is the decoder's own response to an input error of the device's size.  Returns a dict with the pooled noise-response
    distribution (`oracle_noise`), the response to the device's actual delta (`oracle_on_dev_H`), the measured relative
    encoder error, and the bounds a device decode has to meet: p50 / p90 / max <= `factor` x the oracle's own (factor 2)."""
    import numpy as np
    ref_batch_H, dev_batch_H = np.asarray(ref_batch_H, dtype=np.float32), np.asarray(dev_batch_H, dtype=np.float32)
    delta = dev_batch_H - ref_batch_H
    row_rms = np.sqrt((delta.reshape(len(delta), -1) ** 2).mean(axis=1))
    rng = np.random.default_rng(seed)
    pooled = []
    for _ in range(draws):
        noise = rng.standard_normal(ref_batch_H.shape).astype(np.float32) * row_rms[:, None, None]
        pert = oracle_decode_chunks(trba_net, None, mode, chunk, max_len, batch_H=ref_batch_H + noise, beam_size=beam_size)
        pooled.append(row_logit_errors(pert, exp_rows))
    pooled = np.concatenate(pooled)
    rows_on_dev = oracle_decode_chunks(trba_net, None, mode, chunk, max_len, batch_H=dev_batch_H, beam_size=beam_size)
    on_dev = row_logit_errors(rows_on_dev, exp_rows)
    factor = 2.0
    return {"oracle_noise": pooled, "oracle_on_dev_H": on_dev, "factor": factor,
            # the ORACLE's decoder run on the DEVICE's encoder output: the reference decode of what the device decoder was given
            "rows_on_dev_H": rows_on_dev,
            "enc_err_rel": float(np.abs(delta).max() / max(1e-30, np.abs(ref_batch_H).max())),
            "enc_rms_rel": float(np.sqrt((delta ** 2).mean()) / max(1e-30, np.sqrt((ref_batch_H ** 2).mean()))),
            "p50": factor * float(np.quantile(pooled, 0.5)), "p90": factor * float(np.quantile(pooled, 0.9)),
            "max": factor * float(pooled.max())}


def admit_encoder_sensitive(rep_e2e, rep_dec):
    """End-to-end differences that are NOT near-ties of the oracle's decode (rep_e2e["hard"]) are admitted only when they are
    explained by the encoder's in-tolerance f32 error ALONE: the row must be identical, ids and run length, to the ORACLE's own
    decoder started from the DEVICE's encoder output (rep_dec = compare_decodes(device, oracle_decode_chunks(batch_H = device
    batch_H))).  For such a row the device decoder reproduces the reference decoder exactly; the two encoder outputs differ within
    the stated tolerance and the reference's decoder itself maps them to different strings.  Returns (still_hard, admitted)."""
    same_dec = set(rep_dec["same"])
    still, admitted = [], []
    for h in rep_e2e["hard"]:
        (admitted if h[0] in same_dec else still).append(h)
    return still, admitted
