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


def oracle_decode_chunks(trba_net, x_all, mode, chunk=32, max_len=25):
    """The reference's TRBA.predict loop (recognizers/_trba/__init__.py:374-412) on the CPU oracle: one model call per
    `chunk` rows.  Returns a list with one dict per row: ids [T_run], logits [T_run, V], and the margins of the decisions the
    oracle's own decode took for that row (oracle/trba_model.py `diag`)."""
    import numpy as np
    import torch
    rows = []
    for c0 in range(0, x_all.shape[0], chunk):
        d = {}
        with torch.no_grad():
            if mode == "greedy":
                lg, ids = trba_net(x_all[c0:c0 + chunk], max_len=max_len, mode="greedy", diag=d)
            else:
                lg, ids = trba_net(x_all[c0:c0 + chunk], max_len=max_len, mode="beam", beam_size=8, alpha=0.9, temperature=1.7, diag=d)
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
