"""Device-side TRBA network: SE-ResNet31 -> mean over H -> 2 x BiLSTM -> attention decoder.

Host logic only (weight folding / transposition, launch order); arithmetic is in libmsocr.so.
Mirrors /root/reference/src/manuscript/recognizers/_trba/model/model.py:387-416 and
model/seresnet31.py:180-187 with:
  * BatchNorm folded into the bias-free convs (eval mode), NHWC, implicit-GEMM MFMA convs;
  * the 3x3 C=3 stem as a 3-tap conv over a zero-bordered padded-channel canvas;
  * SE gate + residual add + ReLU fused in one kernel per block;
  * LSTM input projections and the Linear layers as MFMA GEMMs (1x1 convs: split-operand bf16x3 with f32 accumulation under
    precision="fp32", exact-f32 MFMA under "fp32-exact"), the recurrence as one persistent launch per layer (both directions);
  * the attention decoder as ONE launch for the whole step loop, i2h(batch_H) hoisted out of it,
    the one-hot matmul replaced by a row gather of W_ih (model.py:36,44: identical arithmetic).
The CNN may run in bf16 (`dtype`).  The recurrent and attention stages keep f32 state and f32 accumulation in every mode; under
precision="fp32" (the default) their matrix products are the split-operand form on the bf16 pipes (f32 result up to summation
order) and their gate nonlinearities use the hardware-rate v_exp_f32 / v_rcp_f32 (1-2 ulp); precision="fp32-exact" runs the
exact-f32 MFMA / VALU kernels with libm-grade expf / tanhf throughout.
"""
import ctypes

import numpy as np
import os

import torch

from ... import _native as nat
from ... import ops
from ...detectors._east.net import BN_EPS, pack_stem_weight, stem_view, to_khwc


def _fold(sd, conv_key, bn_key):
    w = sd[conv_key + ".weight"].float()
    gamma, beta = sd[bn_key + ".weight"].float(), sd[bn_key + ".bias"].float()
    mean, var = sd[bn_key + ".running_mean"].float(), sd[bn_key + ".running_var"].float()
    s = gamma / torch.sqrt(var + BN_EPS)
    return w * s.view(-1, 1, 1, 1), beta - mean * s


def _gate_interleave(w_t, H):
    """[K, 4H] (columns = gate-major i,f,g,o blocks of H units) -> [K, H, 4]: the 4 gates of unit j adjacent,
    so a lane fetches them with one 16-byte load."""
    K = w_t.shape[0]
    return w_t.reshape(K, 4, H).permute(0, 2, 1).contiguous()


# Beam decode: hoist the context half of the LSTMCell input product out of the step loop (csrc/attn_beam_mfma.hip, HOIST);
# MSOCR_BEAM_HOIST=0 keeps the reference's per-step formulation.
HOIST_CTX = os.environ.get("MSOCR_BEAM_HOIST", "1") != "0"

LAYER_SPEC = (("layer1", 1, 2), ("layer2", 2, 1), ("layer3", 5, 2), ("layer4", 3, 1))


class TrbaNet:
    def __init__(self, state_dict, num_classes, hidden=256, dtype=torch.float32, device="cuda", split=None):
        # hidden_size comes from the checkpoint's config.json (reference __init__.py:142-151, default 256).  256 / <= 256 tokens /
        # beam <= 8 run on the fast decoder kernels; other multiples of 64 up to 512, charsets up to 512 tokens and beams up to 16
        # (beam x hidden <= 4096) on the general one (csrc/attn_general.hip)
        if hidden % 64 or not 64 <= hidden <= 512:
            raise ValueError(f"hidden_size must be a multiple of 64 between 64 and 512 for the HIP recurrent / attention kernels, got {hidden}")
        if num_classes > 512:
            raise ValueError(f"charsets above 512 tokens are not supported by the decoder kernels, got {num_classes}")
        self.dtype, self.device = dtype, torch.device(device)
        self.V, self.Hd = num_classes, hidden
        dev, sd = self.device, state_dict
        P = {}
        self.cpad = 4 if dtype == torch.float32 else 8
        self.cin_pad = 16 if dtype == torch.float32 else 32
        w, b = _fold(sd, "cnn.conv0.0", "cnn.conv0.1")
        P["stem"] = (pack_stem_weight(w, self.cin_pad, self.cpad).to(dtype).to(dev), b.to(dev))
        w, b = _fold(sd, "cnn.conv0.3", "cnn.conv0.4")
        P["conv0b"] = (to_khwc(w, dtype, dev, split), b.to(dev))
        for lname, blocks, stride in LAYER_SPEC:
            for i in range(blocks):
                p = f"cnn.{lname}.{i}."
                for j in (1, 2):
                    w, b = _fold(sd, p + f"conv{j}", p + f"bn{j}")
                    P[f"{lname}.{i}.conv{j}"] = (to_khwc(w, dtype, dev, split), b.to(dev))
                P[f"{lname}.{i}.se"] = (sd[p + "se.fc.0.weight"].float().contiguous().to(dev),
                                        sd[p + "se.fc.2.weight"].float().contiguous().to(dev))
                if p + "downsample.0.weight" in sd:
                    w, b = _fold(sd, p + "downsample.0", p + "downsample.1")
                    P[f"{lname}.{i}.down"] = (to_khwc(w, dtype, dev, split), b.to(dev))
        w, b = _fold(sd, "cnn.conv_out.0", "cnn.conv_out.1")
        P["out0"] = (to_khwc(w, dtype, dev, split), b.to(dev))
        w, b = _fold(sd, "cnn.conv_out.3", "cnn.conv_out.4")
        P["out1"] = (to_khwc(w, dtype, dev, split), b.to(dev))
        self.P = P
        # ---- BiLSTM x2 (f32) ----
        H = hidden
        self.rnn = []
        for l in (0, 1):
            p = f"enc_rnn.{l}."
            w_ih = torch.cat([sd[p + "rnn.weight_ih_l0"], sd[p + "rnn.weight_ih_l0_reverse"]]).float()  # [2*4H, In]
            b = torch.cat([sd[p + "rnn.bias_ih_l0"] + sd[p + "rnn.bias_hh_l0"],
                           sd[p + "rnn.bias_ih_l0_reverse"] + sd[p + "rnn.bias_hh_l0_reverse"]]).float()
            whh_t = torch.stack([_gate_interleave(sd[p + "rnn.weight_hh_l0"].float().t(), H),
                                 _gate_interleave(sd[p + "rnn.weight_hh_l0_reverse"].float().t(), H)])  # [2][H(k)][H(j)][4]
            whh_p = None
            if (ops.SPLIT_BF16X3 if split is None else split) and dtype == torch.float32 and H == 256:
                # W_hh of both directions in the packed split form of the matrix-core recurrence (csrc/bilstm_mfma.hip)
                n = nat.lib().msocr_attn_pack_split_elems(4 * H)
                packed = torch.empty((2, n), dtype=torch.int16)
                wt = whh_t.contiguous()
                for d in (0, 1):
                    nat.check(nat.lib().msocr_attn_pack_split_host(wt[d].data_ptr(), 4 * H, 1, packed[d].data_ptr()), "attn_pack_split_host")
                whh_p = packed.to(dev)
            self.rnn.append({
                "w_ih": w_ih.view(8 * H, 1, 1, -1).contiguous().to(dev), "b": b.contiguous().to(dev),
                "whh_t": whh_t.contiguous().to(dev), "whh_p": whh_p,
                "lin_w": sd[p + "linear.weight"].float().view(H, 1, 1, 2 * H).contiguous().to(dev),
                "lin_b": sd[p + "linear.bias"].float().contiguous().to(dev),
            })
        # ---- attention decoder (f32, transposed for coalesced column reads) ----
        a = "attn.attention_cell."
        w_ih = sd[a + "rnn.weight_ih"].float()  # [4H, H + V]
        self.att = {
            "i2h_w": sd[a + "i2h.weight"].float().view(H, 1, 1, H).contiguous().to(dev),
            "h2h_wt": sd[a + "h2h.weight"].float().t().contiguous().to(dev),
            "h2h_b": sd[a + "h2h.bias"].float().contiguous().to(dev),
            "score_w": sd[a + "score.weight"].float().view(H).contiguous().to(dev),
            "wih_ctx_t": _gate_interleave(w_ih[:, :H].t(), H).to(dev),
            "wih_tok": _gate_interleave(w_ih[:, H:].t(), H).to(dev),
            "whh_t": _gate_interleave(sd[a + "rnn.weight_hh"].float().t(), H).to(dev),
            "b_gates": _gate_interleave((sd[a + "rnn.bias_ih"] + sd[a + "rnn.bias_hh"]).float().view(1, 4 * H), H).view(H, 4).contiguous().to(dev),
            "gen_wt": sd["attn.generator.weight"].float().t().contiguous().to(dev),
            "gen_b": sd["attn.generator.bias"].float().contiguous().to(dev),
        }
        # rows of rnn.weight_ih[:, :H] reordered unit-major (row j*4+g): the hoisted context product of the beam kernel
        # (msocr_attn_beam_hoisted) is one GEMM batch_H x this^T -> [B*T, H*4]
        self.att["wih_ctx_rows"] = ops.attach_split(w_ih[:, :H].reshape(4, H, H).permute(1, 0, 2).reshape(4 * H, 1, 1, H).contiguous().to(dev),
                                                    split)
        aw = nat.AttnWeights()
        for k in ("h2h_wt", "h2h_b", "score_w", "wih_ctx_t", "wih_tok", "whh_t", "b_gates", "gen_wt", "gen_b"):
            setattr(aw, k, self.att[k].data_ptr())
        self._aw = aw
        # split form of the three per-step matrices for the matrix-core beam kernel (precision "fp32"; "fp32-exact" keeps the f32 MFMA)
        self._asw = None
        if (ops.SPLIT_BF16X3 if split is None else split) and dtype == torch.float32 and H == 256 and self.V <= 256:
            asw = nat.AttnSplitWeights()
            for src, dst, n, gi in (("h2h_wt", "h2h_p", H, 0), ("whh_t", "whh_p", 4 * H, 1), ("gen_wt", "gen_p", self.V, 0)):
                wt = self.att[src].cpu().contiguous()
                packed = torch.empty((nat.lib().msocr_attn_pack_split_elems(n),), dtype=torch.int16)
                nat.check(nat.lib().msocr_attn_pack_split_host(wt.data_ptr(), n, gi, packed.data_ptr()), "attn_pack_split_host")
                self.att[dst] = packed.to(dev)
                setattr(asw, dst, self.att[dst].data_ptr())
            self._asw = asw

    # ------------------------------------------------------------------------------------- CNN
    def _se_block(self, x, lname, i, stride):
        P = self.P
        w1, b1 = P[f"{lname}.{i}.conv1"]
        w2, b2 = P[f"{lname}.{i}.conv2"]
        o = ops.conv2d(x, w1, b1, stride=(stride, stride), pad=(1, 1), relu=True)
        o = ops.conv2d(o, w2, b2, pad=(1, 1))
        key = f"{lname}.{i}.down"
        idt = ops.conv2d(x, *P[key], stride=(stride, stride)) if key in P else x
        s1, s2 = P[f"{lname}.{i}.se"]
        return ops.se_residual(o, idt, s1, s2)

    def cnn(self, canvases_u8):
        """canvases [B,h,w,3] u8 on device (already resized+padded) -> [B,Hf,T,512] NHWC (dtype)."""
        B, h, w, _ = canvases_u8.shape
        x = ops.normalize_u8(canvases_u8, 1, 1, h + 2, w + 4, 1, self.dtype, cpad=self.cpad)
        ws, bs = self.P["stem"]
        x = ops.conv2d(stem_view(x, self.cin_pad), ws, bs, (1, 1), (0, 0), True, out_hw=(h, w), alg_k=27)
        x = ops.conv2d(x, *self.P["conv0b"], pad=(1, 1), relu=True, pool2=True)  # conv0b + ReLU + MaxPool2d(2, 2)
        for lname, blocks, stride in LAYER_SPEC:
            for i in range(blocks):
                x = self._se_block(x, lname, i, stride if i == 0 else 1)
        x = ops.conv2d(x, *self.P["out0"], stride=(2, 1), pad=(0, 1), relu=True)
        return ops.conv2d(x, *self.P["out1"], relu=True)

    # ------------------------------------------------------------------------------------- encoder
    def _gemm(self, x2d, w, b):
        """x2d [M,K] f32 -> [M,N] via the exact-f32 MFMA implicit-GEMM (1x1 conv)."""
        M, K = x2d.shape
        return ops.conv2d(x2d.view(1, M, 1, K), w, b).view(M, w.shape[0])

    def encode(self, canvases_u8):
        """-> (batch_H [B,T,256] f32, proj_H [B,T,256] f32)."""
        f = self.cnn(canvases_u8)
        B, Hf, T, C = f.shape
        seq = ops.mean_over_h(f)  # [B,T,512] f32
        H = self.Hd
        for l in (0, 1):
            r = self.rnn[l]
            xproj = self._gemm(seq.view(B * T, -1), r["w_ih"], r["b"])  # [B*T, 2*4H] = [B][T][2][4H]
            hcat = ops.bilstm_recurrent(xproj, r["whh_t"], B, T, H, r["whh_p"])   # [B,T,2H]
            seq = self._gemm(hcat.view(B * T, 2 * H), r["lin_w"], r["lin_b"]).view(B, T, H)
        proj = self._gemm(seq.view(B * T, H), self.att["i2h_w"], None).view(B, T, H)
        return seq, proj

    # ------------------------------------------------------------------------------------- decoder
    def greedy(self, batch_H, proj_H, max_len, sos_id, eos_id, blank_id):
        B, T, H = batch_H.shape
        steps = max_len + 1
        logits = torch.empty((B, steps, self.V), dtype=torch.float32, device=self.device)
        ids = torch.empty((B, steps), dtype=torch.int32, device=self.device)
        blank = -1 if blank_id is None else blank_id
        # default: the matrix-core row-block kernel (32 crops per workgroup, split-operand products, hoisted context gates), like
        # the beam path; MSOCR_GREEDY_MFMA=0, precision="fp32-exact" and shapes outside the fast kernels: the VALU / general kernel
        fast = H == 256 and self.V <= 256 and T <= 48 and getattr(self, "_asw", None) is not None
        if fast and HOIST_CTX and os.environ.get("MSOCR_GREEDY_MFMA", "1") != "0":
            ctxg = self._gemm(batch_H.reshape(B * T, H), self.att["wih_ctx_rows"], None)
            nat.check(nat.lib().msocr_attn_greedy_hoisted(batch_H.data_ptr(), proj_H.data_ptr(), ctxg.data_ptr(), ctypes.byref(self._aw),
                                                          ctypes.byref(self._asw), B, T, H, self.V, steps, sos_id, eos_id, blank,
                                                          logits.data_ptr(), ids.data_ptr(), ops._stream()), "attn_greedy_hoisted")
        else:
            nat.check(nat.lib().msocr_attn_greedy(batch_H.data_ptr(), proj_H.data_ptr(), ctypes.byref(self._aw), B, T, H, self.V, steps,
                                                  sos_id, eos_id, blank, logits.data_ptr(), ids.data_ptr(), ops._stream()), "attn_greedy")
        return logits, ids

    def beam(self, batch_H, proj_H, max_len, beam_size, alpha, temperature, sos_id, eos_id, blank_id, chunks=None):
        """Returns (workspace, fin_step [B] i32, lp) for `beam_finalize`.  Runs all `max_len` steps, or — given
        chunks = (chunk_id [B] i32, chunk_size [nchunks] i32, chunk_state [2*nchunks] i32 zeros), all on the device — only
        as many as the reference's own loop would (it breaks once every beam of the chunk is finished, model.py:215)."""
        B, T, H = batch_H.shape
        steps = max_len
        nbytes = nat.lib().msocr_attn_beam_workspace_bytes(B, steps, beam_size, self.V)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
        fin = torch.empty((B,), dtype=torch.int32, device=self.device)
        lp = None
        if alpha > 0:  # model.py:160, evaluated in Python double then applied in f32; cached: an H2D copy here would make
            #            the host wait for every encoder kernel already queued on this stream
            key = (float(alpha), steps)
            if not hasattr(self, "_lp_cache"):
                self._lp_cache = {}
            if key not in self._lp_cache:
                self._lp_cache[key] = torch.tensor([((5.0 + (t + 1)) ** alpha) / (6.0 ** alpha) for t in range(steps)],
                                                   dtype=torch.float32).to(self.device)
                torch.cuda.synchronize()
            lp = self._lp_cache[key]
        if not 1 <= beam_size <= 16 or (beam_size * H > 4096 and not (H == 256 and beam_size <= 8)):
            raise ValueError(f"beam_size {beam_size} with hidden_size {H}: the decoder kernels take beam_size <= 16 and beam_size x hidden_size <= 4096")
        fast = H == 256 and self.V <= 256 and T <= 48 and beam_size <= 8  # the matrix-core kernel's shapes
        hoist = HOIST_CTX and fast and os.environ.get("MSOCR_BEAM_MFMA", "1") != "0"
        if hoist:  # W_ih[:, :H] batch_H_t for every frame, once per call instead of W_ih[:, :H] ctx in every step
            ctxg = self._gemm(batch_H.reshape(B * T, H), self.att["wih_ctx_rows"], None)
        e = ops._prof_begin()
        tail = (B, T, H, self.V, steps, beam_size, lp.data_ptr() if lp is not None else None, float(temperature), sos_id, eos_id,
                -1 if blank_id is None else blank_id, fin.data_ptr(), ws.data_ptr(),
                chunks[0].data_ptr() if chunks else None, chunks[1].data_ptr() if chunks else None,
                chunks[2].data_ptr() if chunks else None, ops._stream())
        if hoist:
            asw = self._asw if os.environ.get("MSOCR_BEAM_SPLIT", "1") != "0" else None
            nat.check(nat.lib().msocr_attn_beam_hoisted(batch_H.data_ptr(), proj_H.data_ptr(), ctxg.data_ptr(), ctypes.byref(self._aw),
                                                        ctypes.byref(asw) if asw is not None else None, *tail), "attn_beam_hoisted")
        else:
            nat.check(nat.lib().msocr_attn_beam(batch_H.data_ptr(), proj_H.data_ptr(), ctypes.byref(self._aw), *tail), "attn_beam")
        # SURVEY.md 8d, per decode step: proj_H + batch_H (shared by the beams) + LSTMCell W_ih (ctx part + one-hot rows), W_hh,
        # generator, h2h + per row (h, c state + logits); the launch runs up to `steps` of them (chunk-level early exit)
        V, R = self.V, B * beam_size
        per_step = 4.0 * (2 * B * T * H + 4 * H * (H + V) + 4 * H * H + H * V + H * H + R * (4 * H + V))
        ops._prof_end(e, "attn_beam", (per_step, steps), (B, T, beam_size))
        return ws, fin, lp

    def beam_finalize(self, ws, B, steps, beam_size, trun_dev):
        logits = torch.empty((B, steps, self.V), dtype=torch.float32, device=self.device)
        ids = torch.empty((B, steps), dtype=torch.int32, device=self.device)
        nat.check(nat.lib().msocr_attn_beam_finalize(ws.data_ptr(), B, self.V, steps, beam_size, trun_dev.data_ptr(), logits.data_ptr(),
                                                     ids.data_ptr(), ops._stream()), "attn_beam_finalize")
        return logits, ids


def trba_cnn_macs(h, w):
    """Algorithmic MACs of SE-ResNet31 for one h x w crop (2.848 G at 32x100)."""
    macs = h * w * 64 * 27 + h * w * 128 * 64 * 9
    hh, ww, cin = h // 2, w // 2, 128
    for (lname, blocks, stride), planes in zip(LAYER_SPEC, (256, 256, 512, 512)):
        for i in range(blocks):
            s = stride if i == 0 else 1
            h2, w2 = (hh + 2 - 3) // s + 1, (ww + 2 - 3) // s + 1
            macs += h2 * w2 * planes * cin * 9 + h2 * w2 * planes * planes * 9
            if s != 1 or cin != planes:
                macs += h2 * w2 * planes * cin
            macs += 2 * planes * (planes // 16)
            hh, ww, cin = h2, w2, planes
    h3, w3 = (hh - 2) // 2 + 1, (ww + 2 - 2) + 1
    macs += h3 * w3 * 512 * 512 * 4
    macs += (h3 - 1) * (w3 - 1) * 512 * 512 * 4
    return macs
