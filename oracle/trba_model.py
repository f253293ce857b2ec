"""CPU fp32 restatement of the reference TRBA recogniser (TEST INFRASTRUCTURE).

Reference (under /root/reference/src/manuscript/recognizers/_trba/):
  SELayer / SEBasicBlock / SEResNet31   model/seresnet31.py:5-187
  BidirectionalLSTM                     model/model.py:9-21
  AttentionCell.forward                 model/model.py:34-46
  Attention._greedy_decode              model/model.py:227-259
  Attention._beam_decode                model/model.py:92-225
  TRBAModel.encode / forward            model/model.py:387-416
  TRBA.predict post (log_softmax, decode_tokens, confidence)  __init__.py:413-432
  load_charset / decode_tokens          data/transforms.py:39-59,196-206

Module names and construction order match the reference so that (a) its
state_dict keys load with strict=True and (b) `torch.manual_seed(s)` followed by
construction yields bit-identical parameters (checked when goldens are made).
Pinned by tests/golden/trba_*.npz, generated from the reference files.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- CNN
class SELayer(nn.Module):
    def __init__(self, channel, reduction=16):
        super().__init__()
        self.fc = nn.Sequential(
            nn.Linear(channel, channel // reduction, bias=False),
            nn.ReLU(inplace=True),
            nn.Linear(channel // reduction, channel, bias=False),
            nn.Sigmoid(),
        )

    def forward(self, x):
        b, c = x.shape[:2]
        gate = self.fc(F.adaptive_avg_pool2d(x, 1).view(b, c))
        return x * gate.view(b, c, 1, 1).expand_as(x)


class SEBasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, downsample=None, reduction=16):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.se = SELayer(planes, reduction)
        self.downsample = downsample

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.se(self.bn2(self.conv2(out)))
        identity = x if self.downsample is None else self.downsample(x)
        return F.relu(out + identity)


class SEResNet31(nn.Module):
    def __init__(self, in_channels=3, out_channels=512, reduction=16):
        super().__init__()
        self.conv0 = nn.Sequential(
            nn.Conv2d(in_channels, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64), nn.ReLU(True),
            nn.Conv2d(64, 128, 3, 1, 1, bias=False), nn.BatchNorm2d(128), nn.ReLU(True),
            nn.MaxPool2d(2, 2),
        )
        self.layer1 = self._make_layer(128, 256, 1, 2, reduction)
        self.layer2 = self._make_layer(256, 256, 2, 1, reduction)
        self.layer3 = self._make_layer(256, 512, 5, 2, reduction)
        self.layer4 = self._make_layer(512, 512, 3, 1, reduction)
        self.conv_out = nn.Sequential(
            nn.Conv2d(512, out_channels, 2, stride=(2, 1), padding=(0, 1), bias=False),
            nn.BatchNorm2d(out_channels), nn.ReLU(True),
            nn.Conv2d(out_channels, out_channels, 2, stride=1, padding=0, bias=False),
            nn.BatchNorm2d(out_channels), nn.ReLU(True),
        )
        self.out_channels = out_channels

    @staticmethod
    def _make_layer(inplanes, planes, blocks, stride, reduction):
        down = None
        if stride != 1 or inplanes != planes:
            down = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride=stride, bias=False), nn.BatchNorm2d(planes))
        seq = [SEBasicBlock(inplanes, planes, stride, down, reduction)]
        for _ in range(1, blocks):
            seq.append(SEBasicBlock(planes, planes, reduction=reduction))
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self.conv0(x)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = layer(x)
        return self.conv_out(x)


# ----------------------------------------------------------------------------- encoder RNN
class BidirectionalLSTM(nn.Module):
    def __init__(self, input_size, hidden_size, output_size):
        super().__init__()
        self.rnn = nn.LSTM(input_size, hidden_size, bidirectional=True, batch_first=True)
        self.linear = nn.Linear(hidden_size * 2, output_size)

    def forward(self, x):
        h, _ = self.rnn(x)
        return self.linear(h)


# ----------------------------------------------------------------------------- attention decoder
class AttentionCell(nn.Module):
    def __init__(self, input_size, hidden_size, num_embeddings):
        super().__init__()
        self.i2h = nn.Linear(input_size, hidden_size, bias=False)
        self.h2h = nn.Linear(hidden_size, hidden_size)
        self.score = nn.Linear(hidden_size, 1, bias=False)
        self.rnn = nn.LSTMCell(input_size + num_embeddings, hidden_size)

    def forward(self, prev_hidden, batch_H, char_onehots):
        # model.py:36-45 — i2h(batch_H) is recomputed every step in the reference
        e = self.score(torch.tanh(self.i2h(batch_H) + self.h2h(prev_hidden[0]).unsqueeze(1)))
        alpha = F.softmax(e, dim=1)
        context = torch.bmm(alpha.transpose(1, 2), batch_H).squeeze(1)
        return self.rnn(torch.cat([context, char_onehots], 1), prev_hidden)


class Attention(nn.Module):
    def __init__(self, input_size, hidden_size, num_classes, sos_id, eos_id, pad_id, blank_id=None):
        super().__init__()
        self.attention_cell = AttentionCell(input_size, hidden_size, num_classes)
        self.hidden_size, self.num_classes = hidden_size, num_classes
        self.sos_id, self.eos_id, self.pad_id, self.blank_id = sos_id, eos_id, pad_id, blank_id
        self.generator = nn.Linear(hidden_size, num_classes)

    def _onehot(self, tok):
        oh = torch.zeros(tok.size(0), self.num_classes)
        oh.scatter_(1, tok.unsqueeze(1), 1.0)
        return oh

    def _mask(self, logits):
        if self.blank_id is not None:
            logits[..., self.blank_id] = -1e4
        return logits

    @torch.no_grad()
    def greedy(self, batch_H, max_len=25, diag=None):
        """model.py:227-259. Returns (logits B x T_run x V, ids B x T_run).
        diag (checker only, no effect on the arithmetic): dict that receives "top2_margin" [B, T_run] = gap between the
        largest and second largest logit of every step, i.e. the margin of every arg-max decision of this decode."""
        B = batch_H.size(0)
        hid = (torch.zeros(B, self.hidden_size), torch.zeros(B, self.hidden_size))
        tok = torch.full((B,), self.sos_id, dtype=torch.long)
        logits_all, ids_all = [], []
        for _ in range(max_len + 1):
            hid = self.attention_cell(hid, batch_H, self._onehot(tok))
            logits = self._mask(self.generator(hid[0]))
            tok = logits.argmax(1)
            logits_all.append(logits)
            ids_all.append(tok)
            if bool((tok == self.eos_id).all()):  # stops only if EVERY row emits EOS at this step
                break
        if diag is not None:
            t2 = torch.stack(logits_all, 1).topk(2, dim=-1).values
            diag["top2_margin"] = (t2[..., 0] - t2[..., 1]).numpy()
        return torch.stack(logits_all, 1), torch.stack(ids_all, 1)

    @torch.no_grad()
    def beam(self, batch_H, max_len=25, beam_size=5, alpha=0.9, temperature=1.7, diag=None):
        """model.py:92-225. Returns (temperature-scaled logits of the best beam's
        path B x T_run x V, tokens B x T_run without SOS).
        diag (checker only, no effect on the arithmetic): dict that receives the margins of every decision this search
        takes — "boundary_gap" [B, T_run]: K-th minus (K+1)-th candidate of each step's top-k (what decides which
        hypotheses survive), "beam_scores" [B, K] / "beam_tokens" [B, K, T_run]: the final hypotheses the arg-max picks from."""
        B, K, H, V = batch_H.size(0), beam_size, self.hidden_size, self.num_classes
        toks = torch.full((B, K, 1), self.sos_id, dtype=torch.long)
        score = torch.full((B, K), float("-inf"))
        score[:, 0] = 0.0
        bh, bc = torch.zeros(B, K, H), torch.zeros(B, K, H)
        done = torch.zeros(B, K, dtype=torch.bool)
        trace = None
        rep_H = batch_H.repeat_interleave(K, dim=0)
        for t in range(max_len):
            h1, c1 = self.attention_cell((bh.reshape(B * K, H), bc.reshape(B * K, H)), rep_H,
                                         self._onehot(toks[:, :, -1].reshape(B * K)))
            logits = self._mask(self.generator(h1))
            if temperature != 1.0:
                logits = logits / max(temperature, 1e-6)
            logp = F.log_softmax(logits, dim=-1).view(B, K, V)
            if bool(done.any()):
                m = done.unsqueeze(-1)
                logp = torch.where(m.expand_as(logp), torch.full_like(logp, float("-inf")), logp)
                logp[..., self.eos_id] = torch.where(done, torch.zeros_like(logp[..., self.eos_id]),
                                                     logp[..., self.eos_id])
            total = score.unsqueeze(-1) + logp
            if alpha > 0:
                lp = ((5.0 + (t + 1)) ** alpha) / (6.0 ** alpha)
                cand = total / lp
            else:
                cand = total
            top, idx = torch.topk(cand.view(B, -1), k=K, dim=-1)
            if diag is not None:
                tk1 = torch.topk(cand.view(B, -1), k=K + 1, dim=-1).values
                diag.setdefault("boundary_gap", []).append((tk1[:, K - 1] - tk1[:, K]).numpy())
            src = idx // V
            nxt = (idx % V).clamp(0, V - 1)
            gat = lambda x: x.gather(1, src.unsqueeze(-1).expand(-1, -1, x.size(-1)))
            bh, bc = gat(h1.view(B, K, H)), gat(c1.view(B, K, H))
            toks = torch.cat([gat(toks), nxt.unsqueeze(-1)], dim=-1)
            score = top * lp if alpha > 0 else top  # f32 round trip, not the exact sum
            done = done.gather(1, src) | (nxt == self.eos_id)
            sel = gat(logits.view(B, K, V)).unsqueeze(2)
            if trace is None:
                trace = sel
            else:
                trace = trace.gather(1, src[:, :, None, None].expand(-1, -1, trace.size(2), V))
                trace = torch.cat([trace, sel], dim=2)
            if bool(done.all()):
                break
        best = score.argmax(-1)
        ar = torch.arange(B)
        if diag is not None:
            import numpy as np
            diag["boundary_gap"] = np.stack(diag["boundary_gap"], 1)
            diag["beam_scores"], diag["beam_tokens"] = score.numpy().copy(), toks[:, :, 1:].numpy().copy()
        return trace[ar, best], toks[ar, best][:, 1:]


class TRBANet(nn.Module):
    def __init__(self, num_classes, hidden_size=256, sos_id=1, eos_id=2, pad_id=0, blank_id=None):
        super().__init__()
        self.num_classes, self.hidden_size = num_classes, hidden_size
        self.cnn = SEResNet31(3, 512)
        self.enc_rnn = nn.Sequential(
            BidirectionalLSTM(512, hidden_size, hidden_size),
            BidirectionalLSTM(hidden_size, hidden_size, hidden_size),
        )
        self.attn = Attention(hidden_size, hidden_size, num_classes, sos_id, eos_id, pad_id, blank_id)

    def cnn_features(self, x):
        """B x 3 x h x w -> B x T x 512 (mean over H, model.py:388-390)."""
        f = self.cnn(x)
        return F.adaptive_avg_pool2d(f, (1, None)).squeeze(2).permute(0, 2, 1)

    def encode(self, x):
        return self.enc_rnn(self.cnn_features(x))

    def forward(self, x, max_len=25, mode="greedy", beam_size=5, alpha=0.6, temperature=1.0, diag=None):
        enc = self.encode(x)
        if mode == "greedy":
            return self.attn.greedy(enc, max_len, diag)
        if mode == "beam":
            return self.attn.beam(enc, max_len, beam_size, alpha, temperature, diag)
        raise ValueError(f"Unknown decode mode: {mode}")


# ----------------------------------------------------------------------------- host post
def load_charset(path):
    itos = []
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            tok = line.rstrip("\n")
            if tok != "":
                itos.append(tok)
    return itos, {s: i for i, s in enumerate(itos)}


def decode_tokens(ids, itos, pad_id, eos_id, blank_id=None):
    out = []
    for t in ids:
        t = int(t)
        if t == eos_id:
            break
        if t == pad_id or (blank_id is not None and t == blank_id):
            continue
        out.append(itos[t])
    return "".join(out)


def texts_and_confidences(logits, ids, itos, pad_id, eos_id, blank_id=None):
    """__init__.py:413-432: confidence = mean over ALL T_run positions of exp(logp[id])."""
    logp = F.log_softmax(logits, dim=-1)
    res = []
    for j in range(ids.size(0)):
        row = ids[j].tolist()
        text = decode_tokens(row, itos, pad_id, eos_id, blank_id)
        conf = logp[j, torch.arange(len(row)), row].exp().mean().item() if row else 0.0
        res.append({"text": text, "confidence": conf, "logits0": logits[j, 0].numpy().copy()})
    return res


def first_token_margin(logits0, itos, eos_id, text_a, text_b):
    """Checker helper for page-scale text comparisons (test infrastructure, not reference behaviour): the gap, in THIS
    CPU path's own first-step logits, between the first tokens of two decodings of one crop (EOS for an empty text).
    Two f32 implementations may legitimately disagree on an arg-max whose margin is at rounding-noise level; with the
    planted decoder of synth.trba_state_dict_confident every later character follows from the first one."""
    stoi = {s: i for i, s in enumerate(itos)}
    ta = stoi[text_a[0]] if text_a else eos_id
    tb = stoi[text_b[0]] if text_b else eos_id
    return abs(float(logits0[ta]) - float(logits0[tb]))
