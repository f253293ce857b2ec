"""Deterministic synthetic manuscript pages and EAST maps.

There are no trained weights offline, so random-weight networks produce
unusable score maps.  Benchmarks and parity tests therefore (a) run the real
network on synthetic pages for the convolution stages and (b) exercise the
data-dependent stages (decode, LANMS, crop, recognise) on *injected* maps
generated here from a ragged "lines of words" layout, following the geometry
semantics of the reference's training target
(/root/reference/src/manuscript/detectors/_east/dataset.py:180-199: score 1
inside the quad shrunk by 0.3, geo = offsets from the pixel to the shrunk
quad's 4 corners in map pixels).
"""
import numpy as np


def synth_layout(seed, height, width, line_pitch=36, word_h=28, margin=40):
    """Ragged word rectangles (x0, y0, x1, y1) in page pixels."""
    rng = np.random.default_rng(seed)
    rects = []
    y = margin
    while y + word_h <= height - margin:
        x = margin + rng.uniform(0, 40)
        while True:
            w = rng.uniform(70, 190)
            if x + w > width - margin:
                break
            yj = y + rng.uniform(-2, 2)
            rects.append((x, yj, x + w, yj + word_h))
            x += w + rng.uniform(18, 40)
        y += line_pitch
    return np.asarray(rects, dtype=np.float64).reshape(-1, 4)


def synth_page(seed, height, width, **layout_kw):
    """u8 RGB page: parchment clip(N(205,12)) with dark word rectangles clip(N(60,25))."""
    rng = np.random.default_rng(seed + 7919)
    page = np.clip(rng.normal(205, 12, size=(height, width, 3)), 0, 255).astype(np.uint8)
    rects = synth_layout(seed, height, width, **layout_kw)
    for x0, y0, x1, y1 in rects:
        a, b, c, d = int(x0), int(y0), int(x1), int(y1)
        page[b:d, a:c] = np.clip(rng.normal(60, 25, size=(d - b, c - a, 3)), 0, 255).astype(np.uint8)
    return page, rects


def synth_maps(rects, page_hw, map_hw, seed, noise=0.05):
    """Injected score (mh,mw) f32 and geo (mh,mw,8) f32 maps for word rectangles."""
    rng = np.random.default_rng(seed + 104729)
    mh, mw = map_hw
    sy, sx = mh / page_hw[0], mw / page_hw[1]
    # background: low, tie-free scores (a sigmoid map never holds exact zeros; exact ties make the
    # reference's unstable argsorts implementation-defined)
    score = (0.2 * rng.random((mh, mw))).astype(np.float32)
    geo = rng.normal(0, 1, size=(mh, mw, 8)).astype(np.float32)
    for x0, y0, x1, y1 in rects:
        a, b, c, d = x0 * sx, y0 * sy, x1 * sx, y1 * sy
        s = 0.3 * min(c - a, d - b)
        a, b, c, d = a + s, b + s, c - s, d - s  # shrunk quad in map px
        if c <= a or d <= b:
            continue
        ys = np.arange(int(np.ceil(b)), int(np.floor(d)) + 1)
        xs = np.arange(int(np.ceil(a)), int(np.floor(c)) + 1)
        ys, xs = ys[(ys >= 0) & (ys < mh)], xs[(xs >= 0) & (xs < mw)]
        if len(ys) == 0 or len(xs) == 0:
            continue
        corners = [(a, b), (c, b), (c, d), (a, d)]  # TL, TR, BR, BL
        # geometry is valid on the text region AND a 2-px ring around it (a trained EAST regresses sensible
        # offsets next to text too); only the score separates text from background.  Without the ring the
        # quantised cell centres (utils.py:349-356) that fall just outside a word would decode noise quads.
        gy = np.arange(max(ys[0] - 2, 0), min(ys[-1] + 3, mh))
        gx = np.arange(max(xs[0] - 2, 0), min(xs[-1] + 3, mw))
        gyy, gxx = np.meshgrid(gy, gx, indexing="ij")
        for i, (vx, vy) in enumerate(corners):
            geo[gyy, gxx, 2 * i] = (vx - gxx + rng.normal(0, noise, gyy.shape)).astype(np.float32)
            geo[gyy, gxx, 2 * i + 1] = (vy - gyy + rng.normal(0, noise, gyy.shape)).astype(np.float32)
        yy, xx = np.meshgrid(ys, xs, indexing="ij")
        score[yy, xx] = (0.9 + 0.05 * rng.random(yy.shape)).astype(np.float32)
    return score, geo


def synth_crops(seed, n, h=32, w=100):
    """n u8 crops h x w x 3: parchment with 3-9 dark strokes."""
    rng = np.random.default_rng(seed)
    crops = np.clip(rng.normal(205, 12, size=(n, h, w, 3)), 0, 255).astype(np.uint8)
    for i in range(n):
        for _ in range(int(rng.integers(3, 10))):
            x = int(rng.integers(2, w - 6))
            ww = int(rng.integers(2, 6))
            y0 = int(rng.integers(2, h // 2))
            y1 = int(rng.integers(h // 2, h - 2))
            crops[i, y0:y1, x:x + ww] = np.clip(rng.normal(60, 25, size=(y1 - y0, ww, 3)), 0, 255).astype(np.uint8)
    return crops


# ------------------------------------------------------------------------------------------------ synthetic weights
# No trained checkpoint exists offline.  These generators emit seeded state_dicts in the REFERENCE key
# layout (torchvision ResNet-50 names under `backbone.extractor.`, `decoder.block*`, `output_head.*`; `cnn.*`,
# `enc_rnn.*`, `attn.*` for TRBA) from shape tables alone, so benchmarks and tests can build the product networks and
# the oracle networks from the same tensors.
def _bn(sd, g, prefix, c, gamma_scale=1.0):
    import torch
    sd[prefix + ".weight"] = (0.8 + 0.4 * torch.rand(c, generator=g)) * gamma_scale
    sd[prefix + ".bias"] = 0.1 * torch.randn(c, generator=g)
    sd[prefix + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
    sd[prefix + ".running_var"] = 0.8 + 0.4 * torch.rand(c, generator=g)
    sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def _conv(sd, g, key, cout, cin, kh, kw, bias=False, gain=1.0):
    import torch
    sd[key + ".weight"] = torch.randn(cout, cin, kh, kw, generator=g) * (2.0 / (cin * kh * kw)) ** 0.5 * gain  # He normal
    if bias:
        sd[key + ".bias"] = 0.1 * torch.randn(cout, generator=g)


def east_state_dict(seed=20260128):
    """He-normal convs, near-identity BatchNorm statistics, damped last BN of every Bottleneck (16 residual adds),
    zero-mean score weights with bias -1.5 (~20 % of pixels above the 0.6 threshold), geometry O(10 px)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    sd = {}
    p = "backbone.extractor."
    _conv(sd, g, p + "conv1", 64, 3, 7, 7)
    _bn(sd, g, p + "bn1", 64)
    inplanes = 64
    for li, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), start=1):
        for b in range(blocks):
            q = f"{p}layer{li}.{b}."
            _conv(sd, g, q + "conv1", planes, inplanes, 1, 1)
            _bn(sd, g, q + "bn1", planes)
            _conv(sd, g, q + "conv2", planes, planes, 3, 3)
            _bn(sd, g, q + "bn2", planes)
            _conv(sd, g, q + "conv3", planes * 4, planes, 1, 1)
            _bn(sd, g, q + "bn3", planes * 4, gamma_scale=0.35)
            if b == 0:
                _conv(sd, g, q + "downsample.0", planes * 4, inplanes, 1, 1)
                _bn(sd, g, q + "downsample.1", planes * 4)
            inplanes = planes * 4
    for k, (cin, mid, cout) in enumerate(((2048, 512, 512), (1536, 256, 256), (768, 128, 128), (384, 64, 32)), start=1):
        q = f"decoder.block{k}."
        _conv(sd, g, q + "conv1x1.0", mid, cin, 1, 1, bias=True)
        _bn(sd, g, q + "conv1x1.1", mid)
        _conv(sd, g, q + "conv3x3.0", cout, mid, 3, 3, bias=True)
        _bn(sd, g, q + "conv3x3.1", cout)
    w = torch.randn(1, 32, 1, 1, generator=g) * 0.25
    sd["output_head.score_map.weight"] = w - w.mean()
    sd["output_head.score_map.bias"] = torch.full((1,), -1.5)
    sd["output_head.geo_map.weight"] = torch.randn(8, 32, 1, 1, generator=g) * 1.5
    sd["output_head.geo_map.bias"] = 0.1 * torch.randn(8, generator=g)
    return sd


def trba_state_dict(num_classes=194, hidden=256, seed=20260128, rnn_scale=6.0, gen_scale=8.0, eos_period=6):
    """He-normal CNN with near-identity BatchNorm; recurrent / attention / generator weights drawn U(-1/sqrt(H), 1/sqrt(H))
    (PyTorch's default range) and scaled up so the decode depends on the input; and an "EOS trigger": emitting a token
    with id % eos_period == 4 drives the decoder LSTM state along a fixed +-1 direction u that the generator's EOS row
    reads, so sequences end at varied steps (exercises finished-beam masking and the early-break emulation)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    sd = {}
    H, V = hidden, num_classes

    def uni(*shape, k=1.0 / hidden ** 0.5, scale=1.0):
        return (torch.rand(*shape, generator=g) * 2 - 1) * k * scale

    _conv(sd, g, "cnn.conv0.0", 64, 3, 3, 3)
    _bn(sd, g, "cnn.conv0.1", 64)
    _conv(sd, g, "cnn.conv0.3", 128, 64, 3, 3)
    _bn(sd, g, "cnn.conv0.4", 128)
    inplanes = 128
    for li, (planes, blocks, stride) in enumerate(((256, 1, 2), (256, 2, 1), (512, 5, 2), (512, 3, 1)), start=1):
        for b in range(blocks):
            q = f"cnn.layer{li}.{b}."
            _conv(sd, g, q + "conv1", planes, inplanes, 3, 3)
            _bn(sd, g, q + "bn1", planes)
            _conv(sd, g, q + "conv2", planes, planes, 3, 3)
            _bn(sd, g, q + "bn2", planes)
            sd[q + "se.fc.0.weight"] = uni(planes // 16, planes, k=1.0 / planes ** 0.5)
            sd[q + "se.fc.2.weight"] = uni(planes, planes // 16, k=1.0 / (planes // 16) ** 0.5)
            if b == 0 and (stride != 1 or inplanes != planes):
                _conv(sd, g, q + "downsample.0", planes, inplanes, 1, 1)
                _bn(sd, g, q + "downsample.1", planes)
            inplanes = planes
    _conv(sd, g, "cnn.conv_out.0", 512, 512, 2, 2)
    _bn(sd, g, "cnn.conv_out.1", 512)
    _conv(sd, g, "cnn.conv_out.3", 512, 512, 2, 2)
    _bn(sd, g, "cnn.conv_out.4", 512)
    for l, cin in ((0, 512), (1, H)):
        q = f"enc_rnn.{l}."
        for suf in ("", "_reverse"):
            sd[q + "rnn.weight_ih_l0" + suf] = uni(4 * H, cin, scale=rnn_scale)
            sd[q + "rnn.weight_hh_l0" + suf] = uni(4 * H, H, scale=rnn_scale)
            sd[q + "rnn.bias_ih_l0" + suf] = uni(4 * H, scale=rnn_scale)
            sd[q + "rnn.bias_hh_l0" + suf] = uni(4 * H, scale=rnn_scale)
        sd[q + "linear.weight"] = uni(H, 2 * H, k=1.0 / (2 * H) ** 0.5, scale=rnn_scale)
        sd[q + "linear.bias"] = uni(H, k=1.0 / (2 * H) ** 0.5, scale=rnn_scale)
    a = "attn.attention_cell."
    sd[a + "i2h.weight"] = uni(H, H, scale=rnn_scale)
    sd[a + "h2h.weight"] = uni(H, H, scale=rnn_scale)
    sd[a + "h2h.bias"] = uni(H, scale=rnn_scale)
    sd[a + "score.weight"] = uni(1, H, scale=rnn_scale)
    sd[a + "rnn.weight_ih"] = uni(4 * H, H + V, scale=rnn_scale)
    sd[a + "rnn.weight_hh"] = uni(4 * H, H, scale=rnn_scale)
    sd[a + "rnn.bias_ih"] = uni(4 * H, scale=rnn_scale)
    sd[a + "rnn.bias_hh"] = uni(4 * H, scale=rnn_scale)
    sd["attn.generator.weight"] = uni(V, H, scale=gen_scale)
    sd["attn.generator.bias"] = uni(V)
    if eos_period:
        u = torch.where(torch.rand(H, generator=g) < 0.5, -1.0, 1.0)
        w_ih = sd[a + "rnn.weight_ih"]  # [4H, H + V], gate order i,f,g,o
        for t in range(4, V):
            if t % eos_period == 4:
                col = H + t
                w_ih[0:H, col] += 6.0
                w_ih[2 * H:3 * H, col] += 6.0 * u
                w_ih[3 * H:4 * H, col] += 6.0
        sd["attn.generator.weight"][2] = u * (10.0 / (0.7 * H)) + 0.1 * sd["attn.generator.weight"][2]
    return sd


def trba_state_dict_confident(num_classes=194, hidden=256, seed=20260128, chain=9, ctx_gain=6.0, tok_gain=10.0, gen_gain=40.0):
    """`trba_state_dict` with a PLANTED decoder whose decisions carry large margins, as a trained checkpoint's do.

    A decoder with random weights decides every character by an arg-max over near-Gaussian logits, so ~1e-5 of
    rounding-order noise flips a few words per thousand in ANY two f32 implementations (CPU vs GPU, or CPU vs itself under
    a 1e-5 perturbation).  Here only the FIRST character is read off the image (sign pattern of a random projection of the
    attention context, matched against +-1 code words in the generator); every later character follows a fixed successor
    table planted in the one-hot columns of the LSTMCell (forget gate shut, input/output gates open), ending in EOS after
    1..`chain` steps.  CNN, BiLSTMs and the attention scorer stay random, all arithmetic stays dense, and the kernels do
    exactly the same work; what changes is that text equality with the CPU reference is meaningful at page scale."""
    import torch
    sd = trba_state_dict(num_classes, hidden, seed=seed, eos_period=0)
    g = torch.Generator().manual_seed(seed + 7)
    H, V = hidden, num_classes
    code = torch.where(torch.rand(V, H, generator=g) < 0.5, -1.0, 1.0)
    nxt = list(range(V))
    for k in range(3, V):  # tokens 3..V-1 form chains of length `chain`; the last of a chain is followed by EOS (2)
        nxt[k] = 2 if (k - 3) % chain == chain - 1 or k == V - 1 else k + 1
    nxt[2] = 2  # EOS repeats (rows keep running until the whole chunk is done, model.py:215,254)
    a = "attn.attention_cell."
    small = lambda *shape: (torch.rand(*shape, generator=g) * 2 - 1) * (0.5 / H ** 0.5)
    w_ih, w_hh = small(4 * H, H + V), small(4 * H, H)
    b_ih, b_hh = small(4 * H), torch.zeros(4 * H)
    b_ih[0:H] += 8.0          # input gate open
    b_ih[H:2 * H] -= 8.0      # forget gate shut: c = i * g
    b_ih[3 * H:4 * H] += 8.0  # output gate open
    w_ih[2 * H:3 * H, :H] = torch.randn(H, H, generator=g) * (ctx_gain / H ** 0.5)  # context -> g (first character)
    for k in range(2, V):
        w_ih[2 * H:3 * H, H + k] = tok_gain * code[nxt[k]]
    sd[a + "rnn.weight_ih"], sd[a + "rnn.weight_hh"] = w_ih, w_hh
    sd[a + "rnn.bias_ih"], sd[a + "rnn.bias_hh"] = b_ih, b_hh
    sd["attn.generator.weight"] = code * (gen_gain / H) + small(V, H) * 0.1
    gb = small(V)
    gb[0] = gb[1] = -50.0  # PAD / SOS are never emitted
    sd["attn.generator.bias"] = gb
    return sd
