"""Device-side EAST network: ResNet-50 trunk + feature-merging decoder + heads on HIP.

Host logic only (weight folding/packing, buffer plumbing, launch order); every FLOP is
in libmsocr.so.  Mirrors the reference forward
(/root/reference/src/manuscript/detectors/_east/east.py:135-139) with:
  * BatchNorm folded into conv weight/bias at load time (eval mode: exact algebra);
  * NHWC activations, weights [Cout][KH][KW][Cin];
  * the 7x7/2 stem run as a 7-tap (kh) conv over 32-element rows of a zero-bordered
    C=4 canvas (kw and c are contiguous in NHWC), so it uses the same MFMA kernel;
  * torch.cat([upsample(h), f]) fused away: layer taps write straight into the
    right channel slice of the concat buffer and the x2 upsample fills the left one.
State-dict keys are the reference's (`backbone.extractor.*`, `decoder.block*`,
`output_head.*`), loaded non-strictly like east.py:130-133.
"""
import os

import torch

from ... import ops

LAYERS = (("layer1", 64, 3, 1), ("layer2", 128, 4, 2), ("layer3", 256, 6, 2), ("layer4", 512, 3, 2))
BN_EPS = 1e-5


def expected_state_shapes():
    """Key -> shape of the inference model's state_dict (reference EAST(backbone_name="resnet50"), east.py:108-127:
    torchvision ResNet-50 trunk without fc under `backbone.extractor.`, FeatureMergingBranchResNet, OutputHead)."""
    shapes = {}

    def conv(k, co, ci, kh, kw, bias=False):
        shapes[k + ".weight"] = (co, ci, kh, kw)
        if bias:
            shapes[k + ".bias"] = (co,)

    def bn(k, c):
        for n in ("weight", "bias", "running_mean", "running_var"):
            shapes[f"{k}.{n}"] = (c,)

    p = "backbone.extractor."
    conv(p + "conv1", 64, 3, 7, 7), bn(p + "bn1", 64)
    inplanes = 64
    for lname, planes, blocks, _ in LAYERS:
        for b in range(blocks):
            q = f"{p}{lname}.{b}."
            conv(q + "conv1", planes, inplanes, 1, 1), bn(q + "bn1", planes)
            conv(q + "conv2", planes, planes, 3, 3), bn(q + "bn2", planes)
            conv(q + "conv3", planes * 4, planes, 1, 1), bn(q + "bn3", planes * 4)
            if b == 0:
                conv(q + "downsample.0", planes * 4, inplanes, 1, 1), bn(q + "downsample.1", planes * 4)
            inplanes = planes * 4
    for k, (cin, mid, cout) in enumerate(((2048, 512, 512), (1536, 256, 256), (768, 128, 128), (384, 64, 32)), start=1):
        q = f"decoder.block{k}."
        conv(q + "conv1x1.0", mid, cin, 1, 1, True), bn(q + "conv1x1.1", mid)
        conv(q + "conv3x3.0", cout, mid, 3, 3, True), bn(q + "conv3x3.1", cout)
    conv("output_head.score_map", 1, 32, 1, 1, True)
    conv("output_head.geo_map", 8, 32, 1, 1, True)
    return shapes


def complete_state_dict(state_dict, seed=0):
    """`load_state_dict(state, strict=False)` as the reference loads its detector (east.py:130-133): unexpected keys (a ResNet-101
    training checkpoint's layer3.6-22, num_batches_tracked, an fc head) are ignored; a MISSING key keeps what a freshly constructed
    module holds — PyTorch's default initialisation (Conv2d: kaiming_uniform(a = sqrt 5) = U(+-1/sqrt(fan_in)) for weight and bias;
    BatchNorm2d: weight 1, bias 0, running_mean 0, running_var 1), drawn here from a seeded generator; a key of the wrong shape
    raises RuntimeError, as load_state_dict does even when strict=False.  Returns (complete dict, missing keys, unexpected keys)."""
    shapes = expected_state_shapes()
    sd = {}
    g = torch.Generator().manual_seed(seed)
    missing, bad = [], []
    for k, shp in shapes.items():
        v = state_dict.get(k)
        if v is not None:
            if tuple(v.shape) != shp:
                bad.append(f"size mismatch for {k}: copying a param with shape {tuple(v.shape)} from checkpoint, the shape in current model is {shp}")
                continue
            sd[k] = v
            continue
        missing.append(k)
        if k.endswith("running_mean") or (k.endswith(".bias") and len(shp) == 1 and k[: -len(".bias")] + ".running_mean" in shapes):
            sd[k] = torch.zeros(shp)
        elif k.endswith("running_var") or (k.endswith(".weight") and len(shp) == 1):
            sd[k] = torch.ones(shp)
        else:
            wk = k[: k.rfind(".")] + ".weight"
            fan_in = shapes[wk][1] * shapes[wk][2] * shapes[wk][3]
            bound = 1.0 / fan_in ** 0.5
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound
    if bad:
        raise RuntimeError("Error(s) in loading state_dict for EAST:\n\t" + "\n\t".join(bad))
    nbt = ".num_batches_tracked"  # a BatchNorm2d buffer of the module itself: expected, unused at inference
    unexpected = [k for k in state_dict if k not in shapes and not (k.endswith(nbt) and k[: -len(nbt)] + ".running_mean" in shapes)]
    return sd, missing, unexpected


def fold_bn(w, conv_bias, prefix_bn, sd):
    """conv (OIHW f32) followed by eval-mode BatchNorm -> (w', b') in f32."""
    gamma, beta = sd[prefix_bn + ".weight"].float(), sd[prefix_bn + ".bias"].float()
    mean, var = sd[prefix_bn + ".running_mean"].float(), sd[prefix_bn + ".running_var"].float()
    s = gamma / torch.sqrt(var + BN_EPS)
    w2 = w.float() * s.view(-1, 1, 1, 1)
    b2 = beta - mean * s
    if conv_bias is not None:
        b2 = b2 + conv_bias.float() * s
    return w2, b2


def to_khwc(w, dtype, device, split=None, square=True):
    """OIHW -> [Cout][KH][KW][Cin] on the device; 3x3 f32 weights also get their Winograd twins (ops.attach_winograd; square=False:
    no F(4,3) x F(4,3) twin, the layer stays on the tall form), f32 1x1 weights and the Winograd U their three bf16 planes
    (ops.attach_split; split=False keeps a layer on the exact-f32 MFMA)."""
    return ops.attach_split(ops.attach_winograd(w.permute(0, 2, 3, 1).contiguous().to(dtype).to(device), split, square), split)


# The detector's 3x3 layers stay on the TALL Winograd form: the geometry map carries the tightest tolerance of the path (BASELINE.md:
# 1e-4 absolute on offsets of up to tens of pixels), the tall form on the round-4 interpolation points has half the rounding error
# of anything before it, and these layers are 4 ms of a 183 ms step (the square form would save 2.5 ms and spend the margin:
# f64 arbitration of the geometry 1.84 of the allowed 2.0 with it).  MSOCR_EAST_WINO_SQUARE=1 switches them over.
EAST_WINO_SQUARE = os.environ.get("MSOCR_EAST_WINO_SQUARE", "0") != "0"


def pack_stem_weight(w, cin_pad, cpad=4):
    """OIHW stem weight (C=3) -> [Cout][KH][1][cin_pad] with element kw*cpad+c = w[co,c,kh,kw], zeros elsewhere."""
    Cout, C, KH, KW = w.shape
    assert C <= cpad and KW * cpad <= cin_pad
    taps = torch.zeros(Cout, KH, KW, cpad, dtype=torch.float32)
    taps[..., :C] = w.float().permute(0, 2, 3, 1)
    out = torch.zeros(Cout, KH, 1, cin_pad, dtype=torch.float32)
    out[:, :, 0, : KW * cpad] = taps.reshape(Cout, KH, KW * cpad)
    return out


def stem_view(canvas, cin_pad, kw=None):
    """[N,Hp,Wp,cpad] canvas -> overlapping-window view [N,Hp,Wp-cin_pad/cpad+1,cin_pad] (in_sW = cpad)."""
    N, Hp, Wp, C = canvas.shape
    assert C in (4, 8) and canvas.is_contiguous()
    return torch.as_strided(canvas, (N, Hp, Wp - cin_pad // C + 1, cin_pad), (Hp * Wp * C, Wp * C, C, 1))


class EastNet:
    def __init__(self, state_dict, dtype=torch.float32, device="cuda", split=None):
        """split: None = ops.SPLIT_BF16X3 (default on), False = exact-f32 MFMA for every f32 layer (precision="fp32-exact")."""
        self.dtype, self.device = dtype, torch.device(device)
        sd, self.missing_keys, self.unexpected_keys = complete_state_dict(state_dict)  # strict=False, as east.py:130-133
        P = {}

        def conv_bn(name_conv, name_bn, bias_key=None):
            w, b = fold_bn(sd[name_conv + ".weight"], sd.get(bias_key) if bias_key else None, name_bn, sd)
            return w, b

        bb = "backbone.extractor."
        w, b = conv_bn(bb + "conv1", bb + "bn1")
        P["stem"] = (pack_stem_weight(w, 32).to(dtype).to(self.device), b.to(self.device))
        for lname, planes, blocks, stride in LAYERS:
            for i in range(blocks):
                p = f"{bb}{lname}.{i}."
                for j in (1, 2, 3):
                    w, b = conv_bn(p + f"conv{j}", p + f"bn{j}")
                    P[f"{lname}.{i}.conv{j}"] = (to_khwc(w, dtype, self.device, split, EAST_WINO_SQUARE), b.to(self.device))
                if i == 0:
                    wd, bd = conv_bn(p + "downsample.0", p + "downsample.1")
                    P[f"{lname}.{i}.down"] = (to_khwc(wd, dtype, self.device, split, EAST_WINO_SQUARE), bd.to(self.device))
                    if stride == 1:
                        # conv3 and the stride-1 downsample of the layer's first block read the same pixels: one GEMM over the
                        # channel concatenation [conv2 output | block input] with [W3 | Wd] (K = planes + Cin) replaces two
                        # launches, the downsample's output round trip through HBM and the residual read (torchvision Bottleneck:
                        # out = relu(bn3(conv3(.)) + downsample(x)))
                        P[f"{lname}.{i}.conv3d"] = (to_khwc(torch.cat([w, wd], dim=1), dtype, self.device, split), (b + bd).to(self.device))
        for k in (1, 2, 3, 4):
            p = f"decoder.block{k}."
            w, b = conv_bn(p + "conv1x1.0", p + "conv1x1.1", p + "conv1x1.0.bias")
            P[f"dec{k}.a"] = (to_khwc(w, dtype, self.device, split, EAST_WINO_SQUARE), b.to(self.device))
            w, b = conv_bn(p + "conv3x3.0", p + "conv3x3.1", p + "conv3x3.0.bias")
            P[f"dec{k}.b"] = (to_khwc(w, dtype, self.device, split, EAST_WINO_SQUARE), b.to(self.device))
        w9 = torch.cat([sd["output_head.score_map.weight"].float().view(1, 32), sd["output_head.geo_map.weight"].float().view(8, 32)])
        b9 = torch.cat([sd["output_head.score_map.bias"].float().view(1), sd["output_head.geo_map.bias"].float().view(8)])
        self.w9, self.b9 = w9.contiguous().to(self.device), b9.contiguous().to(self.device)
        self.P = P

    # -------------------------------------------------------------------------------------
    def _bottleneck(self, x, lname, i, stride, out=None, yx=None):
        """yx (first block of layer1 only): [N,h,w,planes+Cin] buffer whose last Cin channels ARE x; conv2 writes beside them."""
        P = self.P
        w1, b1 = P[f"{lname}.{i}.conv1"]
        w2, b2 = P[f"{lname}.{i}.conv2"]
        w3, b3 = P[f"{lname}.{i}.conv3"]
        o1 = ops.conv2d(x, w1, b1, relu=True)
        if yx is not None:
            w3d, b3d = P[f"{lname}.{i}.conv3d"]
            ops.conv2d(o1, w2, b2, pad=(1, 1), relu=True, out=yx[..., : w2.shape[0]])
            return ops.conv2d(yx, w3d, b3d, relu=True, out=out)
        o2 = ops.conv2d(o1, w2, b2, stride=(stride, stride), pad=(1, 1), relu=True)
        if i == 0:
            wd, bd = P[f"{lname}.{i}.down"]
            idt = ops.conv2d(x, wd, bd, stride=(stride, stride))
        else:
            idt = x
        return ops.conv2d(o2, w3, b3, relu=True, residual=idt, out=out)

    def forward(self, pages_u8):
        """pages_u8 [N,H,W,3] u8 on device, H and W multiples of 32 ->
        (score [N,H/4,W/4] f32, geo [N,H/4,W/4,8] f32)."""
        N, H, W, _ = pages_u8.shape
        if H % 32 or W % 32:
            raise ValueError("EAST network input must be a multiple of 32 in both dimensions (east.py:87-92 concats)")
        dt, dev = self.dtype, self.device
        canvas = ops.normalize_u8(pages_u8, 3, 3, H + 6, W + 6, 0, dt)
        ws, bs = self.P["stem"]
        x = ops.conv2d(stem_view(canvas, 32, 7), ws, bs, (2, 2), (0, 0), True, out_hw=(H // 2, W // 2), alg_k=147)
        del canvas
        yx = torch.empty((N, H // 4, W // 4, 128), dtype=dt, device=dev)  # layer1.0: [conv2 output | block input], see conv3d
        x = ops.maxpool2d(x, 3, 2, 1, out=yx[..., 64:])
        cat1 = torch.empty((N, H // 4, W // 4, 128 + 256), dtype=dt, device=dev)
        cat2 = torch.empty((N, H // 8, W // 8, 256 + 512), dtype=dt, device=dev)
        cat3 = torch.empty((N, H // 16, W // 16, 512 + 1024), dtype=dt, device=dev)
        taps = {"layer1": cat1[..., 128:], "layer2": cat2[..., 256:], "layer3": cat3[..., 512:], "layer4": None}
        for lname, planes, blocks, stride in LAYERS:
            for i in range(blocks):
                x = self._bottleneck(x, lname, i, stride if i == 0 else 1, out=taps[lname] if i == blocks - 1 else None,
                                     yx=yx if (lname == "layer1" and i == 0) else None)
            yx = None
        P = self.P

        def dec(k, t):
            wa, ba = P[f"dec{k}.a"]
            wb, bb_ = P[f"dec{k}.b"]
            return ops.conv2d(ops.conv2d(t, wa, ba, relu=True), wb, bb_, pad=(1, 1), relu=True)

        h4 = dec(1, x)
        ops.upsample2x_into(h4, cat3)
        h3 = dec(2, cat3)
        ops.upsample2x_into(h3, cat2)
        h2 = dec(3, cat2)
        ops.upsample2x_into(h2, cat1)
        h1 = dec(4, cat1)
        return ops.east_head(h1, self.w9, self.b9)


def east_conv_macs(H, W):
    """Algorithmic multiply-accumulates of one page through the network."""
    macs = (H // 2) * (W // 2) * 64 * 147
    h, w, cin = H // 4, W // 4, 64
    for lname, planes, blocks, stride in LAYERS:
        for i in range(blocks):
            s = stride if i == 0 else 1
            macs += h * w * cin * planes
            h2, w2 = h // s, w // s
            macs += h2 * w2 * planes * planes * 9 + h2 * w2 * planes * planes * 4
            if i == 0:
                macs += h2 * w2 * cin * planes * 4
            h, w, cin = h2, w2, planes * 4
    for cin_, mid, cout, div in ((2048, 512, 512, 32), (1536, 256, 256, 16), (768, 128, 128, 8), (384, 64, 32, 4)):
        px = (H // div) * (W // div)
        macs += px * (cin_ * mid + mid * cout * 9)
    macs += (H // 4) * (W // 4) * 32 * 9
    return macs
