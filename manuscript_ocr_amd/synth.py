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
