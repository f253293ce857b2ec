"""Single-page latency of Pipeline.predict_batch([page]) / predict(page) on a synthetic 2048x1536 page (dev tool, GPU only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manuscript_ocr_amd import Pipeline, synth
from manuscript_ocr_amd.detectors import EAST
from manuscript_ocr_amd.recognizers import TRBA

H, W = 1536, 2048
det = EAST(state_dict=synth.east_state_dict(), target_size=(W, H), device="cuda")
rec = TRBA(state_dict=synth.trba_state_dict_confident(194, 256), config={"img_h": 32, "img_w": 100, "max_len": 25, "hidden_size": 256}, device="cuda")
pipe = Pipeline(det, rec)
pg, rects = synth.synth_page(200, H, W)
s, g = synth.synth_maps(rects, (H, W), (H // 4, W // 4), 200)
mo = (torch.from_numpy(s)[None].cuda(), torch.from_numpy(g)[None].cuda())
for name, fn in (("predict_batch([page]) with injected maps, host page -> Page", lambda: pipe.predict_batch([pg], _maps_override=mo)[0]),):
    for _ in range(2): out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {dt * 1e3:.1f} ms/page, {len(out.blocks[0].words)} words; host stages {({k: round(v, 4) for k, v in pipe.last_profile.items()})}")
print("max memory reserved GB:", torch.cuda.max_memory_reserved() / 1e9)
