"""Times the stages of the JPEG ingest of one bench batch (16 pages 2048 x 1536, quality 90, 4:2:0) on the GPU box (dev tool):
host marker walk, upload, device Huffman kernel (one thread per restart interval) for several interval lengths, reconstruction —
and the host thread-pool entropy decode of the same pages without restart markers.

    python tools/jpeg_huffman_time.py [pages]
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manuscript_ocr_amd import ingest, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    pages = [synth.synth_page(100 + k, 2048, 1536)[0] for k in range(n)]
    with tempfile.TemporaryDirectory(prefix="msocr_jt_", dir="/tmp") as td:
        for label, kw in (("rows=1", {"restart_marker_rows": 1}), ("rows=4", {"restart_marker_rows": 4}), ("blocks=32", {"restart_marker_blocks": 32}),
                          ("blocks=16", {"restart_marker_blocks": 16}), ("blocks=4", {"restart_marker_blocks": 4})):
            paths = []
            for k, pg in enumerate(pages):
                paths.append(os.path.join(td, f"{label}_{k}.jpg"))
                Image.fromarray(pg).save(paths[-1], quality=90, **kw)
            size = sum(os.path.getsize(p) for p in paths) / n
            t0 = time.perf_counter()
            parsed = [ingest._read_and_parse(p) for p in paths]
            t1 = time.perf_counter()
            batch = ingest.ScanBatch(parsed)
            t2 = time.perf_counter()
            bytes_dev = torch.from_numpy(batch.bytes).to("cuda")
            for _ in range(2):
                coef, status = ingest.entropy_batch_device(batch, bytes_dev=bytes_dev)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                coef, status = ingest.entropy_batch_device(batch, bytes_dev=bytes_dev)
            e1.record()
            torch.cuda.synchronize()
            assert not status.cpu().numpy().any()
            t3 = time.perf_counter()
            ingest.read_images_device(paths, device_entropy=True)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for _ in range(3):
                out = ingest.read_images_device(paths, device_entropy=True)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            print(f"{label}: {size / 1e3:.0f} kB/page, {batch.max_intervals} intervals/page; read+parse {1e3 * (t1 - t0):.1f} ms, marker walk + layout "
                  f"{1e3 * (t2 - t1):.1f} ms (serial, one thread), tables + bounds upload + memset + Huffman kernel {e0.elapsed_time(e1) / 5:.2f} ms per batch of {n}; "
                  f"read_images_device end to end {1e3 * (t4 - t3) / 3:.1f} ms per batch", flush=True)
        paths = []
        for k, pg in enumerate(pages):
            paths.append(os.path.join(td, f"plain_{k}.jpg"))
            Image.fromarray(pg).save(paths[-1], quality=90)
        ingest.read_images_device(paths)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ingest.read_images_device(paths)
        torch.cuda.synchronize()
        print(f"no restart markers (host thread pool): read_images_device {1e3 * (time.perf_counter() - t0) / 3:.1f} ms per batch of {n}", flush=True)


if __name__ == "__main__":
    main()
