"""Multi-GPU sharding of pages: one process per GPU, contiguous page ranges per rank, NO data-path
collective (pages are independent units).  The single exchange is the final gather of decoded
records: an all_gather of per-rank byte counts followed by one all_gather of a padded uint8
buffer (RCCL over xGMI on GPUs — backend "nccl" is RCCL on ROCm — or gloo on CPU in tests).
The reference has no distributed code at all; this is new in the build (§8e).
"""
import json
from typing import Any, List, Tuple

import numpy as np
import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of `n_items` owned by `rank`; the first n % world ranks get one extra."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def page_records(page_id: int, page) -> List[dict]:
    """Flat records of one recognised Page: id, word index, box, confidences, text."""
    recs = []
    k = 0
    for block in page.blocks:
        for w in block.words:
            recs.append({"page": page_id, "word": k, "polygon": [list(map(float, p)) for p in w.polygon],
                         "det": w.detection_confidence, "text": w.text, "rec": w.recognition_confidence})
            k += 1
    return recs


def gather_records(local_records: List[Any], device: torch.device) -> List[Any]:
    """All ranks' records concatenated in rank order (every rank receives them; rank 0 uses them)."""
    import torch.distributed as dist

    payload = np.frombuffer(json.dumps(local_records, ensure_ascii=False).encode("utf-8"), dtype=np.uint8)
    if not (dist.is_available() and dist.is_initialized()):
        return json.loads(payload.tobytes().decode("utf-8"))
    world = dist.get_world_size()
    n_local = torch.tensor([payload.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    buf = torch.zeros(max(max(sizes), 1), dtype=torch.uint8, device=device)
    buf[: payload.size] = torch.from_numpy(payload.copy()).to(device)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = []
    for b, n in zip(bufs, sizes):
        out += json.loads(b[:n].cpu().numpy().tobytes().decode("utf-8"))
    return out
