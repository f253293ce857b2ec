"""CPU-only tests of the host logic and of the C-ABI surface (no compute calls without a GPU)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from manuscript_ocr_amd import _native
    L = _native.lib()
    header = open(os.path.join(ROOT, "include", "msocr.h")).read()
    declared = set(re.findall(r"\b(msocr_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), f"libmsocr.so lacks {name} declared in include/msocr.h"
    assert declared == set(_native.exported_symbols()), declared ^ set(_native.exported_symbols())
    assert b"gfx950" in L.msocr_version()


def test_native_ops_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from manuscript_ocr_amd import _native, ops
    from manuscript_ocr_amd.detectors import EAST
    from manuscript_ocr_amd.recognizers import TRBA
    with pytest.raises(RuntimeError):
        EAST(state_dict={})
    with pytest.raises(RuntimeError):
        TRBA(state_dict={}, config={})
    with pytest.raises(_native.NativeError):
        ops.maxpool2d(torch.zeros(1, 4, 4, 4), 2, 2, 0)


def test_error_contract_paths(tmp_path):
    from manuscript_ocr_amd.detectors import read_image
    from manuscript_ocr_amd.recognizers import TRBA
    with pytest.raises(FileNotFoundError):
        read_image(str(tmp_path / "missing.jpg"))
    with pytest.raises(TypeError):
        read_image(12345)
    with pytest.raises(FileNotFoundError):
        TRBA(model_path=str(tmp_path / "nope.pth"))
    w = tmp_path / "w.pth"
    w.write_bytes(b"x")
    with pytest.raises(FileNotFoundError):
        TRBA(model_path=str(w), config_path=str(tmp_path / "nope.json"))
    with pytest.raises(FileNotFoundError):
        TRBA(model_path=str(w), charset_path=str(tmp_path / "nope.txt"))
    with pytest.raises(ValueError):
        TRBA(model_path=str(w), weights_path=str(tmp_path / "other.pth"))
    with pytest.raises(TypeError):
        TRBA(model_path=str(w), bogus=1)


def test_host_post_and_transforms_match_oracle(golden_dir):
    from manuscript_ocr_amd.detectors._east import post as P
    from manuscript_ocr_amd.recognizers._trba import transforms as T
    from oracle import east_post as O
    from oracle import imgproc
    g = np.load(os.path.join(golden_dir, "east_post.npz"))
    assert np.array_equal(P.expand_boxes(g["lanms_q2"], 0.9, 0.9).view(np.uint32), g["expanded"].view(np.uint32))
    q = O.scale_boxes_to_original(g["expanded"], (1000, 1400), (1024, 768))
    assert np.array_equal(P.scale_boxes(g["expanded"], (1000, 1400), (1024, 768)).view(np.uint32), q.view(np.uint32))
    # nest a few quads inside existing ones (centre-shrunk copies, one touching an edge) so the filter has work
    big = q[np.argsort(-O.polygon_area_batch(q[:, :8].reshape(-1, 4, 2)))[:6]].copy()
    c = big[:, :8].reshape(-1, 4, 2).mean(axis=1, keepdims=True)
    inner = big.copy()
    inner[:, :8] = (c + 0.4 * (big[:, :8].reshape(-1, 4, 2) - c)).reshape(-1, 8)
    inner[0, :2] = big[0, :2]  # shares a vertex with its container: on-edge counts as inside
    q = np.concatenate([q, inner]).astype(np.float32)
    a, b = O.remove_fully_contained_boxes(q), P.remove_contained(q)
    assert len(a) < len(q) and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(O.remove_area_anomalies(a).view(np.uint32), P.remove_area_anomalies(a).view(np.uint32))
    assert np.array_equal(O.convert_to_axis_aligned(a).view(np.uint32), P.to_axis_aligned(a).view(np.uint32))
    rng = np.random.default_rng(0)
    for (h, w) in ((20, 44), (32, 100), (70, 300), (15, 15), (64, 500), (9, 200)):
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        for (ih, iw) in ((32, 100), (64, 256)):
            assert np.array_equal(T.resize_and_pad(img, ih, iw), imgproc.resize_and_pad(img, ih, iw)), (h, w, ih, iw)
    gray = rng.integers(0, 256, size=(20, 30), dtype=np.uint8)
    assert T.resize_and_pad(gray, 32, 100).shape == (32, 100, 3)


def test_pipeline_glue_matches_golden(golden_dir):
    import json
    from manuscript_ocr_amd.detectors import sort_boxes_reading_order, sort_boxes_reading_order_with_resolutions
    from manuscript_ocr_amd.detectors._east.utils import resolve_intersections
    for c in json.load(open(os.path.join(golden_dir, "pipeline_glue.json"))):
        boxes = [tuple(np.int32(v) for v in b) for b in c["boxes"]]
        assert [list(map(int, b)) for b in resolve_intersections(boxes)] == c["resolved"]
        assert [list(map(int, b)) for b in sort_boxes_reading_order(boxes)] == c["sorted"]
        assert [list(map(int, b)) for b in sort_boxes_reading_order_with_resolutions(boxes)] == c["sorted_res"]


def test_shard_range_partitions():
    from manuscript_ocr_amd.dist import shard_range
    for n in (0, 1, 7, 16, 128, 129):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["REPO"])
import torch, torch.distributed as dist
from manuscript_ocr_amd.dist import shard_range, gather_records
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
lo, hi = shard_range(7, rank, world)
local = [{"page": p, "word": 0, "text": "стр%d — ok" % p, "rec": 0.5 + p / 100} for p in range(lo, hi)]
allr = gather_records(local, torch.device("cpu"))
if rank == 0:
    print("RESULT" + json.dumps(allr, ensure_ascii=False))
dist.barrier()
dist.destroy_process_group()
'''


def test_gather_records_world2_gloo(tmp_path):
    """The N>1 exchange path (sizes all_gather + padded u8 all_gather) with 2 CPU ranks over gloo."""
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, REPO=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-400:] for o in outs]
    import json
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT")][0]
    recs = json.loads(line[len("RESULT"):])
    assert [r["page"] for r in recs] == list(range(7))
    assert recs[5]["text"] == "стр5 — ok"


WORKER8 = r'''
import os, sys, json
sys.path.insert(0, os.environ["REPO"])
import torch, torch.distributed as dist
torch.set_num_threads(1)
from manuscript_ocr_amd.dist import shard_range, gather_records
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
n_pages = int(os.environ["N_PAGES"])
lo, hi = shard_range(n_pages, rank, world)
# page 2 has no words at all; with 5 pages over 8 ranks, ranks 5..7 own no page: zero records, a zero-length payload "[]"
local = [{"page": p, "word": k, "text": "p%d/w%d é" % (p, k)} for p in range(lo, hi) for k in range(0 if p == 2 else 1 + p % 3)]
allr = gather_records(local, torch.device("cpu"))
if rank == 0:
    print("RESULT" + json.dumps({"recs": allr, "spans": [shard_range(n_pages, r, world) for r in range(world)]}, ensure_ascii=False))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("n_pages", [13, 5])
def test_gather_records_world8_gloo_uneven_and_empty_ranks(tmp_path, n_pages):
    """SURVEY 8(e) at the node's width: 8 CPU ranks over gloo, a page count not divisible by 8 (13) and fewer pages than ranks (5:
    three ranks contribute ZERO records, one page has no words) — the padded all_gather must size itself from max(sizes, 1), and the
    gathered records must come back in GLOBAL page order (rank order = page order because the shards are contiguous)."""
    import json
    script = tmp_path / "w8.py"
    script.write_text(WORKER8)
    port = str(29741 + n_pages)
    env = dict(os.environ, REPO=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE="8", N_PAGES=str(n_pages), OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(8)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-400:] for o in outs]
    res = json.loads([l for l in outs[0][0].splitlines() if l.startswith("RESULT")][0][len("RESULT"):])
    want = [(p, k) for p in range(n_pages) for k in range(0 if p == 2 else 1 + p % 3)]
    assert [(r["page"], r["word"]) for r in res["recs"]] == want
    assert res["recs"][0]["text"] == "p0/w0 é"
    sizes = [b - a for a, b in res["spans"]]
    assert sum(sizes) == n_pages and max(sizes) - min(sizes) <= 1 and (n_pages >= 8 or sizes.count(0) == 8 - n_pages)


def test_vectorised_glue_equals_literal_restatement_on_random_boxes():
    """The host fast paths (vectorised resolve_intersections, running-sum line grouping, dict re-match) against the
    oracle's literal restatement of the reference loops, on random overlapping boxes incl. duplicates."""
    from manuscript_ocr_amd.detectors._east import utils as U
    from oracle import pipeline_glue as G
    rng = np.random.default_rng(0)
    for trial in range(25):
        n = int(rng.integers(1, 90))
        boxes = []
        for _ in range(n):
            x0, y0 = rng.integers(0, 400, 2)
            w, h = rng.integers(3, 120), rng.integers(3, 40)
            boxes.append(tuple(np.int32(v) for v in (x0, y0, x0 + w, y0 + h)))
        if n > 3:
            boxes[1] = boxes[0]  # duplicate AABBs collapse in the reference's dict mapping
        as_int = lambda bs: [tuple(int(v) for v in b) for b in bs]
        assert as_int(U.resolve_intersections(boxes)) == as_int(G.resolve_intersections(boxes))
        assert as_int(U.sort_boxes_reading_order(boxes)) == as_int(G.sort_boxes_reading_order(boxes))
        assert as_int(U.sort_boxes_reading_order_with_resolutions(boxes)) == as_int(G.sort_boxes_reading_order_with_resolutions(boxes))


def test_pipeline_order_boxes_equals_oracle_order_and_crop():
    """Pipeline._order_boxes (+ crop descriptors) vs the oracle's literal _pipeline.py:102-137 on random word layouts."""
    from manuscript_ocr_amd import Pipeline, ops, synth
    from manuscript_ocr_amd.detectors._types import Block, Page, Word
    from oracle import pipeline_glue as G
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(600, 900, 3), dtype=np.uint8)
    pipe = Pipeline(detector=object(), recognizer=object(), min_text_size=5)
    for seed in range(6):
        rects = synth.synth_layout(seed, 600, 900)
        rects = rects[rng.permutation(len(rects))]
        polys = []
        for x0, y0, x1, y1 in rects:
            j = rng.uniform(-3, 3, 8)
            polys.append([[x0 + j[0], y0 + j[1]], [x1 + j[2], y0 + j[3]], [x1 + j[4], y1 + j[5]], [x0 + j[6], y1 + j[7]]])
        polys.append([[10.0, 10.0], [12.5, 10.0], [12.5, 13.0], [10.0, 13.0]])  # below min_text_size
        polys.append([[-20.0, -5.0], [40.0, -5.0], [40.0, 30.0], [-20.0, 30.0]])  # clamped at the border
        polys = [[[float(np.float32(c)) for c in pt] for pt in q] for q in polys]
        order, kept, crops = G.order_and_crop(polys, img, 5)
        page = Page(blocks=[Block(words=[Word(polygon=q, detection_confidence=0.5) for q in polys])])
        before = list(page.blocks[0].words)
        words, boxes = pipe._order_boxes(page)
        assert [before.index(w) for w in page.blocks[0].words] == order
        assert [page.blocks[0].words.index(w) for w in words] == kept
        desc, keep = ops.crop_descriptors(boxes, [0] * len(boxes), img.shape[:2], 32, 100)
        assert keep.all() and len(desc) == len(crops)
        for d, c in zip(desc, crops):
            assert (d[4] - d[2], d[3] - d[1]) == c.shape[:2]
            assert np.array_equal(img[d[2]:d[4], d[1]:d[3]], c)


def test_vectorised_texts_equal_decode_tokens():
    """TRBA.texts (vectorised) == decode_tokens over ids[:t_run] row by row (transforms.py:196-206 semantics)."""
    import types

    from manuscript_ocr_amd.recognizers import TRBA
    from manuscript_ocr_amd.recognizers._trba.transforms import decode_tokens
    rng = np.random.default_rng(5)
    itos = [f"<{i}>" if i < 4 else chr(0x410 + i) for i in range(60)]
    for blank in (None, 3):
        stub = types.SimpleNamespace(itos=itos, pad_id=0, eos_id=2, blank_id=blank)
        ids = rng.integers(0, 60, size=(200, 12)).astype(np.int32)
        ids[rng.random((200, 12)) < 0.15] = 2
        ids[rng.random((200, 12)) < 0.10] = 0
        trun = rng.integers(0, 13, size=200).astype(np.int32)
        for j in range(200):
            ids[j, trun[j]:] = -1  # what msocr_attn_beam_finalize leaves beyond t_run
        exp = [decode_tokens(ids[j, : int(trun[j])], itos, 0, 2, blank) for j in range(200)]
        assert TRBA.texts(stub, ids, trun) == exp
    assert TRBA.texts(stub, np.zeros((0, 5), np.int32), np.zeros((0,), np.int32)) == []


def test_device_batches_never_split_a_reference_chunk():
    """TRBA._device_batches: launches of <= device_batch rows aligned to the reference's chunks (slices of batch_size rows
    inside a span); per-launch metadata = chunk id of every row + chunk sizes; no chunk information when rows are uncovered."""
    import types

    from manuscript_ocr_amd.recognizers import TRBA
    stub = types.SimpleNamespace(device_batch=100)
    bounds, metas = TRBA._device_batches(stub, 118, [(0, 70), (70, 45), (115, 3)], 32)
    assert bounds == [(0, 70), (70, 118)]
    assert metas[0][70:].tolist() == [32, 32, 6] and metas[1][48:].tolist() == [32, 13, 3]
    assert metas[0][:70].tolist() == [0] * 32 + [1] * 32 + [2] * 6
    assert metas[1][:48].tolist() == [0] * 32 + [1] * 13 + [2] * 3
    bounds, metas = TRBA._device_batches(stub, 250, [(0, 250)], 32)
    assert bounds == [(0, 96), (96, 192), (192, 250)] and [len(m) for m in metas] == [99, 99, 60]
    for (lo, hi), m in zip(bounds, metas):
        ids, sizes = m[: hi - lo], m[hi - lo:]
        assert np.array_equal(np.bincount(ids), sizes) and sizes.max() <= 32
    assert TRBA._device_batches(stub, 10, [(5, 5)], 32) == ([(0, 10)], None)
    big = types.SimpleNamespace(device_batch=8)   # a chunk larger than a launch is kept whole
    assert TRBA._device_batches(big, 40, [(0, 40)], 32)[0] == [(0, 32), (32, 40)]


def test_no_packed_f32_valu_beside_mfma(tmp_path):
    """Every translation unit the Makefile lists in NOPK_OBJS (kernels that run beside bf16 MFMAs) must contain NO packed-f32 VALU
    instruction: v_pk_fma_f32 with op_sel returns wrong low results in lanes 48..63 beside v_mfma_f32_32x32x16_bf16
    (tools/microbench/pk_fma_beside_mfma.hip, profiles/r04_pk_fma_probe.txt) and any packed f32 op there costs matrix-pipe time
    (profiles/r04_pp_ablations.txt).  Checked on the BUILT objects: the device code object is unbundled and disassembled."""
    import re
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "manuscript_ocr_amd", "csrc")
    bundler, objdump = "/opt/rocm/lib/llvm/bin/clang-offload-bundler", "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(bundler) and os.path.exists(objdump) and shutil.which("objcopy")):
        pytest.skip("ROCm binutils not present")
    mk = open(os.path.join(csrc, "Makefile")).read()
    objs = re.search(r"^NOPK_OBJS = (.*)$", mk, re.M).group(1).split()
    assert {"attn_beam_mfma.o", "conv_split_pp.o", "conv_split.o", "bilstm_mfma.o"} <= set(objs)
    subprocess.check_call(["make", "-s", "-j4", "-C", csrc, "ARCH=gfx950"])   # no-op when built
    for o in objs:
        fat, co = str(tmp_path / (o + ".fat")), str(tmp_path / (o + ".co"))
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", os.path.join(csrc, o), fat])
        subprocess.check_call([bundler, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
        asm = subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True).stdout
        assert "s_endpgm" in asm, o
        packed = re.findall(r"v_pk_(?:fma|add|mul)_f32[^\n]*", asm)
        assert not packed, f"{o}: {len(packed)} packed-f32 VALU instructions, e.g. {packed[0]}"


def test_oracle_is_only_a_checker():
    """The CPU restatement under oracle/ is test infrastructure: the product package never imports it, bench.py only inside
    its cpu_baseline leg (cpu_baseline and the cpu_baseline_* helpers only it calls), __graft_entry__ only inside smoke()."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_imports(tree):
        hits = []
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                hits.append(node)
            elif isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                hits.append(node)
        return hits

    for dirpath, _, files in os.walk(os.path.join(root, "manuscript_ocr_amd")):
        for f in files:
            if f.endswith(".py"):
                tree = ast.parse(open(os.path.join(dirpath, f)).read())
                assert not oracle_imports(tree), f"{f} imports the oracle"
    for fname, allowed in (("bench.py", "cpu_baseline"), ("__graft_entry__.py", "smoke")):
        tree = ast.parse(open(os.path.join(root, fname)).read())
        inside = set()
        for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
            if oracle_imports(fn):
                inside.add(fn.name)
        assert all(n.startswith(allowed) for n in inside), (fname, inside)
        assert len(oracle_imports(tree)) == sum(len(oracle_imports(fn)) for fn in ast.walk(tree)
                                                if isinstance(fn, ast.FunctionDef) and fn.name.startswith(allowed)), (fname, inside)
        if fname == "bench.py":  # the cpu_baseline_* helpers are reachable from cpu_baseline only
            src = open(os.path.join(root, fname)).read()
            for n in inside - {allowed}:
                calls = [c for c in ast.walk(tree) if isinstance(c, ast.Call) and getattr(c.func, "id", None) == n]
                owners = [f.name for f in ast.walk(tree) if isinstance(f, ast.FunctionDef) and any(c in list(ast.walk(f)) for c in calls)]
                assert calls and all(o.startswith(allowed) for o in owners), (n, owners)


def test_reading_order_host_helper_equals_python_glue(golden_dir):
    """msocr_reading_order_host (C++ host helper) == sort_boxes_reading_order_with_resolutions + first-equal-word re-match in
    Python, on the reference-generated fixture boxes and on random layouts with overlaps, zero-size and duplicate boxes."""
    import json

    from manuscript_ocr_amd._pipeline import _reading_order
    from manuscript_ocr_amd.detectors import sort_boxes_reading_order_with_resolutions

    def python_order(aabbs):
        first = {}
        for k, bx in enumerate(aabbs):
            first.setdefault(tuple(int(v) for v in bx), k)
        return [first[tuple(int(v) for v in bx)] for bx in sort_boxes_reading_order_with_resolutions(aabbs)]

    cases = [c["boxes"] for c in json.load(open(os.path.join(golden_dir, "pipeline_glue.json"))) if c["boxes"]]
    rng = np.random.default_rng(1)
    for n in (1, 2, 30, 200):
        for rep in range(4):
            x0, y0 = rng.integers(0, 900, size=n), rng.integers(0, 600, size=n)
            w, h = rng.integers(0, 120 if rep < 2 else 30, size=n), rng.integers(0, 40 if rep < 2 else 12, size=n)
            b = np.stack([x0, y0, x0 + w, y0 + h], 1)
            if n > 3 and rep % 2:
                b[n // 2], b[n - 1] = b[0], b[1]
            cases.append(b.tolist())
    for c in cases:
        as_np = [tuple(np.int32(v) for v in bx) for bx in c]  # what the pipeline passes (np.array(polygon, int32))
        assert _reading_order(np.array(c, dtype=np.int32)) == python_order(as_np), c[:5]


def test_box_tail_host_twin_equals_numpy_tail():
    """msocr_east_box_tail_host (the __host__ __device__ arithmetic of the device box filters, run on the CPU) == the NumPy
    host tail (expand_boxes, scale, contained boxes, area anomalies with NumPy's pairwise f32 sums, axis-aligned): bit-identical
    on random layouts with nested / giant / reversed / integer-coordinate quads and all parameter combinations."""
    from manuscript_ocr_amd import _native as nat
    from manuscript_ocr_amd.detectors._east import post
    rng = np.random.default_rng(5)
    for trial in range(120):
        M = int(rng.choice([0, 1, 2, 5, 20, 31, 32, 60, 200, 400]))
        cx, cy = rng.random(M) * 1800, rng.random(M) * 1400
        w, h = rng.random(M) * 150 + 2, rng.random(M) * 40 + 2
        if trial % 3 == 0 and M > 4:
            cx[1], cy[1], w[1], h[1] = cx[0], cy[0], w[0] * 0.5, h[0] * 0.5
            w[2], h[2] = 1500, 900
        ang = (rng.random(M) - 0.5) * 0.4
        pts = np.stack([np.stack([-w / 2, -h / 2], 1), np.stack([w / 2, -h / 2], 1), np.stack([w / 2, h / 2], 1), np.stack([-w / 2, h / 2], 1)], 1)
        if trial % 5 == 0:
            pts = pts[:, ::-1]
        c, s_ = np.cos(ang), np.sin(ang)
        R = np.stack([np.stack([c, -s_], 1), np.stack([s_, c], 1)], 1)
        pts = np.einsum("mij,mkj->mki", R, pts) + np.stack([cx, cy], 1)[:, None, :]
        q = np.concatenate([pts.reshape(M, 8), rng.random((M, 1))], 1).astype(np.float32)
        if trial % 7 == 0:
            q[:, :8] = np.round(q[:, :8])
        ew, eh = float(rng.choice([0.9, 0.0, 0.3])), float(rng.choice([0.9, 0.0, 0.5]))
        aa, anom, minc = bool(trial % 2), bool(trial % 4), int(rng.choice([30, 5]))
        ohw = (int(rng.choice([1536, 720, 4250])), int(rng.choice([2048, 1280, 5390])))
        twh = (int(rng.choice([1280, 2048])), int(rng.choice([1280, 1536])))
        e = post.expand_boxes(q.copy(), ew, eh)
        e = post.scale_boxes(e, ohw, twh)
        e = post.remove_contained(e)
        e = post.remove_area_anomalies(e, anom, 5.0, minc)
        e = post.to_axis_aligned(e) if aa else e
        out, n = np.empty((max(M, 1), 9), np.float32), ctypes.c_int32(0)
        rc = nat.lib().msocr_east_box_tail_host(np.ascontiguousarray(q).ctypes.data, M, ew, eh, ohw[1] / twh[0], ohw[0] / twh[1], int(aa), int(anom),
                                                5.0, minc, out.ctypes.data, ctypes.byref(n))
        assert rc == 0 and n.value == len(e) and np.array_equal(out[: n.value], e), (trial, M)


def test_vectorised_crop_descriptors_equal_the_loop():
    """ops.crop_descriptors (NumPy) == the literal per-box arithmetic (clamping incl. Python's negative-stop slices, min of the
    two scale factors, banker's rounding of the new size, vertical centring), on boxes inside, across and outside the page."""
    from manuscript_ocr_amd import ops
    rng = np.random.default_rng(17)
    for img_h, img_w in ((32, 100), (64, 256), (32, 128)):
        for trial in range(20):
            n = int(rng.integers(1, 300))
            x0, y0 = rng.integers(-60, 2100, size=n), rng.integers(-60, 1600, size=n)
            w, h = rng.integers(-5, 400, size=n), rng.integers(-5, 120, size=n)
            boxes = [(int(a), int(b), int(a + c), int(b + d)) for a, b, c, d in zip(x0, y0, w, h)]
            if trial % 4 == 0:  # exact .5 cases for the rounding rule: new size = k + 0.5 before rounding
                boxes += [(0, 0, 200, 20), (10, 10, 10 + 64, 10 + 13), (0, 0, 3, 64), (5, 5, 5 + 8, 5 + 5)]
            pids = [int(p) for p in rng.integers(0, 4, size=len(boxes))]
            d1, k1 = ops.crop_descriptors(boxes, pids, (1536, 2048), img_h, img_w)
            d2, k2 = ops._crop_descriptors_loop(boxes, pids, (1536, 2048), img_h, img_w)
            assert np.array_equal(k1, k2) and d1.dtype == d2.dtype and np.array_equal(d1, d2)
    d, k = ops.crop_descriptors([], [], (100, 100), 32, 100)
    assert d.shape == (0, 8) and k.shape == (0,)


def test_winograd_weight_transforms_host():
    """The load-time weight transforms are HOST functions of the C ABI (f64, rounded once): U = G g G^T for F(2x2,3x3), U = G6 g G4^T
    for the tall form F(4,3) x F(2,3) and U = G6 g G6^T for the square form, G6 = the Cook-Toom matrix of F(4,3) on the interpolation
    points {0, 3/2, -3/2, 2/3, -2/3, inf} (round 4), checked against their definitions without a GPU."""
    from manuscript_ocr_amd import _native as nat
    L = nat.lib()
    rng = np.random.default_rng(11)
    Cout, Cin = 32, 16
    w = rng.standard_normal((Cout, 3, 3, Cin)).astype(np.float32)  # [Cout][KH][KW][Cin]
    G4 = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float64)
    pts = [0.0, 1.5, -1.5, 2 / 3, -2 / 3]
    G6 = np.zeros((6, 3), np.float64)
    for j, a in enumerate(pts):   # G[j] = [1, a, a^2] / prod_{l != j}(a_j - a_l); the point at infinity: [0, 0, 1]
        G6[j] = np.array([1.0, a, a * a]) / np.prod([a - b for l, b in enumerate(pts) if l != j])
    G6[5, 2] = 1.0
    for fn, Gh, Gw, npts in ((L.msocr_winograd_weights_host, G4, G4, 16), (L.msocr_winograd42_weights_host, G6, G4, 24),
                             (L.msocr_winograd44_weights_host, G6, G6, 36)):
        u = np.empty((npts, Cout, Cin), np.float32)
        assert fn(w.ctypes.data, Cout, Cin, u.ctypes.data) == 0
        exp = np.einsum("xk,oklc,nl->xnoc", Gh, w.astype(np.float64), Gw).reshape(npts, Cout, Cin)
        assert np.abs(u.astype(np.float64) - exp).max() <= 1.2e-7 * np.abs(exp).max()
    assert L.msocr_winograd42_weights_host(None, Cout, Cin, None) != 0


def test_bench_live_pmc_digest_with_a_stub_profiler(tmp_path, monkeypatch):
    """bench.live_pmc_traffic: the three `rocprofv3 --kernel-trace --pmc ...` child passes are parsed as MI355X_MICROARCH.md
    prescribes (2 x FETCH_SIZE + WRITE_SIZE KiB per step, conv-stage kernels only; MFMA busy / (active / 8 x 1024)).  The profiler is a
    stub script here (no GPU): it writes the counter CSV rocprofv3 would write."""
    import argparse
    import importlib.util
    import stat
    stub = tmp_path / "rocprofv3"
    stub.write_text(r'''#!/bin/bash
out=""; pmc=""
while [ $# -gt 0 ]; do
  case "$1" in
    -d) out="$2"; shift 2;;
    --pmc) shift; while [ $# -gt 0 ] && [[ "$1" != --* ]]; do pmc="$pmc $1"; shift; done;;
    --) break;;
    *) shift;;
  esac
done
mkdir -p "$out/host"
f="$out/host/1_counter_collection.csv"
echo '"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"' > "$f"
row() { echo "1,1,1,1,1,1,1,1,\"$1\",256,0,0,0,0,0,\"$2\",$3,$4,$5" >> "$f"; }
for c in $pmc; do
  case "$c" in
    FETCH_SIZE) row "void conv_igemm_kernel<float, 128>(ConvParams)" FETCH_SIZE 4000 0 10; row "wino42_input_kernel(float const*)" FETCH_SIZE 1000 0 10; row "other_kernel(int)" FETCH_SIZE 99999 0 10;;
    WRITE_SIZE) row "void conv_igemm_kernel<float, 128>(ConvParams)" WRITE_SIZE 2000 0 10; row "wino42_input_kernel(float const*)" WRITE_SIZE 3000 0 10;;
    SQ_VALU_MFMA_BUSY_CYCLES) row "void conv_igemm_kernel<float, 128>(ConvParams)" SQ_VALU_MFMA_BUSY_CYCLES 512000 0 1000; row "wino42_input_kernel(float const*)" SQ_VALU_MFMA_BUSY_CYCLES 0 0 1000;;
    GRBM_GUI_ACTIVE) row "void conv_igemm_kernel<float, 128>(ConvParams)" GRBM_GUI_ACTIVE 8000 0 1000; row "wino42_input_kernel(float const*)" GRBM_GUI_ACTIVE 8000 2000 3000;;
  esac
done
if [ -z "$MSOCR_TEST_NO_STEPS" ]; then echo "[bench] steps_executed 4" >&2; fi
exit 0
''')
    stub.chmod(stub.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", f"{tmp_path}:{os.environ['PATH']}")
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = argparse.Namespace(workload="pipeline", precision="fp32", height=64, width=64, pages=0, target_size=0, sub_batches=0, graphs=False)
    res, note = bench.live_pmc_traffic(a)
    assert note == "ok" and res["measured_in_this_run"] is True
    per = res["per_kernel_bytes_per_step"]
    assert per["conv_igemm_kernel"] == (2 * 4000 + 2000) * 1024 / 4 and per["wino42_input_kernel"] == (2 * 1000 + 3000) * 1024 / 4
    assert res["hbm_bytes_per_step"] == sum(per.values()) and "other_kernel" not in per
    m = res["mfma_pmc"]
    assert abs(m["gemm_kernels"]["mfma_pipe_busy_fraction"] - 512000 / (8000 / 8 * 1024)) < 1e-12
    assert abs(m["conv_stage"]["mfma_pipe_busy_fraction"] - 512000 / (16000 / 8 * 1024)) < 1e-12
    # the step count comes from the child (its "[bench] steps_executed N" line), never from a constant kept by hand
    monkeypatch.setenv("MSOCR_TEST_NO_STEPS", "1")
    res2, note2 = bench.live_pmc_traffic(a)
    assert res2 is None and "step count" in note2


def test_east_loader_is_non_strict_like_the_reference():
    """EAST loads its checkpoint with strict=False (reference east.py:130-133): a ResNet-101 training checkpoint (extra
    layer3.6-22 blocks, num_batches_tracked buffers) loads into the ResNet-50 inference model, missing keys keep a fresh module's
    default initialisation, a wrong shape raises RuntimeError (as load_state_dict does even when strict=False)."""
    import torch
    from manuscript_ocr_amd import synth
    from manuscript_ocr_amd.detectors._east.net import complete_state_dict, expected_state_shapes
    sd = synth.east_state_dict(seed=3)
    shapes = expected_state_shapes()
    assert {k for k in sd if not k.endswith("num_batches_tracked")} == set(shapes) and all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    full, missing, unexpected = complete_state_dict(sd)
    assert not missing and not unexpected and all(full[k] is sd[k] for k in shapes)
    # ResNet-101 style: 17 extra layer3 blocks + bookkeeping buffers
    r101 = dict(sd)
    for b in range(6, 23):
        for j, (co, ci, k) in enumerate(((256, 1024, 1), (256, 256, 3), (1024, 256, 1)), start=1):
            r101[f"backbone.extractor.layer3.{b}.conv{j}.weight"] = torch.zeros(co, ci, k, k)
            for n in ("weight", "bias", "running_mean", "running_var"):
                r101[f"backbone.extractor.layer3.{b}.bn{j}.{n}"] = torch.zeros(co)
            r101[f"backbone.extractor.layer3.{b}.bn{j}.num_batches_tracked"] = torch.tensor(7)
    r101["backbone.extractor.bn1.num_batches_tracked"] = torch.tensor(7)
    full, missing, unexpected = complete_state_dict(r101)
    assert not missing and len(unexpected) == 17 * 3 * 6 and set(full) == set(shapes)
    # missing keys: BN -> identity statistics, conv -> default-init range, nothing raised
    part = {k: v for k, v in sd.items() if not k.startswith("decoder.block4.") and k != "backbone.extractor.bn1.running_var"}
    full, missing, unexpected = complete_state_dict(part)
    assert set(missing) == {k for k in shapes if k.startswith("decoder.block4.")} | {"backbone.extractor.bn1.running_var"}
    assert torch.equal(full["backbone.extractor.bn1.running_var"], torch.ones(64))
    assert torch.equal(full["decoder.block4.conv3x3.1.weight"], torch.ones(32)) and torch.equal(full["decoder.block4.conv3x3.1.bias"], torch.zeros(32))
    assert torch.equal(full["decoder.block4.conv1x1.1.running_mean"], torch.zeros(64))
    w = full["decoder.block4.conv3x3.0.weight"]
    assert w.shape == (32, 64, 3, 3) and w.abs().max() <= 1 / (64 * 9) ** 0.5 and w.std() > 0.01
    assert full["decoder.block4.conv1x1.0.bias"].abs().max() <= 1 / 384 ** 0.5
    bad = dict(sd)
    bad["output_head.geo_map.weight"] = torch.zeros(5, 32, 1, 1)
    with pytest.raises(RuntimeError, match="size mismatch for output_head.geo_map.weight"):
        complete_state_dict(bad)


def test_visualize_page_matches_the_reference_contract():
    """visualize_page (reference detectors/_east/utils.py:95-220): keyword-only style parameters with the reference's defaults,
    RGB PIL image of the page's size; quads drawn in `color` on a page darkened by dark_alpha outside the (blurred) quad mask and
    untouched deep inside it; show_order adds green centre-to-centre lines and a numbered black box per word; an empty page returns
    the input."""
    import inspect
    from PIL import Image
    from manuscript_ocr_amd import visualize_page
    from manuscript_ocr_amd.detectors._types import Block, Page, Word
    sig = inspect.signature(visualize_page)
    assert [(n, p.default) for n, p in list(sig.parameters.items())[2:]] == [
        ("show_order", False), ("color", (0, 0, 255)), ("thickness", 2), ("dark_alpha", 0.3), ("blur_ksize", 11),
        ("line_color", (0, 255, 0)), ("number_color", (255, 255, 255)), ("number_bg", (0, 0, 0))]
    assert all(p.kind is inspect.Parameter.KEYWORD_ONLY for p in list(sig.parameters.values())[2:])
    img = np.full((120, 300, 3), 200, dtype=np.uint8)
    page = Page(blocks=[Block(words=[Word(polygon=[[20.6, 20.2], [120.9, 20.0], [120.0, 80.0], [20.0, 80.0]], detection_confidence=0.9),
                                     Word(polygon=[[160.0, 30.0], [280.0, 30.0], [280.0, 90.0], [160.0, 90.0]], detection_confidence=0.8)])])
    out = visualize_page(img, page)
    assert isinstance(out, Image.Image) and out.mode == "RGB" and out.size == (300, 120)
    a = np.array(out)
    assert tuple(a[50, 70]) == (200, 200, 200)            # deep inside a quad: untouched
    assert tuple(a[110, 140]) == (140, 140, 140)          # far outside: img * (1 - 0.3), truncated to u8
    assert tuple(a[20, 70]) == (0, 0, 255) and tuple(a[50, 20]) == (0, 0, 255)   # outline, vertex coordinates truncated to int
    assert 140 < a[50, 123, 0] < 200                       # the blurred edge of the mask between the two quads
    assert np.array_equal(np.array(visualize_page(Image.fromarray(img), page)), a)
    o = np.array(visualize_page(img, page, show_order=True, color=(255, 0, 0), thickness=3))
    c1 = ((20.6 + 120.9 + 120.0 + 20.0) / 4, (20.2 + 20.0 + 80.0 + 80.0) / 4)
    assert tuple(o[int(c1[1]) + 10, int(c1[0]) + 10]) == (0, 0, 0)              # number box 24 x 24 around the centre
    assert tuple(o[55, 150]) == (0, 255, 0)                                      # the line between the two centres
    assert o[40:60, 60:80].max() > 128 and (o[40:60, 60:80] == 0).all(axis=2).mean() > 0.5   # light digits on the black box
    empty = Page(blocks=[Block(words=[])])
    pil = Image.fromarray(img)
    assert visualize_page(pil, empty) is pil and np.array_equal(np.array(visualize_page(img, empty)), img)


BENCH_WORKER = r'''
import importlib.util, json, os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["REPO"])
spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(os.environ["REPO"], "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from manuscript_ocr_amd.dist import gather_records, shard_range
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
log = []
def run_steps(k):                      # the workload stub: rank 1 is the slow rank
    log.append(("run", k))
    time.sleep(0.05 * k * (1 + 3 * rank))
    return [f"rank{rank}-page{i}" for i in range(2)]
class D:                               # torch.distributed with the calls recorded, so that their order can be asserted
    ReduceOp = dist.ReduceOp
    @staticmethod
    def barrier():
        log.append(("barrier",)); dist.barrier()
    @staticmethod
    def all_reduce(t, op=None):
        log.append(("all_reduce", str(op))); dist.all_reduce(t, op=op)
out, dt = bench.timed_region(run_steps, 3, 1, lambda: log.append(("sync",)), D, "cpu")
recs = gather_records([{"page": rank * 2 + i, "text": o} for i, o in enumerate(out)], torch.device("cpu"))
lo, hi = shard_range(7, rank, world)
r, w = os.pipe()
bench.emit(rank, {"value": 2 * 3 * world / dt, "n_gpus": world}, w)
os.close(w)
line = os.read(r, 4096).decode()
print("RESULT" + json.dumps({"rank": rank, "dt": dt, "log": log, "recs": recs, "shard": [lo, hi], "line": line}))
dist.destroy_process_group()
'''


def test_bench_timed_region_world2_gloo(tmp_path):
    """bench.py's own N > 1 control flow with 2 CPU ranks over gloo and the workload stubbed: the order
    prime -> sync -> barrier -> warm-up -> sync -> barrier -> EXACTLY K timed steps -> sync -> barrier -> all_reduce(MAX);
    both ranks end with the SAME elapsed time and it is the slow rank's; the records of both ranks reach rank 0 in page order;
    only rank 0 writes the JSON line; shards are contiguous and cover all pages."""
    import json
    script = tmp_path / "bw.py"
    script.write_text(BENCH_WORKER)
    env = dict(os.environ, REPO=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-600:] for o in outs]
    res = [json.loads([l for l in o[0].splitlines() if l.startswith("RESULT")][0][len("RESULT"):]) for o in outs]
    res.sort(key=lambda r: r["rank"])
    expected = [["run", 2], ["sync"], ["barrier"], ["run", 1], ["sync"], ["barrier"], ["run", 3], ["sync"], ["barrier"],
                ["all_reduce", "RedOpType.MAX"]]
    for r in res:
        assert r["log"] == expected, r["log"]
    assert res[0]["dt"] == res[1]["dt"] and res[0]["dt"] >= 0.05 * 3 * 4 * 0.95  # the slow rank's 0.6 s, on both
    assert [x["page"] for x in res[0]["recs"]] == [0, 1, 2, 3] and res[0]["recs"][3]["text"] == "rank1-page1"
    assert json.loads(res[0]["line"])["n_gpus"] == 2 and res[1]["line"] == ""
    assert res[0]["shard"] == [0, 4] and res[1]["shard"] == [4, 7]


def test_attn_pack_split_host_layout_and_exactness():
    """msocr_attn_pack_split_host: planes sum to the f32 weight exactly; layout [plane][k/16][column][k%16], columns padded to 32,
    gate-interleaved input reordered gate-major."""
    import torch
    from manuscript_ocr_amd import _native as nat
    L = nat.lib()
    torch.manual_seed(3)
    for N, gi in ((256, 0), (1024, 1), (194, 0)):
        wt = (torch.randn(256, N) * 0.1).contiguous()
        Np = (N + 31) // 32 * 32
        assert L.msocr_attn_pack_split_elems(N) == 3 * 256 * Np
        out = torch.full((3 * 256 * Np,), -1, dtype=torch.int16)
        assert L.msocr_attn_pack_split_host(wt.data_ptr(), N, gi, out.data_ptr()) == 0
        f = ((out.view(3, 16, Np, 16).to(torch.int32) & 0xffff) << 16).view(torch.float32).double()
        rec = (f[0] + f[1] + f[2]).permute(0, 2, 1).reshape(256, Np)
        if gi:
            rec = rec.view(256, 4, N // 4).permute(0, 2, 1).reshape(256, N)
        else:
            assert float(rec[:, N:].abs().max()) == 0.0 if Np > N else True
            rec = rec[:, :N]
        assert torch.equal(rec, wt.double())
    assert L.msocr_attn_pack_split_host(None, 256, 0, out.data_ptr()) != 0
    assert L.msocr_attn_pack_split_host(wt.data_ptr(), 6, 1, out.data_ptr()) != 0


def test_built_library_has_no_cross_dword_packed_f32_instruction(tmp_path):
    """The gfx950 code objects inside libmsocr.so hold no packed-f32 VALU instruction whose LOW result takes the HIGH dword of a
    source pair (`v_pk_*_f32 ... op_sel:[..]`): that form returned wrong lanes beside bf16 MFMAs on MI355X (csrc/Makefile,
    DESIGN.md section 4, profiles/r03_attn_packed_probe.txt), so the kernels that had it are compiled without packed-f32 ops.
    This looks at the artifact the tests and the bench actually load."""
    import re
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "manuscript_ocr_amd", "libmsocr.so")
    if not os.path.exists(objdump) or not os.path.exists(lib):
        pytest.skip("llvm-objdump or the built library is missing")
    shutil.copy(lib, tmp_path / "libmsocr.so")
    subprocess.run([objdump, "--offloading", "libmsocr.so"], cwd=tmp_path, check=True, capture_output=True)
    objs = sorted(p for p in os.listdir(tmp_path) if p.endswith("gfx950"))
    assert objs, "no gfx950 code object found in libmsocr.so"
    packed, bad = 0, []
    for o in objs:
        asm = subprocess.run([objdump, "-d", o], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        for line in asm.splitlines():
            if re.search(r"\bv_pk_[a-z]+_f32\b", line):
                packed += 1
                if "op_sel:[" in line:
                    bad.append(line.strip())
    assert not bad, f"{len(bad)} cross-dword packed-f32 instructions, e.g. {bad[:3]}"
    assert packed > 0  # conv_split / conv_igemm keep their same-dword packed forms: the scan really saw device code


def test_new_recurrent_entry_points_reject_bad_arguments_before_any_launch():
    """msocr_bilstm_recurrent_split / msocr_attn_beam_hoisted validate their arguments first (no GPU is touched for these calls):
    hidden sizes other than 256, null or misaligned packed planes, an xproj too large for 32-bit offsets, a split-weight struct with a
    missing plane."""
    import ctypes
    from manuscript_ocr_amd import _native as nat
    L = nat.lib()
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    assert L.msocr_bilstm_recurrent_split(p, p16, 4, 13, 128, p, None) != 0          # hidden size
    assert L.msocr_bilstm_recurrent_split(p, None, 4, 13, 256, p, None) != 0         # no planes
    assert L.msocr_bilstm_recurrent_split(p, p16 + 2, 4, 13, 256, p, None) != 0      # misaligned planes
    assert L.msocr_bilstm_recurrent_split(p, p16, 1 << 20, 4, 256, p, None) != 0     # 2^20 x 4 x 2 x 1024 elements >= 2^32
    aw = nat.AttnWeights()
    for name, _ in aw._fields_:
        setattr(aw, name, p)
    sw = nat.AttnSplitWeights()
    sw.h2h_p, sw.whh_p, sw.gen_p = p16, None, p16
    args = (4, 13, 256, 194, 25, 8, None, 1.0, 1, 2, -1, p, p16, None, None, None, None)
    assert L.msocr_attn_beam_hoisted(p, p, p16, ctypes.byref(aw), ctypes.byref(sw), *args) != 0   # a plane is missing
    assert L.msocr_attn_beam_hoisted(p, p, None, ctypes.byref(aw), None, *args) != 0              # no hoisted product
