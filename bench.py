#!/usr/bin/env python3
"""bench.py — manuscript pages/s on MI355X for the EAST + TRBA hot path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic pages already resident in HBM
(u8 pages + injected score/geo maps): Pipeline.predict_batch = EAST ResNet-50
forward -> quad decode -> locality-aware NMS -> box filters -> reading order -> device crop +
ResizeAndPadA -> TRBA SE-ResNet31 + BiLSTM + beam-8 attention decode -> text (workload
"pipeline", BASELINE configs[3]); "--workload east" stops after the detector (configs[1]).
Pages shard across ranks with no data-path collective (weak scaling: fixed per-GPU batch); the
only exchange is the final gather of decoded records (all_gather of a padded u8 buffer over
RCCL).  Rank 0 prints ONE JSON line: whole-job pages/s, the live HIP-event roofline of the
dominant kernel (implicit-GEMM convolution on MFMA) and a bounded CPU baseline (the oracle =
CPU restatement of the reference path, timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0}  # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
TRBA_CFG = {"img_h": 32, "img_w": 100, "max_len": 25, "hidden_size": 256}
STEP_TIMES = [] if os.environ.get("MSOCR_STEP_TIMES") else None  # diagnostics: host time after every collected step
TIE_TOL = 5e-3  # first-step logit gap (|logit| ~ 5) treated as a tie between two f32 implementations (tests/conftest.py)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="pipeline", choices=["east", "pipeline"])
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="conv storage/MFMA input type; fp32 = parity mode (text identical to the CPU reference)")
    ap.add_argument("--pages", type=int, default=0, help="pages per step per GPU (default 16 pipeline / 8 east)")
    ap.add_argument("--height", type=int, default=1536)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--target-size", type=int, default=0,
                    help="reference-faithful detector geometry: resize every page to T x T on the device (EAST.predict default T=1280, "
                         "infer.py:304) instead of feeding the page at its native size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-overlap-steps", action="store_true", help="do not enqueue step i+1's detector work before collecting step i")
    ap.add_argument("--sub-batches", type=int, default=0, help="sub-batch groups pipelined on separate streams (0 = auto)")
    ap.add_argument("--graphs", action="store_true",
                    help="replay the detector's launch sequence from a hipGraph (measured 2 % slower than plain launches, DESIGN.md 7)")
    ap.add_argument("--serialize-streams", action="store_true",
                    help="run the sub-batch pipeline on ONE stream (no cross-stream kernel overlap): per-kernel profiling mode")
    return ap.parse_args()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local)
    # Native libraries (RCCL prints a version banner) write to the process's stdout; the contract is ONE JSON line there.
    # Keep the real stdout for that line and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    dist = None
    if world > 1 or os.environ.get("MSOCR_FORCE_DIST"):  # MSOCR_FORCE_DIST: exercise the RCCL path with a single rank
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        # every lazily created piece of the communicator (first barrier, first all_reduce) before anything is timed
        warm = torch.zeros(1, dtype=torch.float64, device="cuda")
        for _ in range(2):
            dist.barrier()
            dist.all_reduce(warm, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()

    from manuscript_ocr_amd import Pipeline, ops, synth
    from manuscript_ocr_amd.detectors import EAST
    from manuscript_ocr_amd.detectors._east.net import east_conv_macs
    from manuscript_ocr_amd.dist import gather_records, page_records
    from manuscript_ocr_amd.recognizers import TRBA
    from manuscript_ocr_amd.recognizers._trba.net import trba_cnn_macs

    H, W = a.height, a.width
    TW, TH = (a.target_size, a.target_size) if a.target_size else (W, H)  # network input (width, height)
    NP = a.pages or (16 if a.workload == "pipeline" else 8)
    esd = synth.east_state_dict(seed=20260128)
    tsd = synth.trba_state_dict_confident(194, 256, seed=20260128)
    det = EAST(state_dict=esd, target_size=(TW, TH), device="cuda", precision=a.precision, use_graphs=a.graphs)
    rec = TRBA(state_dict=tsd, config=TRBA_CFG, device="cuda", precision=a.precision) if a.workload == "pipeline" else None
    pipe = Pipeline(detector=det, recognizer=rec) if rec is not None else None
    if rec is not None and os.environ.get("MSOCR_DEVICE_BATCH"):
        rec.device_batch = int(os.environ["MSOCR_DEVICE_BATCH"])
    if pipe is not None and os.environ.get("MSOCR_UPLOAD_ON_REC"):
        pipe.upload_on_det_stream = False
    if pipe is not None and a.serialize_streams:
        pipe.serialize_streams = True

    # synthetic pages + injected maps of THIS rank's shard (global page id = rank*NP + i)
    pages, scores, geos = [], [], []
    for i in range(NP):
        seed = 200 + rank * NP + i
        pg, rects = synth.synth_page(seed, H, W)
        s, g = synth.synth_maps(rects, (H, W), (TH // 4, TW // 4), seed)
        pages.append(pg), scores.append(s), geos.append(g)
    pages_dev = torch.from_numpy(np.stack(pages)).cuda()
    maps_dev = (torch.from_numpy(np.stack(scores)).cuda(), torch.from_numpy(np.stack(geos)).cuda())

    def submit():
        return pipe.submit_batch(pages, pages_dev=pages_dev, sub_batches=a.sub_batches, _maps_override=maps_dev)

    def step():
        if pipe is not None:
            return pipe.collect_batch(submit())
        return [r["page"] for r in det.predict_batch(pages, _pages_dev=pages_dev, _maps_override=maps_dev)]

    # detector-only workload (BASELINE configs[1]): groups of 2 pages on their own streams, two stream sets, the groups of step
    # i+1 enqueued before step i's boxes are read back and filtered on the host
    east_bounds = [(lo, min(lo + 2, NP)) for lo in range(0, NP, 2)]
    east_streams = [[torch.cuda.Stream() for _ in east_bounds] for _ in range(2)] if pipe is None else None

    def east_submit(i):
        main = torch.cuda.current_stream()
        hs = []
        for st, (lo, hi) in zip(east_streams[i & 1], east_bounds):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                hs.append((st, lo, hi, det.detect_start(pages_dev[lo:hi], (maps_dev[0][lo:hi], maps_dev[1][lo:hi]))))
        return hs

    def east_collect(hs):
        main, res = torch.cuda.current_stream(), []
        for st, lo, hi, h in hs:
            with torch.cuda.stream(st):
                res += [r["page"] for r in det.detect_finish(h, pages[lo:hi])]
            main.wait_stream(st)
        return res

    def run_steps_east(k):
        h, out_ = east_submit(0), None
        for i in range(k):
            h_next = east_submit(i + 1) if i + 1 < k else None
            out_ = east_collect(h)
            h = h_next
        return out_

    def run_steps(k):
        """k steps, software-pipelined across steps: the detector work of step i+1 is enqueued and its host stage
        (box filters, reading order, crop descriptors -> recogniser enqueue) runs BEFORE step i is collected, so the
        device always holds queued recogniser work while the host annotates step i.  All work of the k steps is inside."""
        if pipe is None and not (a.serialize_streams or a.no_overlap_steps):
            return run_steps_east(k)
        if pipe is None or a.serialize_streams or a.no_overlap_steps:
            out_ = None
            for _ in range(k):
                out_ = step()
            return out_
        depth = int(os.environ.get("MSOCR_PIPE_DEPTH", "1"))  # batches advanced ahead of the one being collected
        pipe.stream_sets = depth + 1
        queue, out_, nsub = [], None, 0
        for i in range(k):
            while nsub < k and len(queue) <= depth:
                queue.append(pipe.advance_batch(submit()))
                nsub += 1
            out_ = pipe.collect_batch(queue.pop(0))
            if STEP_TIMES is not None:
                STEP_TIMES.append(time.perf_counter())
        return out_

    def barrier():
        if dist is not None:
            dist.barrier()

    # Setup, untimed and independent of --warmup: two priming steps (both alternating stream sets, allocator pools, lazily
    # loaded code objects) followed by the same synchronize + barrier sequence that brackets the timed region, so that
    # whatever the runtime or RCCL initialise on first use is initialised before the W warm-up steps even when W = 0.
    run_steps(2)
    torch.cuda.synchronize()
    barrier()
    out = None
    out = run_steps(a.warmup) if a.warmup else None
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    out = run_steps(a.steps)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the path's one real exchange: decoded strings/boxes of every rank gathered (RCCL when world > 1)
    records = gather_records([r for i, p in enumerate(out) for r in page_records(rank * NP + i, p)], torch.device("cuda", local))

    total_pages = NP * a.steps * world
    words = [w for p in out for b in p.blocks for w in b.words]
    n_crops = sum(1 for w in words if w.text is not None)
    gflop_page = 2 * east_conv_macs(TH, TW) / 1e9
    res = {
        "metric": "manuscript pages/sec end-to-end (EAST+TRBA) at 1/2/4/8 MI355X; CER vs CPU ref",
        "value": total_pages / dt,
        "unit": "pages/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"fp32": "f32", "bf16": "bf16"}[a.precision],
        "data": "synthetic",
        "config": {
            "workload": (f"full EAST->crop->TRBA pipeline (BASELINE configs[3]): batch={NP} pages @ {W}x{H} per GPU, "
                         + (f"native network input {H}x{W}" if not a.target_size else f"pages resized to {TW}x{TH} on the device") +
                         f"; EAST forward + decode + LANMS + filters + reading order + device crop/ResizeAndPadA + "
                         "TRBA 32x100 beam-8; decode/NMS on injected synthetic maps (random weights give unusable maps)")
            if a.workload == "pipeline" else
            (f"EAST detector only (BASELINE configs[1]): batch={NP} pages @ {W}x{H} per GPU, network input {TH}x{TW}; "
             "forward + decode + LANMS + filters; decode/NMS on injected synthetic maps"),
            "pages_per_step_per_gpu": NP,
            "page_hw": [H, W],
            "gflop_per_page_east": gflop_page,
            "gflop_per_crop_trba_cnn": 2 * trba_cnn_macs(32, 100) / 1e9,
            "words_per_page": len(words) / NP,
            "crops_per_page": n_crops / NP,
            "weights": "seeded synthetic (no checkpoint offline)",
            "trba_decode": (None if rec is None or not getattr(rec, "last_rows", 0) else
                            {"max_len": TRBA_CFG["max_len"], "mean_chunk_run_length": round(rec.last_run_length_sum / rec.last_rows, 2),
                             "note": "the beam kernel leaves the step loop at each 32-crop chunk's run length, like the reference "
                                     "(model.py:215); 1.8 ms per 960 crops here, 4.2 ms when all 25 steps run"}),
            "parallelism": f"pages sharded over {world} rank(s), no data-path collective; final all_gather of {len(records)} records",
        },
    }

    if pipe is not None:
        res["host_stage_s_last_step"] = {k: round(v, 4) for k, v in pipe.last_profile.items()}
        res["east_stage_s_last_step"] = {k: round(v, 4) for k, v in det.last_profile.items()}
    if rank == 0 and not a.no_roofline:
        # Live HIP-event timing of every implicit-GEMM launch (events on the launch stream = torch's current stream).
        peak = PEAK_TFLOPS[a.precision]

        def instrumented(serialize):
            if pipe is not None:
                pipe.serialize_streams = serialize
            step()  # settle allocator pools of this mode
            torch.cuda.synchronize()
            ref = torch.cuda.Event(enable_timing=True)
            ref.record()
            ops.PROFILE = []
            run_steps(a.steps)
            torch.cuda.synchronize()
            prof, ops.PROFILE = ops.PROFILE, None
            if pipe is not None:
                pipe.serialize_streams = a.serialize_streams
            iv = sorted((ref.elapsed_time(e0), ref.elapsed_time(e1)) for e0, e1, _, _ in prof)
            fl = float(sum(f for _, _, f, _ in prof))
            instrumented.executed = float(sum(t[4] for _, _, _, t in prof))
            dur = np.array([e - s for s, e in iv])
            union, cs, ce = 0.0, iv[0][0], iv[0][1]
            for s_, e_ in iv[1:]:
                if s_ > ce:
                    union += ce - cs
                    cs, ce = s_, e_
                else:
                    ce = max(ce, e_)
            union += ce - cs
            return prof, fl, dur, union

        # (1) the timed configuration: sub-batch streams overlap, so a launch's event-to-event time includes the other
        #     streams' kernels sharing the chip.  Chip-level rate = FLOP / union of the intervals in which >= 1 conv runs.
        prof, fl, dur, union = instrumented(a.serialize_streams)
        tf = fl / (union * 1e-3) / 1e12
        res["roofline"] = {
            "kernel": "convolution stage = every msocr_conv2d / msocr_conv3x3_winograd call of a step (conv_igemm_kernel, plus the "
                      "two Winograd transform kernels around its 16-GEMM launch); achieved = ALGORITHMIC direct-convolution FLOP "
                      "(2*MACs, SURVEY.md 8d) / time in which at least one such call is executing, sub-batch streams overlapping as "
                      "in the timed region.  Winograd F(2x2,3x3) executes 2.25x fewer matrix FLOPs than the algorithmic count on "
                      "the 3x3/1/1 layers, so `achieved` may approach or pass the MFMA peak; `executed_mfma_tflops` is what the "
                      "matrix pipes really run",
            "bound": "mfma",
            "achieved": tf,
            "executed_mfma_tflops": instrumented.executed / (union * 1e-3) / 1e12,
            "executed_frac_of_peak": instrumented.executed / (union * 1e-3) / 1e12 / peak,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": tf / peak,
            "traffic": pmc_traffic() if (a.workload == "pipeline" and a.precision == "fp32" and not a.pages and not a.sub_batches) else None,
            "launches_per_step": len(prof) // a.steps,
            "avg_launch_ms": float(dur.mean()),
            "conv_busy_ms_per_step": float(union / a.steps),
            "alg_gflop_per_step": fl / a.steps / 1e9,
        }
        # (2) the same launches on ONE stream: isolated per-launch durations (FLOP-weighted rate of a launch running alone)
        if pipe is not None and not a.serialize_streams:
            prof2, fl2, dur2, _ = instrumented(True)
            tf2 = fl2 / (dur2.sum() * 1e-3) / 1e12
            res["roofline"]["isolated"] = {"achieved": tf2, "frac": tf2 / peak, "avg_launch_ms": float(dur2.mean()),
                                           "conv_ms_per_step": float(dur2.sum() / a.steps)}
        if os.environ.get("MSOCR_DUMP_CONV"):
            agg = {}
            for (e0, e1, f, tag) in prof:
                a_ = agg.setdefault(tag[:4], [0, 0.0, 0.0])
                a_[0] += 1
                a_[1] += e0.elapsed_time(e1)
                a_[2] += f
            with open(os.environ["MSOCR_DUMP_CONV"], "w") as fh:
                for tag, (cnt, m, f) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    fh.write(f"M={tag[0]} N={tag[1]} K={tag[2]} {tag[3]} calls={cnt} ms={m:.3f} TF/s={f / (m * 1e-3) / 1e12:.1f}\n")

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(a.workload, esd, tsd, pages, scores, geos, H, W, out, target_wh=(TW, TH))

    if STEP_TIMES:
        print("step end times (s):", [round(t - STEP_TIMES[0], 3) for t in STEP_TIMES], file=sys.stderr)
    if rank == 0:
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


def pmc_traffic():
    """HBM bytes per step of the convolution stage (conv_igemm_kernel + Winograd transforms) from the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate
    rocprofv3 --pmc passes of this same command; bench.py cannot run the profiler on itself, so the committed
    measurement is reported, or null when absent / not for this workload)."""
    path = os.path.join(ROOT, "profiles", "r01c_pmc_traffic.json")
    try:
        with open(path) as f:
            return float(json.load(f)["hbm_bytes_per_step"])
    except Exception:
        return None


def cpu_baseline(workload, esd, tsd, pages, scores, geos, H, W, gpu_pages, budget_s=12.0, target_wh=None):
    """The oracle (CPU restatement of the reference path) on a bounded sample of the same workload; also the
    CER of the GPU text against this CPU text on the sampled pages."""
    from oracle import east_model as oem
    from oracle import east_post as P
    from oracle import imgproc
    from oracle import lanms as L
    from oracle import pipeline_glue as G
    from oracle import trba_model as otm

    TW, TH = target_wh if target_wh else (W, H)
    net = oem.EASTNet()
    net.load_state_dict(esd)
    net.eval()
    tnet = otm.TRBANet(194, 256)
    tnet.load_state_dict(tsd)
    tnet.eval()
    itos, _ = otm.load_charset(os.path.join(ROOT, "manuscript_ocr_amd", "recognizers", "_trba", "configs", "charset.txt"))
    L.lib()
    n, t0 = 0, time.perf_counter()
    edits = chars = mism = n_words = n_diff = n_tie = 0
    with torch.no_grad():
        for pi, (pg, s, g) in enumerate(zip(pages, scores, geos)):
            net(torch.from_numpy(imgproc.east_preprocess(pg, TW, TH)))
            quads = P.east_postprocess(s, g, (H, W), (TW, TH), L.locality_aware_nms)
            if workload == "pipeline":
                polys = [q[:8].reshape(4, 2).tolist() for q in quads]
                order, kept, crops = G.order_and_crop(polys, pg, 5)
                res = []
                for c0 in range(0, len(crops), 32):
                    x = torch.from_numpy(np.stack([imgproc.trba_preprocess(c, 32, 100) for c in crops[c0:c0 + 32]]))
                    lg, ids = tnet(x, max_len=25, mode="beam", beam_size=8, alpha=0.9, temperature=1.7)
                    res += otm.texts_and_confidences(lg, ids, itos, 0, 2, None)
                gw = [w for w in gpu_pages[pi].blocks[0].words]
                gtexts = [gw[pos].text for pos in kept] if len(gw) == len(order) else []
                mism += int(len(gw) != len(order))
                for r, hyp in zip(res, gtexts):
                    ref_t, hyp = r["text"], hyp or ""
                    edits += _lev(ref_t, hyp)
                    chars += max(len(ref_t), 1)
                    n_words += 1
                    if ref_t != hyp:
                        # a word may differ only where the CPU path's own first-character arg-max is a rounding-level tie
                        n_diff += 1
                        n_tie += int(hyp[:1] != ref_t[:1] and otm.first_token_margin(r["logits0"], itos, 2, hyp, ref_t) < TIE_TOL)
            n += 1
            if time.perf_counter() - t0 > budget_s:
                break
    el = time.perf_counter() - t0
    return {
        "value": n / el,
        "unit": "pages/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{n} page(s) @ {W}x{H} of the same synthetic workload: oracle torch-CPU fp32 EAST forward + C LANMS + NumPy "
                  f"filters" + (" + reading order + crops + torch-CPU TRBA beam-8" if workload == "pipeline" else "") + f" ({el:.1f} s)",
        "cer_gpu_vs_cpu": (edits / chars) if chars else None,
        "words_compared": n_words,
        "words_differing": n_diff,
        "words_differing_at_cpu_near_tie": n_tie,  # first-character margin < TIE_TOL in the CPU path's own logits
        "box_count_mismatch_pages": mism,
    }


def _lev(a, b):
    if a == b:
        return 0
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


if __name__ == "__main__":
    main()
