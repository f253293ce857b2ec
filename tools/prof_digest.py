"""Digest rocprofv3 outputs pulled back from a GPU box (gpurun_out/<run>/runc/*.csv) into the summaries committed under
profiles/ (dev tool; runs on the CPU).

    python tools/prof_digest.py <tag> <conc_dir> <serialized_dir> <fetch_dir> <write_dir> [<mfma_dir>]

Copies the two kernel_stats.csv files and writes profiles/<tag>_pmc_traffic.json: HBM bytes per step of the convolution
stage (conv_igemm_kernel instances + the two Winograd transform kernels), FETCH_SIZE doubled as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (wide coalesced reads are counted at 64 B per 128-B
request), WRITE_SIZE as reported; both counters in KiB, collected in separate --pmc passes of
`bench.py --no-cpu-baseline --no-roofline --steps 1 --warmup 1` (2 priming + 1 warm-up + 1 timed step -> divided by 4)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONV_STAGE = ("conv_split_kernel", "conv_split_pp_kernel", "conv_igemm_kernel", "wino_input_kernel", "wino_output_kernel", "wino42_input_kernel", "wino42_output_kernel", "wino44_input_kernel", "wino44_output_kernel", "wino42_fused64_kernel", "wino42_fused64_v2_kernel", "wino_gemm4_kernel", "wino_rows_in_kernel", "wino_rows_out_kernel")


def one(pattern):
    f = glob.glob(pattern)
    if not f:
        raise SystemExit(f"no file matches {pattern}")
    return f[0]


def pmc(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0, 0])
    for r in csv.DictReader(open(one(os.path.join(d, "runc", "*_counter_collection.csv")))):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        a = agg[n]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg


def main():
    tag, conc, ser, fetch, write = sys.argv[1:6]
    mfma = sys.argv[6] if len(sys.argv) > 6 else None
    out = os.path.join(ROOT, "profiles")
    shutil.copy(one(os.path.join(conc, "runc", "*_kernel_stats.csv")), os.path.join(out, f"{tag}_kernel_stats_concurrent.csv"))
    shutil.copy(one(os.path.join(ser, "runc", "*_kernel_stats.csv")), os.path.join(out, f"{tag}_kernel_stats_serialized.csv"))
    fe, wr = pmc(fetch, "FETCH_SIZE"), pmc(write, "WRITE_SIZE")
    steps = 4.0  # bench.py executes 2 priming + 1 warm-up + 1 timed step with --steps 1 --warmup 1 ...
    log = fetch.rstrip("/") + ".log"  # ... and says so itself: "[bench] steps_executed N" in the pass's stderr (profile_round.sh)
    if os.path.exists(log):
        for line in open(log, errors="replace"):
            if "[bench] steps_executed" in line:
                steps = float(line.split()[-1])
    per = {}
    tot = 0.0
    for n in sorted(set(fe) | set(wr)):
        if not n.startswith(CONV_STAGE):
            continue
        f, w = fe.get(n, [0, 0.0, 0]), wr.get(n, [0, 0.0, 0])
        b = (2.0 * f[1] + w[1]) * 1024.0 / steps
        per[n] = {"dispatches_per_step": f[0] / steps, "fetch_kib_raw_per_step": f[1] / steps, "write_kib_per_step": w[1] / steps,
                  "hbm_bytes_per_step": b}
        tot += b
    res = {
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py "
                   "--no-cpu-baseline --no-roofline --steps 1 --warmup 1",
        "kernels": "convolution stage = conv_split_kernel + conv_igemm_kernel (all instances: Winograd GEMMs + direct convolutions) + wino_gemm4_kernel + "
                   "the Winograd transform kernels (wino42_input / wino42_output / wino_input / wino_output / wino_rows_in / wino_rows_out)",
        "gfx950_correction": "FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads: doubled (MI355X_MICROARCH.md, HBM)",
        "hbm_bytes_per_step": tot,
        "per_kernel": per,
    }
    if mfma:
        busy = pmc(mfma, "SQ_VALU_MFMA_BUSY_CYCLES")
        gui = pmc(mfma, "GRBM_GUI_ACTIVE")
        mm = {}
        for n in busy:
            if n.startswith(("conv_split_kernel", "conv_split_pp_kernel", "conv_igemm_kernel", "wino_gemm4_kernel", "wino42_fused64_kernel", "wino42_fused64_v2_kernel")) and gui[n][1] > 0:
                # FLOP per busy cycle and SIMD: 64 for exact-f32 MFMA, 1024 for the bf16 MFMAs of the split-operand kernels
                fpc = 1024.0 if (n.startswith(("conv_split_kernel", "conv_split_pp_kernel", "wino42_fused64_v2_kernel")) or (n.startswith("wino42_fused64_kernel") and n.rstrip(">").endswith("true"))) else 64.0
                mm[n] = {"dispatches": busy[n][0], "SQ_VALU_MFMA_BUSY_CYCLES": busy[n][1], "GRBM_GUI_ACTIVE_sum_8xcd": gui[n][1],
                         "kernel_ns": gui[n][2],
                         "mfma_pipe_busy_fraction": busy[n][1] / (gui[n][1] / 8.0 * 1024.0),   # 1024 SIMDs, GUI summed over 8 XCDs
                         "clock_ghz": gui[n][1] / 8.0 / gui[n][2],
                         "executed_mfma_tflops": busy[n][1] * fpc / (gui[n][2] * 1e-9) / 1e12,
                         "f32_equivalent_tflops": busy[n][1] * (fpc / 6.0 if fpc > 64 else fpc) / (gui[n][2] * 1e-9) / 1e12}
        # the whole convolution stage (GEMM kernels AND the Winograd transforms, which run no MFMA), launches one at a time:
        # north_star's "MFMA utilisation of the conv stages"
        sb = sum(busy[n][1] for n in busy if n.startswith(CONV_STAGE))
        sg = sum(gui[n][1] for n in gui if n.startswith(CONV_STAGE))
        sn = sum(gui[n][2] for n in gui if n.startswith(CONV_STAGE))
        stage = {"mfma_pipe_busy_fraction": sb / (sg / 8.0 * 1024.0), "clock_ghz": sg / 8.0 / sn, "kernel_ns_per_step": sn / steps}
        res["mfma_pmc"] = {"conv_stage": stage, "command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py "
                                      "--no-cpu-baseline --no-roofline --serialize-streams --steps 1 --warmup 1", "per_kernel": mm}
    with open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_kernel"}, indent=1)[:1500])


if __name__ == "__main__":
    main()
