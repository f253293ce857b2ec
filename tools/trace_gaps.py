"""Where the device is idle or runs no convolution kernel: digest of a rocprofv3 --kernel-trace CSV (dev tool, CPU).

    python tools/trace_gaps.py gpurun_out/<run>/<host>/<pid>_kernel_trace.csv [first_step last_step total_steps]

Takes the dispatches of steps [first_step, last_step) out of total_steps equally long (in dispatch count) steps — the default
bench runs 2 priming + 2 warm-up + 5 timed + 6 instrumented + 6 serialized steps = 21, so "4 9 21" is the timed region — merges
the kernel intervals and prints:
busy fraction, time with no conv-stage kernel running, and the largest idle gaps with the kernels that end before / start after
them."""
import csv
import sys
from collections import defaultdict

CONV = ("conv_igemm_kernel", "wino_input_kernel", "wino_output_kernel", "wino42_input_kernel", "wino42_output_kernel", "wino44_input_kernel", "wino44_output_kernel", "wino42_fused64_kernel", "wino_gemm4_kernel", "wino_rows_in_kernel", "wino_rows_out_kernel")


def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def main():
    path = sys.argv[1]
    a, b, tot = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (4, 9, 21)
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Dispatch_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    n = len(rows)
    rows = sorted((s, e, k) for _, s, e, k in rows[n * a // tot: n * b // tot])
    span = (max(r[1] for r in rows) - rows[0][0]) / 1e6
    allu = union([(s, e) for s, e, _ in rows])
    convu = union([(s, e) for s, e, n in rows if n.startswith(CONV)])
    busy = sum(e - s for s, e in allu) / 1e6
    convb = sum(e - s for s, e in convu) / 1e6
    print(f"window {span:.1f} ms: some kernel running {busy:.1f} ms ({busy / span:.3f}), a conv-stage kernel running {convb:.1f} ms ({convb / span:.3f})")
    # what runs while no conv kernel runs
    other = defaultdict(float)
    ci = 0
    for s, e, n in rows:
        if n.startswith(CONV):
            continue
        # portion of [s, e) outside convu
        cur = s
        for cs, ce in convu:
            if ce <= cur:
                continue
            if cs >= e:
                break
            if cs > cur:
                other[n] += min(cs, e) - cur
            cur = max(cur, ce)
            if cur >= e:
                break
        if cur < e:
            other[n] += e - cur
    print("kernel time outside any conv-stage kernel (ms, may overlap each other):")
    for n, v in sorted(other.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {v / 1e6:8.2f}  {n}")
    gaps = []
    for (s0, e0), (s1, e1) in zip(allu, allu[1:]):
        gaps.append((s1 - e0, e0, s1))
    gaps.sort(reverse=True)
    print(f"idle gaps: {len(gaps)} totalling {sum(g[0] for g in gaps) / 1e6:.1f} ms; the largest:")
    for g, e0, s1 in gaps[:12]:
        before = [n for s, e, n in rows if e == e0][:1]
        after = [n for s, e, n in rows if s == s1][:1]
        print(f"  {g / 1e3:8.1f} us  after {before}  before {after}")


if __name__ == "__main__":
    main()
