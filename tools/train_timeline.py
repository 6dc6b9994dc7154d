"""Dev tool: timeline of ONE training step from a rocprofv3 --kernel-trace CSV (steps are delimited by the fused Adam
launch): per window of `bin_ms` the share of time with >= 1 kernel running, the summed kernel time (concurrency) and the
kernels that account for most of it.   usage: python tools/train_timeline.py <kernel_trace.csv> [bin_ms] [step_index]"""
import csv
import sys
from collections import defaultdict


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
    return n.split("(")[0][:48] if not n.startswith("Cijk") else "Cijk(GEMM)"


def main(path, bin_ms="1.0", step="-1"):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    adam = [(s, e) for s, e, n in rows if "FusedAdam" in n]  # several launches per step: the last of each burst
    ends = [e for i, (s, e) in enumerate(adam) if i + 1 == len(adam) or adam[i + 1][0] - e > 2_000_000]
    k = int(step) % len(ends)
    t0, t1 = (ends[k - 1] if k > 0 else rows[0][0]), ends[k]
    ks = [(s, e, n) for s, e, n in rows if s >= t0 and e <= t1]
    print(f"step {k}: {len(ks)} launches, wall {(t1 - t0) / 1e6:.2f} ms, kernel time {sum(e - s for s, e, _ in ks) / 1e6:.2f} ms")
    b = float(bin_ms) * 1e6
    nb = int((t1 - t0) / b) + 1
    busy = [0.0] * nb
    ksum = [0.0] * nb
    who = [defaultdict(float) for _ in range(nb)]
    # union-busy: sweep
    cur_end = t0
    for s, e, n in ks:
        # kernel time per bin
        x = s
        while x < e:
            i = int((x - t0) / b)
            nx = min(e, t0 + (i + 1) * b)
            ksum[i] += nx - x
            who[i][short(n)] += nx - x
            x = nx
        us, ue = max(s, cur_end), e
        if ue > us:
            x = us
            while x < ue:
                i = int((x - t0) / b)
                nx = min(ue, t0 + (i + 1) * b)
                busy[i] += nx - x
                x = nx
            cur_end = ue
    for i in range(nb):
        top = sorted(who[i].items(), key=lambda kv: -kv[1])[:3]
        print(f"{i * float(bin_ms):6.1f} ms  busy {busy[i] / b:4.2f}  kernels {ksum[i] / b:5.2f}x  " +
              ", ".join(f"{n} {t / 1e3:.0f}us" for n, t in top))


if __name__ == "__main__":
    main(*sys.argv[1:])
