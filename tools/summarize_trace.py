"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid size) call count and average duration.
Separates the level-1 sparse-conv launches (grid = 142116/16 groups) from the other levels that share the
same template instantiation.   usage: python tools/summarize_trace.py <kernel_trace.csv> [substr]"""
import csv
import sys
from collections import defaultdict


def main(path, substr="k_conv_os"):
    acc = defaultdict(lambda: [0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if substr not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            key = (short, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"]))
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            acc[key][0] += 1
            acc[key][1] += d
    print("| kernel | workgroups | wg size | calls | avg us | total ms |")
    print("|---|---|---|---|---|---|")
    for (k, g, w), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {g} | {w} | {n} | {t / n / 1e3:.2f} | {t / 1e6:.3f} |")


if __name__ == "__main__":
    main(*sys.argv[1:])
