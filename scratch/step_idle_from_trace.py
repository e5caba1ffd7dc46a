#!/usr/bin/env python3
"""Chip-idle time of a training step from a rocprofv3 kernel trace (no GPU needed).

    python scratch/step_idle_from_trace.py gpurun_out/prof_r04final/r04final_kernel_trace.csv [first_step last_step]

A step starts at `pack_conv3x3_multi_kernel` (the first launch of a forward).  Per step: wall span, time with NO kernel running on
either queue (split at the first weight-gradient launch into forward / backward), the kernel time of the split-K reducers, the sum of all
kernel durations over both queues, launches.  Used for profiles/r04_step_idle.txt.
"""
import csv
import statistics
import sys


def main():
    path = sys.argv[1]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "pack_conv3x3_multi" in r["Kernel_Name"]]
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 3            # bench.py default: 3 warm-up steps, then 10 timed
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 13
    out = []
    for si in range(lo, hi):
        rs = rows[starts[si]:starts[si + 1]]
        t0 = int(rs[0]["Start_Timestamp"])
        iv = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"]) for r in rs)
        span = int(rows[starts[si + 1]]["Start_Timestamp"]) - t0
        bwd0 = min(a for a, b, n in iv if "wgrad_v2" in n)
        idle_f = idle_b = 0
        last = 0
        for a, b, n in iv:
            if a > last:
                if a < bwd0:
                    idle_f += a - last
                else:
                    idle_b += a - last
            last = max(last, b)
        idle_b += max(0, span - last)
        red = sum(b - a for a, b, n in iv if "wgrad_reduce" in n)
        out.append((span / 1e3, idle_f / 1e3, idle_b / 1e3, red / 1e3, sum(b - a for a, b, n in iv) / 1e3, len(iv)))
    print("step   wall_us  idle_fwd_us  idle_bwd_us  reducer_kernel_us  kernel_time_sum_us  launches")
    for k, r in enumerate(out):
        print(f"{k:4d} {r[0]:9.1f} {r[1]:12.1f} {r[2]:12.1f} {r[3]:18.1f} {r[4]:19.1f} {r[5]:9d}")
    med = [statistics.median(c) for c in zip(*out)]
    print(f" med {med[0]:9.1f} {med[1]:12.1f} {med[2]:12.1f} {med[3]:18.1f} {med[4]:19.1f} {int(med[5]):9d}")


if __name__ == "__main__":
    main()
