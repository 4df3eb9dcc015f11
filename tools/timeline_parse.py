"""Per-iteration timeline from a rocprofv3 --kernel-trace CSV: duration of every kernel of
the iteration loop and the gaps between consecutive dispatches (end of one -> start of the
next), from the dispatch packets' own begin/end time stamps -- no HIP events in the stream.
Usage: python tools/timeline_parse.py <dir or kernel_trace.csv> [label]"""
import csv
import glob
import os
import statistics
import sys


def short(name):
    for key in ("stress_grad_kernel", "reduce_sliced_kernel", "reduce_kernel", "row_owner_kernel", "apply_kernel",
                "peer_receive_kernel", "iterate_kernel", "sweep_reduce_kernel"):
        if key in name:
            return key
    return name.split("(")[0][:40]


def main():
    path = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else path
    files = [path] if path.endswith(".csv") else glob.glob(
        os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                         int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"])))
    rows.sort()
    # segments: maximal runs of loop kernels; a marker kernel (copyBuffer etc.) ends one
    loop = ("stress_grad_kernel", "reduce_sliced_kernel", "reduce_kernel", "row_owner_kernel", "apply_kernel",
            "peer_receive_kernel", "iterate_kernel", "sweep_reduce_kernel")
    seg, segs = [], []
    for r in rows:
        if r[2] in loop:
            seg.append(r)
        else:
            if len(seg) >= 40:
                segs.append(seg)
            seg = []
    if len(seg) >= 40:
        segs.append(seg)
    print("== %s: %d dispatches, %d loop segments" % (label, len(rows), len(segs)))
    for seg in segs:
        # drop the first quarter (clocks settling)
        seg = seg[len(seg) // 4:]
        dur, gap = {}, {}
        for i, (a, b, k, g) in enumerate(seg):
            key = "%s[%d]" % (k, g)
            dur.setdefault(key, []).append((b - a) / 1e3)
            if i + 1 < len(seg):
                nk = "%s[%d]" % (seg[i + 1][2], seg[i + 1][3])
                gap.setdefault(key + " -> " + nk, []).append((seg[i + 1][0] - b) / 1e3)
        span = (seg[-1][1] - seg[0][0]) / 1e3
        n_sweeps = sum(1 for r in seg if r[2] in ("stress_grad_kernel", "row_owner_kernel",
                                                   "iterate_kernel", "sweep_reduce_kernel"))
        print("  segment of %d dispatches, %.2f us per sweep launch" % (len(seg), span / max(n_sweeps, 1)))
        for k, v in dur.items():
            print("    dur %-46s n=%4d  med %8.2f  mean %8.2f  min %8.2f  max %8.2f us" % (
                k, len(v), statistics.median(v), statistics.mean(v), min(v), max(v)))
        for k, v in gap.items():
            print("    gap %-70s n=%4d  med %6.2f  mean %6.2f us" % (k, len(v), statistics.median(v),
                                                                   statistics.mean(v)))


if __name__ == "__main__":
    main()
