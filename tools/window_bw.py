"""HBM-side bytes per window of ONE train step: is the overlapped backward pass at the copy rate of the lease (bytes-bound) or not?

    python tools/window_bw.py <kernel_trace.csv of a plain --kernel-trace run> <dir with FETCH_SIZE_/WRITE_SIZE_counter_collection.csv> [copy_TBps]

The PMC passes serialise dispatches, so their timestamps say nothing about the overlapped step; their BYTES per dispatch do.  Every dispatch of the last
complete step of the kernel trace (two queues, real overlap) is given the bytes of the same dispatch of the PMC passes - matched by (kernel symbol, k-th
launch of that symbol within the step), FETCH_SIZE doubled for gfx950 (MI355X_MICROARCH.md) - and spread evenly over its duration; the step is cut at the
level boundaries of the backward pass (the launches of conv_stream / stemx / conv3x16 kernels mark the 96^3 level, conv3x with 1728 workgroups the 48^3
level) and into fixed 250 us slices.  Output: per window wall time, bytes of the main queue / of the weight-gradient queue, aggregate TB/s."""
import collections
import csv
import glob
import os
import sys


def steps_of(rows, key):
    idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    return rows[idx[-2] + 1: idx[-1] + 1] if len(idx) >= 2 else rows


def pmc_bytes(d):
    """{(symbol, k): bytes} for the last complete step of the PMC passes"""
    out = collections.defaultdict(float)
    for ctr, mul in (("FETCH_SIZE", 2.0 * 1024), ("WRITE_SIZE", 1024.0)):
        c = glob.glob(os.path.join(d, "**", "%s_counter_collection.csv" % ctr), recursive=True)
        if not c:
            raise SystemExit("no %s pass under %s" % (ctr, d))
        rows = [r for r in csv.DictReader(open(c[0])) if r["Counter_Name"] == ctr]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        st = steps_of(rows, None)
        seen = collections.Counter()
        for r in st:
            k = seen[r["Kernel_Name"]]
            seen[r["Kernel_Name"]] += 1
            out[(r["Kernel_Name"], k)] += float(r["Counter_Value"]) * mul
    return out


def main():
    trace, pmcdir = sys.argv[1], sys.argv[2]
    copy = float(sys.argv[3]) if len(sys.argv) > 3 else None
    rows = list(csv.DictReader(open(trace)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    step = steps_of(rows, None)
    by = pmc_bytes(pmcdir)
    # k-th launch of a symbol: in queue order (a symbol's launches live on one queue; dispatch ids grow in enqueue order)
    seen = collections.Counter()
    miss = 0
    for r in sorted(step, key=lambda r: int(r["Dispatch_Id"])):
        k = seen[r["Kernel_Name"]]
        seen[r["Kernel_Name"]] += 1
        r["bytes"] = by.get((r["Kernel_Name"], k))
        if r["bytes"] is None:
            r["bytes"] = 0.0
            miss += 1
    t0, t1 = step[0]["s"], max(r["e"] for r in step)
    mainq = collections.Counter(r["Queue_Id"] for r in step).most_common(1)[0][0]
    tot = sum(r["bytes"] for r in step)
    print("step: %d dispatches, wall %.1f us, %.2f GB (PMC, fetch x2 + write) = %.2f TB/s average; %d dispatches without a PMC match" %
          (len(step), (t1 - t0) / 1e3, tot / 1e9, tot / (t1 - t0) / 1e3, miss))

    def window(a, b):
        m = s = 0.0
        for r in step:
            lo, hi = max(a, r["s"]), min(b, r["e"])
            if hi > lo and r["e"] > r["s"]:
                v = r["bytes"] * (hi - lo) / (r["e"] - r["s"])
                if r["Queue_Id"] == mainq:
                    m += v
                else:
                    s += v
        return m, s

    def line(name, a, b):
        m, s = window(a, b)
        us = (b - a) / 1e3
        tb = (m + s) / (b - a) / 1e3
        extra = "  = %.2f of the copy rate" % (tb / copy) if copy else ""
        print("%-34s %8.1f us  main %7.1f MB  weight-gradient queue %7.1f MB  -> %5.2f TB/s%s" % (name, us, m / 1e6, s / 1e6, tb, extra))

    # phase boundaries from the main queue's own landmarks
    def first(pred, after=0):
        for r in step:
            if r["Queue_Id"] == mainq and r["s"] >= after and pred(r):
                return r
        return None
    loss = first(lambda r: "loss_reduce" in r["Kernel_Name"])
    tl = loss["s"] if loss else t0
    line("forward pass", t0, tl)
    # backward: finest level ends where the first 48^3-level halo conv (1728 workgroups of 256) starts after the loss
    c48 = first(lambda r: "conv3x_kernel" in r["Kernel_Name"] and r["Grid_Size_X"] == "442368", tl)
    c24 = first(lambda r: "conv3x_kernel" in r["Kernel_Name"] and r["Grid_Size_X"] == "110592", tl)
    stem = first(lambda r: "stemx_kernel" in r["Kernel_Name"], tl)
    marks = [("backward 96^3 decoder level (head ... first 48^3 conv)", tl, c48["s"] if c48 else None),
             ("backward 48^3 decoder level", c48["s"] if c48 else None, c24["s"] if c24 else None)]
    for name, a, b in marks:
        if a and b and b > a:
            line(name, a, b)
    if stem:
        line("backward input block ... optimiser", stem["s"], t1)
    print("-- 250 us slices")
    a = t0
    while a < t1:
        b = min(a + 250000, t1)
        line("  %6.0f .. %6.0f us" % ((a - t0) / 1e3, (b - t0) / 1e3), a, b)
        a = b


if __name__ == "__main__":
    main()
