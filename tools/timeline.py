#!/usr/bin/env python3
"""Timeline of steady-state generations from a rocprofv3 kernel trace (+ memory-copy trace if present).

usage: timeline.py <trace-dir> [n_generations_to_print [first_generation]]
A generation starts at a k_glob_skip launch; prints, for n generations from `first_generation` (default: the last ones of the
run), every launch: offset from the window's first launch, duration, queue, name -- and how much of the window had a kernel
running at all."""
import csv
import glob
import os
import sys


def load(d):
    ev = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", "")))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "copy")))
    ev.sort()
    return ev


def main():
    d = sys.argv[1]
    ngen = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ev = load(d)
    marks = [i for i, e in enumerate(ev) if e[3].startswith("k_glob_skip")]
    if len(marks) < ngen + 2:
        raise SystemExit("too few generations in the trace")
    first = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) - ngen - 1
    lo, hi = marks[first], marks[first + ngen]
    t0 = ev[lo][0]
    print("generations in the trace:", len(marks), " window starts (us):", [round((ev[m][0] - t0) / 1e3, 1) for m in marks[first:first + ngen + 1]])
    for s, e, q, n in ev[lo:hi]:
        print("%9.1f %8.1f  %-5s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:60]))
    # busy fraction: union of all intervals over the window
    iv = sorted((s, e) for s, e, _, _ in ev[lo:hi])
    busy = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: busy += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    busy += ce - cs
    print("window %.1f us, some kernel running %.1f us (%.2f)" % ((ev[hi][0] - t0) / 1e3, busy / 1e3, busy / (ev[hi][0] - t0)))


if __name__ == "__main__":
    main()
