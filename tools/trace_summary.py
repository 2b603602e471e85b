"""Prints the kernel / copy timeline of the LAST `window_ms` milliseconds of a rocprofv3 --kernel-trace --memory-copy-trace
run (csv): start offset, duration, gap to the previous event, name."""
import csv, sys, glob, os
d = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
ct = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:78]) for f in kt for r in csv.DictReader(open(f))]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"]) for f in ct for r in csv.DictReader(open(f))]
ev.sort()
end = ev[-1][1]
sel = [e for e in ev if e[0] > end - win * 1e6]
t0, prev = sel[0][0], None
for s, e, n in sel:
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0, n))
    prev = e
