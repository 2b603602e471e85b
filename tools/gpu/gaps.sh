# kernel timeline of a few reductions of one bench instance: where the GPU idles between launches (rocprofv3 --kernel-trace)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
W=${1:-closed_scheme}
rm -rf /tmp/gaps
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/gaps -o g -- python3 bench.py --steps 6 --warmup 4 --skip-roofline --workload $W > gpurun_out/gaps_$W.json 2> gpurun_out/gaps_$W.err
python3 - $W <<'PY'
import csv, glob, sys
kt = glob.glob("/tmp/gaps/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in csv.DictReader(open(kt))]
mc = glob.glob("/tmp/gaps/**/*memory_copy_trace.csv", recursive=True)
for f in mc:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", "")))
rows.sort()
# the last complete reduction: from one first launch of the pair-source insert to the next
starts = [i for i, r in enumerate(rows) if "SrcPair, 1>" in r[2]]  # the first launch of a reduction's first refinement
a, b = starts[-3], starts[-2]
seg = rows[a:b]
t0 = seg[0][0]
busy = sum(e - s for s, e, _ in seg)
span = rows[b][0] - t0
print("one reduction of %s: %d launches, span %.1f us, kernels busy %.1f us, idle %.1f us" % (sys.argv[1], len(seg), span / 1e3, busy / 1e3, (span - busy) / 1e3))
prev_end = seg[0][1]
for i in range(1, len(seg) + 1):
    s = seg[i][0] if i < len(seg) else rows[b][0]
    gap = (s - prev_end) / 1e3
    if gap > 4.0:
        print("  gap %6.1f us after %-50s (at +%.1f us) before %s" % (gap, seg[i - 1][2], (prev_end - t0) / 1e3, seg[i][2] if i < len(seg) else "next reduction"))
    if i < len(seg): prev_end = max(prev_end, seg[i][1])
PY
