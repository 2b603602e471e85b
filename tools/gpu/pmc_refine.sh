# FETCH_SIZE / WRITE_SIZE of every kernel of the many-classes refinement probe (separate --pmc passes, per the guide);
# prints bytes per entry per kernel: traffic = 2 * FETCH_SIZE (gfx950 correction) + WRITE_SIZE, in KiB
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
N=${1:-4096}; CLS=${2:-8388608}
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o p -- python3 tools/pmc_probe.py 3 $N $CLS > gpurun_out/pmc_refine_$C.log 2>&1
  cp $(find /tmp/pmc_$C -name "*counter_collection.csv" | head -1) gpurun_out/pmc_refine_${C}.csv
done
python3 - $N <<'PY'
import csv, sys, collections
n = int(sys.argv[1]); ln = n * n
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for i, C in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    rows = [r for r in csv.DictReader(open(f"gpurun_out/pmc_refine_{C}.csv")) if r["Counter_Name"] == C]
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        tot[k][i] += float(r["Counter_Value"])
        if i == 0: tot[k][2] += 1
# the probe runs the refinement 3 times (warm-up + 2): per call = / 3
s = 0.0
for k, (f, w, c) in sorted(tot.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1])):
    b = (2 * f + w) * 1024 / 3.0
    if "fill_test" in k: continue
    s += b
    print("%-50s launches %3d  %8.1f MB per call = %6.1f B/entry" % (k[:50], c, b / 1e6, b / ln))
print("total %.1f MB per refinement = %.1f B/entry (algorithmic 16)" % (s / 1e6, s / ln))
PY
