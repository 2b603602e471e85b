# HBM-side traffic of the dominant kernels by PMC (separate --pmc passes per counter, --kernel-trace only, per
# MI355X_MICROARCH.md): traffic = 2 * FETCH_SIZE (gfx950 correction) + WRITE_SIZE, in KiB.  Writes gpurun_out/r05/pmc.json
# (copied to profiles/r05_pmc.json; bench.py reads roofline.traffic from it).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
run() {  # tag, probe args
  local tag=$1; shift
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pr_${tag}_$C
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pr_${tag}_$C -o p -- python3 tools/pmc_probe.py "$@" > gpurun_out/r05/pmc_${tag}_$C.log 2>&1
    cp $(find /tmp/pr_${tag}_$C -name "*counter_collection.csv" | head -1) gpurun_out/r05/pmc_${tag}_$C.csv
    cp $(find /tmp/pr_${tag}_$C -name "*kernel_trace.csv" | head -1) gpurun_out/r05/pmc_${tag}_${C}_trace.csv
  done
}
run i8x2_lower 0 4096 102
run sytrd_symv 5 4096
python3 - <<'PY'
import csv, json
out = {}
def avg(tag, C, pat):
    rows = [r for r in csv.DictReader(open(f"gpurun_out/r05/pmc_{tag}_{C}.csv")) if r["Counter_Name"] == C and pat in r["Kernel_Name"]]
    return sum(float(r["Counter_Value"]) for r in rows) / max(len(rows), 1), len(rows)
for tag, pat, name, alg in (("i8x2_lower", "i8_symsquare_kernel", "i8_symsquare_kernel (persistent 256 x 256 macro-tiles), N=4096, 2 channels, lower triangle (tools/pmc_probe.py 0 4096 102)", 2 * (4096 * 4096 + 4 * 4096 * 4096)),
                            ("sytrd_symv", "sytrd_symv_kernel", "sytrd_symv_kernel, N=4096, average over the columns of the panel part (tools/pmc_probe.py 5 4096)", None)):
    f, nl = avg(tag, "FETCH_SIZE", pat)
    w, _ = avg(tag, "WRITE_SIZE", pat)
    dur = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):  # the kernel's duration IN the counter passes (kernel trace of the same run)
        rows = [r for r in csv.DictReader(open(f"gpurun_out/r05/pmc_{tag}_{C}_trace.csv")) if pat in r["Kernel_Name"]]
        dur[C] = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / max(len(rows), 1) / 1e3
    e = {"kernel": name, "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "launches": nl, "traffic_bytes_per_launch": (2 * f + w) * 1024,
         "avg_duration_us_in_the_counter_passes": {k: round(v, 2) for k, v in dur.items()},
         "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950); separate --pmc passes per counter"}
    if alg:
        e["algorithmic_bytes_per_launch"] = alg
        e["traffic_over_algorithmic"] = e["traffic_bytes_per_launch"] / alg
    out[tag] = e
    print(tag, nl, "launches", round(e["traffic_bytes_per_launch"] / 1e6, 1), "MB per launch")
json.dump(out, open("gpurun_out/r05/pmc.json", "w"), indent=1)
PY
