"""Builds profiles/r02_pmc.json from the separate rocprofv3 --pmc passes collected by
tools/collect_profiles.sh (gpurun_out/r02/pmc_*): HBM bytes per launch = FETCH_SIZE * 2 (the gfx950
correction of MI355X_MICROARCH.md: FETCH_SIZE tallies 128-byte requests at 64 bytes) + WRITE_SIZE;
both counters are reported in KiB by rocprofv3."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "r02")


def avg(counter_dir, counter, kernel_pat, skip_first=1):
    rows = [r for r in csv.DictReader(open(os.path.join(src, counter_dir, "p_counter_collection.csv")))
            if r["Counter_Name"] == counter and kernel_pat in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rows][skip_first:]
    return sum(vals) / max(1, len(vals)), len(vals)


def entry(tag, kernel_pat, what, alg_bytes, skip_first=1):
    f, nf = avg(f"pmc_{tag}_FETCH_SIZE", "FETCH_SIZE", kernel_pat, skip_first)
    w, nw = avg(f"pmc_{tag}_WRITE_SIZE", "WRITE_SIZE", kernel_pat, skip_first)
    traffic = (2.0 * f + w) * 1024.0
    return {"kernel": what, "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "launches": min(nf, nw),
            "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg_bytes,
            "traffic_over_algorithmic": traffic / alg_bytes,
            "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950); separate --pmc passes per counter"}


n = 4096
out = {
    "i8x4_lower": entry("i8tri", "gemm_tn_dma_kernel<0", "gemm_tn_dma_kernel<i8>, N=4096, 4 channels, lower-triangle tiles (tools/pmc_probe.py 0 4096 104)",
                        4 * (n * n + 4 * n * n)),
    "sytrd_symv": entry("symv", "sytrd_symv_kernel", "sytrd_symv_kernel, N=4096, average over the 4095 launches of a sweep (tools/pmc_probe.py 5 4096)",
                        8.0 * n * (2 * n - 1) / 12.0, skip_first=0),
    "f32_n8192": entry("f32_8192", "gemm_tn_dma256_kernel<1", "gemm_tn_dma256_kernel<f32>, N=8192 square (tools/pmc_probe.py 1 8192 1)",
                       2 * 4 * 8192 * 8192),
}
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_pmc.json"), "w"), indent=1)
for k, v in out.items():
    print(k, "traffic %.1f MB" % (v["traffic_bytes_per_launch"] / 1e6), "algorithmic %.1f MB" % (v["algorithmic_bytes_per_launch"] / 1e6),
          "ratio %.2f" % v["traffic_over_algorithmic"], "launches", v["launches"])
