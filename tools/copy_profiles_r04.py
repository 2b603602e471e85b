"""Copies the summaries collected by tools/collect_profiles_r04.sh (gpurun_out/r04) into profiles/ under their per-round
names and builds profiles/r04_pmc.json from the separate rocprofv3 --pmc passes: HBM bytes per launch = FETCH_SIZE * 2
(the gfx950 correction of MI355X_MICROARCH.md: 128-byte requests tallied at 64 bytes) + WRITE_SIZE, both in KiB; SQ counters
averaged per launch; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)."""
import collections, csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "r04"), os.path.join(ROOT, "profiles")


def stats(sub):
    f = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    return f[0] if f else None


pairs = [(stats("bench"), "r04_bench_n4096_kernel_stats.csv"), (stats("bench_theta"), "r04_bench_theta_c32xk128_kernel_stats.csv"),
         (stats("bench_er7"), "r04_bench_theta_er7xk72_kernel_stats.csv"), (stats("dense"), "r04_bench_dense_driver_kernel_stats.csv"),
         (stats("bench8192"), "r04_bench_n8192_kernel_stats.csv"), (stats("squares8192"), "r04_squares_n8192_kernel_stats.csv"),
         (stats("squares_i8sym"), "r04_square_i8_product_launch_kernel_stats.csv"), (stats("refine_bucket"), "r04_refine_8388608_classes_kernel_stats.csv")]
for name in ("bench_under_rocprof.json", "bench_theta_under_rocprof.json", "bench_er7_under_rocprof.json", "dense_under_rocprof.json",
             "bench_n8192.json", "bench_n8192_under_rocprof.json", "config_times.txt", "eig_drivers.txt", "stress_seeds.txt", "big_instance_seeds.txt",
             "bd_failure_rates.txt", "ab_full_basis_image.json", "ab_square_kernel_128tiles.json", "ab_square_kernel_128tiles_theta.json",
             "restarts_per_gpu_2.json", "stedc_check.txt", "sytrd_time.txt", "sytrd_forms_check.txt"):
    pairs.append((os.path.join(src, name), "r04_" + name))
pairs.append((os.path.join(src, "bench_default.json"), "r04_bench_n4096.json"))
for a, b in pairs:
    if a and os.path.exists(a) and os.path.getsize(a) > 0:
        shutil.copyfile(a, os.path.join(dst, b))
        print("copied", b)
    else:
        print("MISSING", b)


def counters(sub, pat, skip_first=0):
    p = os.path.join(src, sub, "p_counter_collection.csv")
    if not os.path.exists(p):
        return {}
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {"avg": sum(v[skip_first:]) / max(1, len(v[skip_first:])), "launches": len(v[skip_first:])} for k, v in acc.items()}


def traffic(tag, pat, what, alg_bytes):
    f = counters(f"pmc_{tag}_FETCH_SIZE", pat).get("FETCH_SIZE")
    w = counters(f"pmc_{tag}_WRITE_SIZE", pat).get("WRITE_SIZE")
    if not f or not w:
        return None
    t = (2.0 * f["avg"] + w["avg"]) * 1024.0
    return {"kernel": what, "FETCH_SIZE_KB_avg": f["avg"], "WRITE_SIZE_KB_avg": w["avg"], "launches": min(f["launches"], w["launches"]),
            "traffic_bytes_per_launch": t, "algorithmic_bytes_per_launch": alg_bytes, "traffic_over_algorithmic": t / alg_bytes,
            "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950); separate --pmc passes per counter"}


def sq(tag, pat, what, mfma_cycles_each):
    m = {}
    for grp in ("mfma", "lds", "grbm"):
        m.update({k: v["avg"] for k, v in counters(f"pmc_{tag}_{grp}", pat).items()})
    if not m:
        return None
    out = {"kernel": what, **m}
    g = m.get("GRBM_GUI_ACTIVE")
    if g and m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        cyc = g / 8.0  # the counter is summed over the 8 XCDs
        out["derived"] = {"kernel_cycles": cyc, "MfmaUtil": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0),
                          "busy_cycles_per_mfma": m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_INSTS_MFMA"] if m.get("SQ_INSTS_MFMA") else None,
                          "wait_any_frac_of_wave_cycles": m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
                          "wait_inst_any_frac": m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
                          "active_inst_any_frac": m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
                          "wait_inst_lds_frac": m.get("SQ_WAIT_INST_LDS", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
                          "lds_array_active_frac_of_kernel": m.get("SQ_LDS_IDX_ACTIVE", 0) / (cyc * 256.0),
                          "note": "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* "
                                  "count quad-cycles summed over all waves; LDS array cycles per CU against the kernel's cycles"}
    return out


n = 4096
out = {
    "i8x2_lower": traffic("i8sym", "i8_symsquare_kernel", "i8_symsquare_kernel (persistent 256 x 256 macro-tiles), N=4096, 2 channels, lower triangle "
                          "(tools/pmc_probe.py 0 4096 102)", 2 * (n * n + 4 * n * n)),
    "i8x2_lower_counters": sq("i8sym", "i8_symsquare_kernel", "i8_symsquare_kernel, N=4096, 2 channels (tools/pmc_probe.py 0 4096 102), averages over 3 launches", 32),
    "i8x2_lower_128tiles": traffic("i8tri128", "gemm_tn_dma_kernel<0", "gemm_tn_dma_kernel<i8> (128 x 128 tiles; round 3's product launch), same operands "
                                   "(tools/pmc_probe.py 0 4096 202)", 2 * (n * n + 4 * n * n)),
    "i8x2_lower_128tiles_counters": sq("i8tri128", "gemm_tn_dma_kernel<0", "gemm_tn_dma_kernel<i8>, 128 x 128 lower-triangle tiles, N=4096, 2 channels", 32),
    "f32_n8192": traffic("f32n8192", "gemm_tn_dma256_kernel<1", "gemm_tn_dma256_kernel<f32>, N=8192 square (tools/pmc_probe.py 1 8192 1)", 2 * 4 * 8192 * 8192),
    "f32_n8192_counters": sq("f32n8192", "gemm_tn_dma256_kernel<1", "gemm_tn_dma256_kernel<f32>, N=8192 square", 32),
}
rb = {}
for kname in ("bk_count_kernel", "bk_scatter_kernel<1>", "bk_scatter_kernel<2>", "bk_resolve_kernel", "bk_first_count_kernel", "bk_label_first_kernel", "bk_label_rest_kernel"):
    t = traffic("refine_bucket", kname, kname, 16 * n * n)
    if t:
        rb[kname] = {k: t[k] for k in ("FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg", "traffic_bytes_per_launch", "launches")}
if rb:
    tot = sum(v["traffic_bytes_per_launch"] for v in rb.values())
    out["refine_bucketed_8388608_classes"] = {"kernels": rb, "traffic_bytes_per_refinement": tot, "algorithmic_bytes": 16 * n * n, "traffic_over_algorithmic": tot / (16.0 * n * n),
                                              "bytes_per_entry": tot / (n * n), "note": "tools/pmc_probe.py 3 4096 8388608: bucketed grouping (kernels_refine_bucket.hip), 16.7 M entries"}
json.dump(out, open(os.path.join(dst, "r04_pmc.json"), "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict) and "traffic_bytes_per_launch" in v:
        print(k, "traffic %.1f MB" % (v["traffic_bytes_per_launch"] / 1e6), "algorithmic %.1f MB" % (v["algorithmic_bytes_per_launch"] / 1e6),
              "ratio %.2f" % v["traffic_over_algorithmic"], "launches", v["launches"])
    elif isinstance(v, dict) and "derived" in v:
        print(k, "MfmaUtil %.3f" % v["derived"]["MfmaUtil"], "cycles %.0f" % v["derived"]["kernel_cycles"])
if "refine_bucketed_8388608_classes" in out:
    print("bucketed refine: %.1f B/entry" % out["refine_bucketed_8388608_classes"]["bytes_per_entry"])
