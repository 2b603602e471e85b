"""Copies the summaries collected by tools/collect_profiles_r03.sh (gpurun_out/r03) into profiles/ under their
per-round names and builds profiles/r03_pmc.json from the separate rocprofv3 --pmc passes (HBM bytes per launch =
FETCH_SIZE * 2 (the gfx950 correction of MI355X_MICROARCH.md: 128-byte requests tallied at 64 bytes) + WRITE_SIZE, both
in KiB; SQ counters averaged per launch)."""
import collections, csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "r03"), os.path.join(ROOT, "profiles")


def stats(sub):
    f = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    return f[0] if f else None


pairs = [(stats("bench"), "r03_bench_n4096_kernel_stats.csv"), (stats("bench_theta"), "r03_bench_theta_c32xk128_kernel_stats.csv"),
         (stats("bench_er7"), "r03_bench_theta_er7xk72_kernel_stats.csv"), (stats("dense"), "r03_bench_dense_driver_kernel_stats.csv"),
         (stats("dense1024"), "r03_syev_n1024_kernel_stats.csv")]
for name in ("bench_under_rocprof.json", "bench_theta_under_rocprof.json", "bench_er7_under_rocprof.json", "dense_under_rocprof.json",
             "power_under_kernels.txt", "clock_under_kernels.txt", "config_times.txt", "config2_bd_phases.txt", "eig_drivers.txt",
             "stress_seeds.txt", "big_instance_seeds.txt", "bd_failure_rates.txt", "ab_full_basis_image.json", "ab_no_verify_shortcut.json",
             "ab_small_eigen_on_device.json", "ab_channels4.json", "ab_channels4_theta.json", "stedc_check.txt", "sytrd_time.txt"):
    pairs.append((os.path.join(src, name), "r03_" + name))
pairs.append((os.path.join(src, "bench_default.json"), "r03_bench_n4096.json"))
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


n = 4096
lenp = n * (n + 1) // 2
out = {
    # 2 channels, lower-triangle tiles (33/64 of the square): int8 operands read once, int32 results of the computed tiles written
    "i8x2_lower": traffic("i8tri", "gemm_tn_dma_kernel<0", "gemm_tn_dma_kernel<i8>, N=4096, 2 channels, lower-triangle tiles (tools/pmc_probe.py 0 4096 102)",
                          2 * (n * n + 4 * n * n)),
    # joint insert pass on the packed lower triangle, r = 2, T = 2: label 4 + U 16 + channels 8 bytes read, 4 bytes of slots written per entry
    "insert_joint_r2_t2": traffic("insert", "refine_insert_kernel<sdpsr::SrcJoint", "refine_insert_kernel<SrcJoint<2,2>,8,1024> in theta_c32xk128 "
                                  "(n (n + 1) / 2 packed entries)", lenp * 32),
    "insert_pair": traffic("insert", "refine_insert_kernel<sdpsr::SrcPair", "refine_insert_kernel<SrcPair,8,1024> (initial partition, packed)", lenp * 20),
    "verify_joint_r2_t2": traffic("insert", "verify_lower_kernel<2, 2, true>", "verify_lower_kernel<2,2,true> (compare with class representatives, packed)", lenp * 28),
}
# the row form of the tridiagonalisation: launches of sytrd_row_kernel<NCH> while the trailing matrix spans NCH chunks of 128 columns
# (tools/sytrd_time.py with SDPSR_FLAG_NO_GRAPH).  Algorithmic bytes of launch j: 16 (n - j - 1)^2 (read + write of the trailing matrix);
# averaged over the launches of the variant.
def row_alg(n, nch):
    js = [j for j in range(n - 1) if (((n + 127) // 128 - ((j + 1) >> 7)) <= nch) and (nch == 2 or ((n + 127) // 128 - ((j + 1) >> 7)) > nch // 2)]
    return sum(16.0 * (n - j - 1) ** 2 for j in js) / max(1, len(js))
for nn, nch in ((2048, 16), (2048, 8), (1024, 8), (1024, 4)):
    t = traffic("rows%d" % nn, "sytrd_row_kernel<%d>" % nch, "sytrd_row_kernel<%d> of a tridiagonalisation of order %d (launches with %d..%d chunks of 128 columns left)" % (nch, nn, nch // 2 + 1, nch),
                row_alg(nn, nch))
    if t:
        out["sytrd_rows_n%d_nch%d" % (nn, nch)] = t
sq = {}
for tag, pat in (("joint", "refine_insert_kernel<sdpsr::SrcJoint"), ("pair", "refine_insert_kernel<sdpsr::SrcPair"), ("verify", "verify_lower_kernel<2, 2, true>")):
    a = counters("pmc_insert_A", pat)
    b = counters("pmc_insert_B", pat)
    g = counters("pmc_insert_G", pat)
    m = {**{k: v["avg"] for k, v in a.items()}, **{k: v["avg"] for k, v in b.items()}, **{k: v["avg"] for k, v in g.items()}}
    if m:
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        entries64 = lenp / 64.0
        m["derived"] = {"valu_wave_instructions_per_64_entries": m.get("SQ_INSTS_VALU", 0) / entries64,
                        "salu_per_64_entries": m.get("SQ_INSTS_SALU", 0) / entries64, "lds_per_64_entries": m.get("SQ_INSTS_LDS", 0) / entries64,
                        "vmem_rd_per_64_entries": m.get("SQ_INSTS_VMEM_RD", 0) / entries64,
                        "wait_any_frac_of_wave_cycles": m.get("SQ_WAIT_ANY", 0) / wc if wc else None,
                        "wait_inst_any_frac": m.get("SQ_WAIT_INST_ANY", 0) / wc if wc else None,
                        "active_inst_any_frac": m.get("SQ_ACTIVE_INST_ANY", 0) / wc if wc else None,
                        "lds_bank_conflict_frac_of_lds_active": (m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]) if m.get("SQ_LDS_IDX_ACTIVE") else None,
                        "wait_inst_lds_frac": m.get("SQ_WAIT_INST_LDS", 0) / wc if wc else None}
        sq[tag] = m
out["insert_pass_sq_counters"] = {"note": "rocprofv3 --pmc, two SQ passes of 8 counters + GRBM_GUI_ACTIVE, bench.py --workload theta_c32xk128; SQ_WAVE_CYCLES / SQ_WAIT_* / "
                                          "SQ_ACTIVE_INST_* count quad-cycles summed over all waves", **sq}
json.dump(out, open(os.path.join(dst, "r03_pmc.json"), "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict) and "traffic_bytes_per_launch" in v:
        print(k, "traffic %.1f MB" % (v["traffic_bytes_per_launch"] / 1e6), "algorithmic %.1f MB" % (v["algorithmic_bytes_per_launch"] / 1e6),
              "ratio %.2f" % v["traffic_over_algorithmic"], "launches", v["launches"])
