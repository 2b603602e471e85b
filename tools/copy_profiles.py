"""Copies the summaries collected by tools/collect_profiles.sh (gpurun_out/r02) into profiles/ under
their per-round names."""
import glob, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "r02"), os.path.join(ROOT, "profiles")


def stats(sub):
    f = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    return f[0] if f else None


pairs = [(stats("bench"), "r02_bench_n4096_kernel_stats.csv"), (stats("bench_theta"), "r02_bench_theta_c32xk128_kernel_stats.csv"),
         (stats("bench_er7"), "r02_bench_theta_er7xk72_kernel_stats.csv"), (stats("dense"), "r02_bench_dense_driver_kernel_stats.csv"),
         (stats("sq8192"), "r02_squares_n8192_kernel_stats.csv"),
         (os.path.join(src, "bench_under_rocprof.json"), "r02_bench_under_rocprof.json"),
         (os.path.join(src, "bench_theta_under_rocprof.json"), "r02_bench_theta_under_rocprof.json"),
         (os.path.join(src, "bench_er7_under_rocprof.json"), "r02_bench_er7_under_rocprof.json"),
         (os.path.join(src, "dense_under_rocprof.json"), "r02_dense_under_rocprof.json"),
         (os.path.join(src, "sq8192.json"), "r02_squares_n8192.json"),
         (os.path.join(src, "bench_default.json"), "r02_bench_n4096.json"),
         (os.path.join(src, "clock_under_kernels.txt"), "r02_clock_under_kernels.txt"),
         (os.path.join(src, "mfma_clock_probe.txt"), "r02_mfma_clock_probe.txt"),
         (os.path.join(src, "i8_w4_diag.txt"), "r02_i8_w4_diag.txt"),
         (os.path.join(src, "config_times.txt"), "r02_config_times.txt"),
         (os.path.join(src, "config2_bd_phases.txt"), "r02_config2_bd_phases.txt"),
         (os.path.join(src, "small_syev_time.txt"), "r02_small_syev_time.txt"),
         (os.path.join(src, "label_product.txt"), "r02_label_product.txt"),
         (os.path.join(src, "mfma_f64_probe.txt"), "r02_mfma_f64_probe.txt")]
for a, b in pairs:
    if a and os.path.exists(a) and os.path.getsize(a) > 0:
        shutil.copyfile(a, os.path.join(dst, b))
        print("copied", b)
    else:
        print("MISSING", b)
