"""Socket power and shader clock while ONE product kernel runs in a loop of >= 2 s.

usage: power_trace.py <kind> <n> <aux> [seconds=3]     (kind / aux as in sdpsr_profile_kernel, include/sdpsr_prof.h:
                                                        0 4096 102 = the int8 square of the product path: 2 channels,
                                                        lower-triangle tiles; 1 4096 1 = fp32 square; 2 4096 1 = fp64 GEMM)
A child process runs the launches (so that this process never touches the GPU and only samples); the samples come
from the amdgpu hwmon files (power1_average / power1_input in microwatts, freq1_input = sclk in Hz), read every ~10 ms;
`rocm-smi --showpower --showclocks --json` is the fallback (~3 samples/s).  Prints one line per phase (idle before,
under load, idle after): samples, median / max power in W, median sclk in MHz, and the power cap."""
import glob, json, os, statistics, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kind, n, aux = sys.argv[1], sys.argv[2], sys.argv[3]
seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0


def hwmon_files(bdf=None):
    """hwmon files of the GPU with PCI address `bdf` (the one the child process computes on: a box shows all of its
    GPUs in sysfs, the process sees one), else of the first card that has a power sensor"""
    out = {}
    roots = sorted(glob.glob(f"/sys/bus/pci/devices/{bdf.lower()}/hwmon/hwmon*")) if bdf else []
    roots += [] if roots else sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    for h in roots:
        for key, names in (("power", ("power1_average", "power1_input")), ("sclk", ("freq1_input",)), ("cap", ("power1_cap",))):
            for nm in names:
                p = os.path.join(h, nm)
                if key not in out and os.path.exists(p):
                    try:
                        float(open(p).read())
                        out[key] = p
                    except Exception:  # noqa: BLE001
                        pass
        if "power" in out:
            break
    return out


def sample_hwmon(files):
    r = {}
    for k, p in files.items():
        try:
            r[k] = float(open(p).read())
        except Exception:  # noqa: BLE001
            pass
    return r.get("power", float("nan")) / 1e6, r.get("sclk", float("nan")) / 1e6, r.get("cap", float("nan")) / 1e6


def sample_smi():
    try:
        js = json.loads(subprocess.check_output(["rocm-smi", "--showpower", "--showclocks", "--json"], stderr=subprocess.DEVNULL, timeout=5))
        card = js[sorted(js)[0]]
        pw = next((float(v) for k, v in card.items() if "ower" in k and "(W)" in k), float("nan"))
        sc = next((float(str(v).strip("()").replace("Mhz", "").replace("MHz", "")) for k, v in card.items() if k.startswith("sclk")), float("nan"))
        return pw, sc, float("nan")
    except Exception:  # noqa: BLE001
        return float("nan"), float("nan"), float("nan")


files, use_hwmon, sample = {}, False, sample_smi


def choose_source(bdf):
    global files, use_hwmon, sample
    files = hwmon_files(bdf)
    use_hwmon = "power" in files
    sample = (lambda: sample_hwmon(files)) if use_hwmon else sample_smi
    print("GPU", bdf, "source:", files if use_hwmon else "rocm-smi --showpower --showclocks --json", flush=True)


def collect(duration, proc=None):
    rows = []
    t0 = time.time()
    while (time.time() - t0 < duration) if proc is None else (proc.poll() is None):
        rows.append((time.time(),) + sample())
        if use_hwmon:
            time.sleep(0.01)
    return rows


def show(tag, rows):
    pw = [r[1] for r in rows if r[1] == r[1]]
    sc = [r[2] for r in rows if r[2] == r[2]]
    cap = [r[3] for r in rows if r[3] == r[3]]
    print(f"{tag}: samples {len(rows)}, power W median {statistics.median(pw) if pw else float('nan'):.1f} max {max(pw) if pw else float('nan'):.1f}, "
          f"sclk MHz median {statistics.median(sc) if sc else float('nan'):.0f} min {min(sc) if sc else float('nan'):.0f}, power cap W {cap[0] if cap else float('nan'):.0f}", flush=True)


child = r'''
import sys, os, time, ctypes as C
sys.path.insert(0, %r)
from __graft_entry__ import load_package
pkg = load_package()
prof = pkg._lib.load_prof_library()
kind, n, aux, seconds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
with pkg.Context(seed=1) as ctx:
    v = C.c_double(0)
    ctx.check(prof.sdpsr_profile_kernel(ctx._h, kind, n, aux, 20, C.byref(v)))
    reps = max(20, int(seconds * 1e3 / max(v.value, 1e-3)))
    import ctypes as C2
    hip = C2.CDLL("libamdhip64.so")
    buf = C2.create_string_buffer(64)
    hip.hipDeviceGetPCIBusId(buf, 64, 0)
    print("READY", buf.value.decode(), flush=True)
    time.sleep(1.0)
    t = time.time()
    ctx.check(prof.sdpsr_profile_kernel(ctx._h, kind, n, aux, reps, C.byref(v)))
    print("kernel kind %%d n %%d aux %%d: %%d launches, %%.4f ms per launch, loop %%.2f s" %% (kind, n, aux, reps, v.value, time.time() - t), flush=True)
''' % ROOT
p = subprocess.Popen([sys.executable, "-c", child, kind, n, aux, str(seconds)], stdout=subprocess.PIPE, text=True)
line = p.stdout.readline()  # "READY <pci bus id>": library loaded, buffers allocated, kernel warmed; the loop starts 1 s later
choose_source(line.split()[1] if len(line.split()) > 1 else None)
warm = collect(0.9)
show("idle (kernel warmed, before the loop)", warm)
rows = collect(0, proc=p)
print(p.stdout.read().strip(), flush=True)
# the loop starts ~1 s after READY: drop the samples of the sleep and the first 0.2 s of the ramp
t_start = rows[0][0] + 0.3 if rows else 0
show("under load", [r for r in rows if r[0] >= t_start] or rows)
show("idle after", collect(1.0))
