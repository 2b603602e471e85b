# three clock readings under i8_symsquare_kernel, side by side, plus the socket power:
#  (1) hwmon power / sclk while the kernel loops (tools/power_trace.py), (2) GRBM_GUI_ACTIVE per launch against the SAME
#  launches' durations from the kernel trace of that PMC pass, (3) the in-kernel clock64 / wall_clock64 sampler.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/clock_story.txt; : > $O
for K in "0 4096 102" "0 4096 202" "1 8192 1"; do
  echo "== power_trace $K" >> $O
  python3 tools/power_trace.py $K 3 >> $O 2>&1
done
rm -rf /tmp/pmc_grbm
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_grbm -o p -- python3 tools/pmc_probe.py 0 4096 102 > gpurun_out/pmc_grbm.log 2>&1
python3 - >> $O <<'PY'
import csv, glob
cc = glob.glob("/tmp/pmc_grbm/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("/tmp/pmc_grbm/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    if "i8_symsquare" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("== GRBM_GUI_ACTIVE pass: launches of i8_symsquare_kernel (2 channels, N = 4096) with their own durations")
vals = []
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "i8_symsquare" in r["Kernel_Name"] and r["Dispatch_Id"] in dur:
        cyc = float(r["Counter_Value"]); us = dur[r["Dispatch_Id"]]
        vals.append((cyc, us))
for cyc, us in vals:
    print("  GRBM_GUI_ACTIVE %.0f (sum over 8 XCDs) = %.0f per XCD, duration %.2f us in this pass -> %.0f MHz" % (cyc, cyc / 8, us, cyc / 8 / us))
PY
echo "== in-kernel sampler (sdpsr_profile_clock: clock64 against the 100 MHz wall clock on a side stream)" >> $O
python3 tools/clock_under_kernels.py >> $O 2>&1
cat $O
