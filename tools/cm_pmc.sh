#!/bin/bash
# Counter evidence for the ContactMap stage at d = 24,927: rocprofv3 kernel-trace + stats, then
# FETCH_SIZE and WRITE_SIZE in passes of their own (--kernel-trace only, as MI355X_MICROARCH.md's
# HBM section prescribes; gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read stream -> x2,
# WRITE_SIZE exact), of tools/bench_contactmap.py.  Per kernel: the LARGEST dispatch (the
# chr1@10kb-sized one; the warm-up calls on a 300-bin map are the small ones).
#   -> gpurun_out/cm_pmc.txt
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp; O=$R/gpurun_out/cmpmc; rm -rf $O; mkdir -p $O
B="python3 $R/tools/bench_contactmap.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O.trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > $O.fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B > $O.write.log 2>&1 || exit 1
# calibration on known byte counts in the SAME access widths (the guide: "other access widths
# are uncalibrated"): tools/probes/rw_probe reads / writes 4.97 GB with 8 and 16 B per lane
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -- $R/tools/probes/rw_probe > $O.cal_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -- $R/tools/probes/rw_probe > $O.cal_write.log 2>&1 || exit 1
python3 - > $R/gpurun_out/cm_pmc.txt <<PY
import csv, glob, re
d = 24927
pairs = d * (d + 1) // 2
alg = {"normalize128_kernel": ("8 B read + 16 B written per upper pair", pairs * 24),
       "column_sums_kernel": ("8 B per element", d * d * 8),
       "symv_upper_kernel": ("8 B per upper pair", pairs * 8),
       "symv_kernel": ("8 B per element (both triangles)", d * d * 8),
       "gram_kernel": ("the centred matrix once (8 B per element; tiles re-read it from L2 / MALL)", d * d * 8),
       "center_rows_kernel": ("8 B read + 8 B written per element", d * d * 16),
       "center_rows_reg_kernel": ("8 B read + 8 B written per element", d * d * 16),
       "corr_finalize_kernel": ("8 B read per upper pair of the Gram matrix + 16 B written", pairs * 24),
       "pack_units_from_matrix_kernel": ("8 B read per upper pair + 4 B written", pairs * 12)}
key = re.compile("(" + "|".join(alg) + r")\\b")
def biggest(sub, counter):
    out = {}
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            m = key.search(r["Kernel_Name"])
            if m and r["Counter_Name"] == counter:
                out[m.group(1)] = max(out.get(m.group(1), 0.0), float(r["Counter_Value"]))
    return out
dur = {}
for f in glob.glob("$O/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        m = key.search(r["Kernel_Name"])
        if m:
            dur[m.group(1)] = max(dur.get(m.group(1), 0.0), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
fetch, write = biggest("fetch", "FETCH_SIZE"), biggest("write", "WRITE_SIZE")
# calibration: counter KB per known byte, by access width
probe_bytes = d * d * 8 // 16 * 16
def probe(sub, counter, pattern):
    v = []
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and re.search(pattern, r["Kernel_Name"]):
                v.append(float(r["Counter_Value"]))
    return sum(v) / len(v) if v else None
cal = {"read8": probe("cal_fetch", "FETCH_SIZE", r"k<double, 0>"),
       "read16": probe("cal_fetch", "FETCH_SIZE", r"k<double __vector\(2\), 0>|k<.*vector.*, 0>"),
       "write8": probe("cal_write", "WRITE_SIZE", r"k<double, 1>"),
       "write16": probe("cal_write", "WRITE_SIZE", r"k<.*vector.*, 1>")}
fac = {k: (probe_bytes / (v * 1024) if v else None) for k, v in cal.items()}
print("calibration (tools/probes/rw_probe, %.2f GB known): bytes per counted KiB x 1024 -> factor read 8 B/lane %s, "
      "read 16 B/lane %s, write 8 B/lane %s, write 16 B/lane %s" % (probe_bytes / 1e9,
      *("%.3f" % fac[k] if fac[k] else "n/a" for k in ("read8", "read16", "write8", "write16"))))
f_r8, f_r16 = fac["read8"] or 2.0, fac["read16"] or 2.0
f_w8 = fac["write8"] or 1.0
print("ContactMap stage at d = 24,927 (4.97 GB fp64 matrix): largest dispatch of each kernel; HBM = FETCH_SIZE x 1024 x "
      "read factor + WRITE_SIZE x 1024 x write factor (8 B/lane kernels; symv_kernel and gram_kernel load 16 B/lane)")
print("%-30s %10s %12s %12s %10s %8s  %s" % ("kernel", "us", "alg GB", "HBM GB", "HBM/alg", "TB/s alg", "algorithmic bytes"))
for k, (what, b) in alg.items():
    if k not in dur:
        continue
    fr = f_r16 if k in ("symv_kernel", "gram_kernel") else f_r8
    hbm = fetch.get(k, 0.0) * 1024 * fr + write.get(k, 0.0) * 1024 * f_w8
    print("%-30s %10.1f %12.3f %12.3f %10.3f %8.2f  %s" % (k, dur[k], b / 1e9, hbm / 1e9, hbm / b, b / dur[k] / 1e6, what))
PY
cat $R/gpurun_out/cm_pmc.txt
