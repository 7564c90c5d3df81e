#!/bin/bash
# Average shader clock of the fp32 GEMM launches with and without their operand loads: GRBM_GUI_ACTIVE cycles / kernel duration.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05/gemm_clock
mkdir -p $O
export GEMM32_SKIP_LEGACY=1 LD_LIBRARY_PATH=$R/aas_enhancement_amd/lib AAS_ABLATION=1
for fl in 0 64 1; do
  export GEMM32_FLAGS=$fl
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/f$fl -o gemm --output-format csv -- $R/tools/bin/gemm32_bench time > $O/f$fl.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for fl in (0, 64, 1):
    f = glob.glob("$O/f%d/**/*counter_collection.csv" % fl, recursive=True)
    if not f:
        print("flags", fl, "no counter file", glob.glob("$O/f%d/**/*" % fl, recursive=True)[:5]); continue
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for r in rows:
        if "gemm32_kernel" not in r["Kernel_Name"] or r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        key = (r["Kernel_Name"][:60], r["Grid_Size"])
        a = agg[key]; a[0] += float(r["Counter_Value"]); a[1] += dur; a[2] += 1
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        print("flags %3d  %-60s grid %-8s n=%4d  avg %.1f us  GUI_ACTIVE/ns = %.3f (x8 XCDs summed?)" % (fl, k[0], k[1], a[2], a[1] / a[2] / 1e3, a[0] / a[1]))
PY
