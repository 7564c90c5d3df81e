#!/bin/bash
# Round 6: which stores of the forward recurrent launches leave L2?  FETCH_SIZE / WRITE_SIZE of the timed step
# (bench.py --traffic-child) with the shipped publish stores, with write-through publish stores on the same XCD-aware grid
# (kernel-selection bit 524288) and on the plain 3-D grid (bit 262144): per launch class, MB per launch.
#   tools/r06_fwd_traffic.sh > gpurun_out/r06_fwd_traffic.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for FL in 0 524288 262144; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pt_${FL}_$C
    if [ "$FL" = "0" ]; then
      rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pt_${FL}_$C -o run -- python3 $R/bench.py --traffic-child --precision 0 --steps 4 --warmup 3 > /dev/null 2>&1
    else
      AAS_ABLATION=1 AAS_DEBUG_FLAGS=$FL rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pt_${FL}_$C -o run -- python3 $R/bench.py --traffic-child --precision 0 --steps 4 --warmup 3 --allow-ablation > /dev/null 2>&1
    fi
  done
  python3 $R/tools/pmc_summary.py /tmp/pt_${FL}_FETCH_SIZE /tmp/pt_${FL}_WRITE_SIZE /tmp/pt_$FL.json "bench.py --traffic-child (debug flags $FL)" 7 > /dev/null 2>&1
done
python3 - <<'PY'
import json
names = {0: "shipped (XCD-aware grid, L2-resident publish stores once a set is verified co-located)", 524288: "bit 524288: XCD-aware grid, write-through publish stores",
         262144: "bit 262144: plain 3-D grid, write-through publish stores"}
print("HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes each; fetched = 2 x FETCH_SIZE x 1024, gfx950 correction), config-2 fp32 step")
for fl in (0, 524288, 262144):
    try:
        j = json.load(open("/tmp/pt_%d.json" % fl))
    except Exception as e:
        print(fl, "no summary:", e); continue
    print("\n== %s: %.2f GB per step" % (names[fl], (j["hbm_bytes_per_step"] or 0) / 1e9))
    for k in ("gru_fwd[N=30,H=1000]", "lstm_fwd[N=60,H=500]", "lstm_fwd[N=30,H=500]", "gru_bwd[N=30,H=1000]", "lstm_bwd[N=60,H=500]", "lstm_bwd[N=30,H=500]", "gemm_splitk_reduce"):
        v = j["by_class"].get(k)
        if v:
            print("  %-24s fetched %7.1f MB  written %7.1f MB  total %7.1f MB / launch  (%d launches)" % (k, 2 * v["FETCH_SIZE_KiB_avg"] * 1024 / 1e6, v["WRITE_SIZE_KiB_avg"] * 1024 / 1e6,
                                                                                                          v["hbm_bytes_per_launch"] / 1e6, v["dispatches"]))
PY
