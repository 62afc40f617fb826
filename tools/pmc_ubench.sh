# calibration of the SQ pipe-busy counters on streams whose issue cost is known (tools/ubench/valu_rate, 8 and 4 waves per SIMD)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for wps in 8 4; do
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_ubench_$wps -o p -- $R/tools/ubench/valu_rate /dev/null 0 $wps > $R/gpurun_out/pmc_ubench_$wps.log 2>&1 || echo "pmc ubench failed"
done
python3 - <<PY
import csv, glob, collections
for wps in (8, 4):
    acc = collections.defaultdict(dict)
    for f in glob.glob("$R/gpurun_out/pmc_ubench_%d/**/*counter_collection.csv" % wps, recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            acc[(row["Kernel_Name"], row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
    seen = {}
    for (k, d), c in acc.items():
        seen.setdefault(k, []).append(c)
    print("waves per SIMD", wps)
    for k, lst in seen.items():
        c = lst[-1]                      # the timed launch (the first is the warm-up)
        if "SQ_INSTS_VALU" not in c or c["SQ_INSTS_VALU"] < 1e6: continue
        print("  %-40s VALU %.3e  ACTIVE_VALU/INSTS %.3f  ACTIVE_VALUx4/(BUSY_CUx4) %.3f  WAIT_INST/WAVE_CYC %.3f  BUSY_CU/(GRBM/8*256) %.3f" % (
            k[:40], c["SQ_INSTS_VALU"], c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"], c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CU_CYCLES"],
            c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_BUSY_CU_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 256)))
PY
