set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in base noperxcd; do
  if [ $v = base ]; then unset DR3HIP_LIB; else export DR3HIP_LIB=$R/3dr_amd/variants/libdr3hip_$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_w2_${v}_$c -o p -- python3 $R/tools/warp_quick2.py > $R/gpurun_out/pmc_w2_${v}_$c.log 2>&1
  done
done
python3 - <<PY
import csv, glob, statistics, os
R=os.environ["GRAFT_REPO_ROOT"]
for v in ("base","noperxcd"):
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        acc={}
        for f in glob.glob(R+"/gpurun_out/pmc_w2_%s_%s/**/*counter_collection.csv"%(v,c), recursive=True):
            for row in csv.DictReader(open(f,newline="")):
                if "warp_tile" in row["Kernel_Name"]:
                    acc.setdefault(row["Grid_Size"] if "Grid_Size" in row else row.get("Grid_Size_X","?"),[]).append(float(row["Counter_Value"]))
        print(v,c,{k:(len(x),statistics.median(x)) for k,x in acc.items()})
PY
