R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in "" sl32 sl127; do
  if [ -n "$v" ]; then export H264E_LIB=$R/ab/lib_$v.so; else unset H264E_LIB; fi
  rm -rf $R/gpurun_out/fz_$v; timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/fz_$v --output-format csv -- python3 $R/tools/clip_debug.py 600 > $R/gpurun_out/fz_$v.log 2>&1
  python3 - <<PY
import csv,glob
t=0;n=0
for f in glob.glob("$R/gpurun_out/fz_$v/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "h264e_mb" in r["Kernel_Name"]: t+=float(r["Counter_Value"]); n+=1
print("variant '$v': FETCH KB per launch %.0f (%d launches)" % (t/max(n,1), n), open("$R/gpurun_out/fz_$v.log").read().strip().split("\n")[-1][:60])
PY
done
