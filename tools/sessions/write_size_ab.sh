cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in libmi_pt.so libmi_sp64.so; do
  MI_PT_LIB=$GRAFT_REPO_ROOT/master_amd/$lib rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/ws_$lib -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/ws_$lib.json 2>/dev/null
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/ws_$lib/*/*counter_collection.csv")[0]
tot={}
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:60]; tot[k]=tot.get(k,0)+float(r["Counter_Value"])
for k,v in tot.items():
    if "megakernel" in k: print("$lib", k, v*1024/1e9, "GB")
PY
done
