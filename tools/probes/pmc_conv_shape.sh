ulimit -c 0; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/r06_convpmc_$tag -- python3 $R/tools/probes/conv_f16x1_shapes.py 32 256-1024 > /dev/null 2>$R/gpurun_out/r06_convpmc_$tag.err
done
cd $R
python3 - <<"PY"
import csv,glob,collections
for d in sorted(glob.glob("gpurun_out/r06_convpmc_*/")):
    fs=glob.glob(d+"*/*counter_collection.csv")
    if not fs: print(d,"no file"); continue
    agg=collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
        if not n.startswith("conv_gemm"): continue
        agg.setdefault((n,r["Grid_Size"],r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(k, len(v), "%.4g"%(sum(v)/len(v)))
PY
