cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rm -rf gpurun_out/prof/ef_sq gpurun_out/prof/ef_sq2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/ef_sq -- python3 scripts/quick_eval_fused.py 64 > gpurun_out/prof/ef_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof/ef_sq2 -- python3 scripts/quick_eval_fused.py 64 > gpurun_out/prof/ef_sq2.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ('ef_sq','ef_sq2'):
    f=glob.glob('gpurun_out/prof/%s/*/*_counter_collection.csv'%d)[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if 'fused_topk' not in r['Kernel_Name']: continue
        k='topk'
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
    for k,c in agg.items():
        print(k,{a:round(v/len(n[k])/1e6,2) for a,v in c.items()})
PY
