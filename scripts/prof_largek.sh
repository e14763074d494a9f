# rocprofv3 passes over the K > 64 kernels (kernel trace + two SQ PMC passes per configuration)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
for K in 128 256; do
  for S in CHOLESKY CG; do
    rm -rf gpurun_out/prof/lk_kt_${K}_$S gpurun_out/prof/lk_sq_${K}_$S gpurun_out/prof/lk_sq2_${K}_$S
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/lk_kt_${K}_$S -- python3 bench.py --K $K --solver $S --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/prof/lk_kt_${K}_$S.log 2>&1
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/lk_sq_${K}_$S -- python3 bench.py --K $K --solver $S --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/prof/lk_sq_${K}_$S.log 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/prof/lk_sq2_${K}_$S -- python3 bench.py --K $K --solver $S --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/prof/lk_sq2_${K}_$S.log 2>&1
    tail -1 gpurun_out/prof/lk_kt_${K}_$S.log | cut -c1-400
  done
done
