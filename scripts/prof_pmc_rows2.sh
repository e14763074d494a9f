# A/B of the two-rows-per-wave Cholesky kernel under the SQ counters (development)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
for V in 1 0; do
export IRSPACK_AMD_IALS_ROWS2=$V
rm -rf gpurun_out/prof/r2sq_$V gpurun_out/prof/r2sq2_$V
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/r2sq_$V -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --solver CHOLESKY > gpurun_out/prof/r2sq_$V.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof/r2sq2_$V -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --solver CHOLESKY > gpurun_out/prof/r2sq2_$V.log 2>&1
done
ls gpurun_out/prof/r2sq_1/*/ | head
