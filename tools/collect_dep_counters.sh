#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: SQ / LDS counters of every deployed-net kernel.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/collect_dep_counters.sh r02'
# then, back in the container:  python tools/summarize_dep_counters.py r02
# Counter passes only (no API traces beside --pmc).  With a second argument "mfma" the f32 kernels are profiled in both
# forms (default: the all-VALU kernel; MDC_DEP_F32_MFMA=1: the alternates build's variant with the dense layer on the f32
# matrix pipe, measured slower in round 2 -- profiles/r02_depmfma_kernel_stats.csv) in separate processes.
set -e -o pipefail
R=$PWD
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
P="--output-format csv"
rm -rf $R/gpurun_out/pmc_${TAG}_dep_A $R/gpurun_out/pmc_${TAG}_dep_B $R/gpurun_out/pmc_${TAG}_depmfma_A $R/gpurun_out/pmc_${TAG}_depmfma_B $R/gpurun_out/prof_${TAG}_depmfma
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
B="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
D="python3 $R/tools/prof_deployed.py"
for pass in A B; do
  eval C=\$$pass
  rocprofv3 --pmc $C $P -d $R/gpurun_out/pmc_${TAG}_dep_$pass -- $D f32 bf16 f16 u8 > $R/gpurun_out/pmc_${TAG}_dep_$pass.log 2>&1
  if [ "$2" = "mfma" ]; then
  export MDC_DEP_F32_MFMA=1
  rocprofv3 --pmc $C $P -d $R/gpurun_out/pmc_${TAG}_depmfma_$pass -- $D f32 u8 > $R/gpurun_out/pmc_${TAG}_depmfma_$pass.log 2>&1
  unset MDC_DEP_F32_MFMA
  fi
  echo "pass $pass done"
done
# (kernel stats of the default kernels: tools/collect_profiles.sh, prof_${TAG}_dep)
if [ "$2" = "mfma" ]; then
export MDC_DEP_F32_MFMA=1
rocprofv3 --kernel-trace --stats $P -d $R/gpurun_out/prof_${TAG}_depmfma -- $D f32 u8 > $R/gpurun_out/prof_${TAG}_depmfma.log 2>&1
echo "kernel stats done"
fi
