#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: collects the rocprofv3 data behind profiles/.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r02'
# then, back in the container:  python tools/summarize_profiles.py r02 ; python tools/summarize_dep_counters.py r02
# Kernel timing and PMC counters are separate runs (gpurun refuses --pmc combined with API traces), and
# FETCH_SIZE / WRITE_SIZE are separate passes as MI355X_MICROARCH.md prescribes.  Directory names carry no round tag
# (gpurun_out/ is scratch): summarize_profiles.py stamps the round on what it copies into profiles/.
set -e -o pipefail
R=$PWD
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
P="--output-format csv"
rm -rf $R/gpurun_out/prof_${TAG}_bench $R/gpurun_out/prof_${TAG}_bench_extras $R/gpurun_out/pmc_*_vt $R/gpurun_out/pmc_*_dep $R/gpurun_out/prof_${TAG}_dep
if [ "$2" != "dep" ]; then
# the headline kernels at the headline launch size only (the extra legs launch the same kernels at other sizes -- one
# window, 65,536 f32 frames -- which would mix into the per-kernel averages): --no-extras; the extras get their own file
rocprofv3 --kernel-trace --stats $P -d $R/gpurun_out/prof_${TAG}_bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-live-traffic > $R/gpurun_out/prof_${TAG}_bench.log 2>&1
rocprofv3 --kernel-trace --stats $P -d $R/gpurun_out/prof_${TAG}_bench_extras -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic > $R/gpurun_out/prof_${TAG}_bench_extras.log 2>&1
echo "kernel stats done"
B="python3 $R/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-traffic"
rocprofv3 --pmc FETCH_SIZE $P -d $R/gpurun_out/pmc_fetch_vt -- $B > $R/gpurun_out/pmc_fetch_vt.log 2>&1
rocprofv3 --pmc WRITE_SIZE $P -d $R/gpurun_out/pmc_write_vt -- $B > $R/gpurun_out/pmc_write_vt.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA $P -d $R/gpurun_out/pmc_mfma_vt -- $B > $R/gpurun_out/pmc_mfma_vt.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES $P -d $R/gpurun_out/pmc_lds_vt -- $B > $R/gpurun_out/pmc_lds_vt.log 2>&1
# HBM traffic of the other two VT-CNN2 modes' kernels (fp8 conv2 at 2^20 frames, f32 at 65,536): bench.py's extra legs
for W in "vtcnn2-c11-fp8-n2^20:fp8" "vtcnn2-c3-f32-n65536:f32"; do
  WL=${W%%:*}; SFX=${W##*:}
  BW="python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-traffic"
  rm -rf $R/gpurun_out/pmc_fetch_vt$SFX $R/gpurun_out/pmc_write_vt$SFX
  rocprofv3 --pmc FETCH_SIZE $P -d $R/gpurun_out/pmc_fetch_vt$SFX -- $BW > $R/gpurun_out/pmc_fetch_vt$SFX.log 2>&1
  rocprofv3 --pmc WRITE_SIZE $P -d $R/gpurun_out/pmc_write_vt$SFX -- $BW > $R/gpurun_out/pmc_write_vt$SFX.log 2>&1
  if [ "$SFX" = "fp8" ]; then      # MFMA-pipe and LDS / wait counters of the fp8 conv kernel too (BASELINE configs[4])
    rm -rf $R/gpurun_out/pmc_mfma_vtfp8 $R/gpurun_out/pmc_lds_vtfp8
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA $P -d $R/gpurun_out/pmc_mfma_vtfp8 -- $BW > $R/gpurun_out/pmc_mfma_vtfp8.log 2>&1
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES $P -d $R/gpurun_out/pmc_lds_vtfp8 -- $BW > $R/gpurun_out/pmc_lds_vtfp8.log 2>&1
  fi
done
echo "vtcnn2 counters done"
fi
D="python3 $R/tools/prof_deployed.py f32 bf16 f16 fp8 u8 q612"
rocprofv3 --kernel-trace --stats $P -d $R/gpurun_out/prof_${TAG}_dep -- $D > $R/gpurun_out/prof_${TAG}_dep.log 2>&1
rocprofv3 --pmc FETCH_SIZE $P -d $R/gpurun_out/pmc_fetch_dep -- $D > $R/gpurun_out/pmc_fetch_dep.log 2>&1
rocprofv3 --pmc WRITE_SIZE $P -d $R/gpurun_out/pmc_write_dep -- $D > $R/gpurun_out/pmc_write_dep.log 2>&1
echo "deployed done"
# the training step (round 5): kernel times at the reference's batch geometry
rm -rf $R/gpurun_out/prof_${TAG}_train
rocprofv3 --kernel-trace --stats $P -d $R/gpurun_out/prof_${TAG}_train -- python3 $R/tools/prof_train.py > $R/gpurun_out/prof_${TAG}_train.log 2>&1
echo "training done"
