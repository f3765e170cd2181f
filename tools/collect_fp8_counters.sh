set -e -o pipefail
R=$PWD
cd /tmp && export TMPDIR=/tmp
P="--output-format csv"
BW="python3 $R/bench.py --workload vtcnn2-c11-fp8-n2^20 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-traffic"
rm -rf $R/gpurun_out/pmc_mfma_vtfp8 $R/gpurun_out/pmc_lds_vtfp8
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA $P -d $R/gpurun_out/pmc_mfma_vtfp8 -- $BW > $R/gpurun_out/pmc_mfma_vtfp8.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES $P -d $R/gpurun_out/pmc_lds_vtfp8 -- $BW > $R/gpurun_out/pmc_lds_vtfp8.log 2>&1
echo done
