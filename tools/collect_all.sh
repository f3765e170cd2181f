#!/bin/bash
# Everything profiles/rNN_* is made from, in ONE gpurun call (one box: the HIP-event launch times of the bench line and the
# rocprofv3 averages then come from the same silicon; boxes differ by +-3 %):
#   /usr/local/graft/bin/gpurun --timeout 1150 -- 'bash tools/collect_all.sh r02'
# then here:  bash tools/collect_all.sh r02 summarize
set -e -o pipefail
TAG=${1:-r05}
if [ "$2" = "summarize" ]; then
    python3 tools/summarize_profiles.py $TAG > /dev/null
    python3 tools/summarize_dep_counters.py $TAG > /dev/null
    cp gpurun_out/bench_${TAG}_final.json profiles/${TAG}_bench_line.json
    echo "profiles/${TAG}_* refreshed"
    exit 0
fi
bash tools/collect_profiles.sh $TAG > gpurun_out/collect_${TAG}.log 2>&1
bash tools/collect_dep_counters.sh $TAG >> gpurun_out/collect_${TAG}.log 2>&1
python3 bench.py > gpurun_out/bench_${TAG}_final.json 2> gpurun_out/bench_${TAG}_final.err
tail -c 300 gpurun_out/bench_${TAG}_final.json
