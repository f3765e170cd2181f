#!/bin/bash
# Print per-kernel register/LDS/scratch usage of one .hip file (hipcc -Rpass-analysis).
# usage: tools/kernel_resources.sh modulationdetectioncnn_amd/csrc/deployed.hip [extra flags]
src=$1; shift
hipcc -O3 -std=c++17 --offload-arch=gfx950 -I /root/repo/include -I /root/repo/modulationdetectioncnn_amd/csrc \
  -c "$src" -o /tmp/kr.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | \
  awk '/Function Name/{n=$0; sub(/.*Function Name: /,"",n); sub(/ \[-R.*/,"",n)} /VGPRs:|AGPRs:|ScratchSize|Occupancy|SGPRs:|LDS Size/{v=$0; sub(/.*remark: [^ ]* /,"",v); sub(/ \[-R.*/,"",v); printf "%s | %s\n", n, v}' | \
  awk -F' \\| ' '{a[$1]=a[$1] "  " $2} END{for(k in a) print k ":" a[k]}' | c++filt | sort
