#!/bin/bash
# ON THE GPU BOX: time the phased dense1 kernel built with each experiment flag (see vtcnn2_bf16_dense1.hip).
R=$PWD
for f in "" "-DD1_NOPRIO" "-DD1_NOSTAGGER"; do
  python3 -c "import sys; sys.path.insert(0,'$R'); from modulationdetectioncnn_amd import build as b; b.build(force=True, extra_flags=[x for x in ['$f'] if x])" > /dev/null 2>&1
  echo "flag [$f]: $(python3 $R/tools/prof_conv.py bf16 65536 prof 2>&1 | tail -1)"
done
python3 -c "import sys; sys.path.insert(0,'$R'); from modulationdetectioncnn_amd import build as b; b.build(force=True)" > /dev/null 2>&1
