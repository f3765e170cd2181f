#!/bin/bash
# ON THE GPU BOX: timing-only probes of the fp8 conv kernel's step (vtcnn2_fp8_conv.hip, #ifdef F8P): what would the step
# cost without its partial-sum adds (the K-quarter chain's gain), without the cross-wave partial reads, without the
# conv1 pack, with MFMAs only?  Results are wrong by construction; only the kernel time is read.  Rebuilds the product
# library without the flag at the end.
R=$PWD
for f in "" "-DF8P=1" "-DF8P=2" "-DF8P=3" "-DF8P=5"; do
  python3 -c "import sys; sys.path.insert(0,'$R'); from modulationdetectioncnn_amd import build as b; b.build(force=True, extra_flags=[x for x in ['$f'] if x])" > /dev/null 2>&1 || { echo "build failed for [$f]"; exit 1; }
  echo "flag [$f]: $(timeout -k 10 120 python3 $R/tools/time_vt.py fp8 4 2>&1 | tail -1)"
done
python3 -c "import sys; sys.path.insert(0,'$R'); from modulationdetectioncnn_amd import build as b; b.build(force=True)" > /dev/null 2>&1
