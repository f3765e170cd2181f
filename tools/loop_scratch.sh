#!/bin/bash
# Report scratch/vmcnt instructions inside the deepest loop of the product bf16 conv kernel.
cd /tmp/t && hipcc -O3 -std=c++17 --offload-arch=gfx950 -I /root/repo/include -I /root/repo/modulationdetectioncnn_amd/csrc -c /root/repo/modulationdetectioncnn_amd/csrc/vtcnn2_bf16.hip -o vb.o -save-temps 2>/dev/null
S=vtcnn2_bf16-hip-amdgcn-amd-amdhsa-gfx950.s
a=$(grep -n "^_ZN3mdc.*vt_conv_bf16_kernelILi0.*:" $S | head -1 | cut -d: -f1); b=$(awk -v s=$a 'NR>s && /s_endpgm/{print NR; exit}' $S); sed -n "${a},${b}p" $S > conv.s
e=$(grep -n "s_cbranch_scc1" conv.s | tail -1 | cut -d: -f1)
st=$(grep -n "Depth=2" conv.s | head -1 | cut -d: -f1)
echo "inner loop lines $st..$e of $(wc -l < conv.s)"
sed -n "${st},${e}p" conv.s > loop.s
echo "scratch in loop: $(grep -c scratch_ loop.s); vmcnt waits: $(grep -c vmcnt loop.s); accvgpr: $(grep -c v_accvgpr loop.s); mfma: $(grep -c v_mfma loop.s); nop: $(grep -c s_nop loop.s)"
grep -n "scratch_\|vmcnt" loop.s | head -${1:-12}
