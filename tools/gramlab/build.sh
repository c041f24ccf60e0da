#!/bin/bash
# builds the lab tools and prints register / spill figures of the il kernel; run from anywhere
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -save-temps=obj -o gramlab gramlab.hip 2>&1 | grep -E "error" -A8 | head -30
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o ovl ovl.hip 2>&1 | grep -E "error" -A8 | head
S=gramlab-hip-amdgcn-amd-amdhsa-gfx950.s
for k in ${1:-_Z7gram_ilILi3}; do
  grep -A12 "name:.*$k" $S | grep -E "vgpr_count|vgpr_spill|private_segment"
  awk "/^$k/,/s_endpgm/" $S > /tmp/$k.s
  echo "lines $(wc -l < /tmp/$k.s) scratch $(grep -c scratch_ /tmp/$k.s) mfma $(grep -c v_mfma /tmp/$k.s)"
  echo "scratch at: $(grep -n scratch_ /tmp/$k.s | cut -d: -f1 | tr '\n' ' ')"
  echo "mfma blocks at: $(grep -n v_mfma /tmp/$k.s | awk -F: 'NR%68==1{print $1}' | tr '\n' ' ')"
done
rm -f gramlab-* *.hipfb
