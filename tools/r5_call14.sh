#!/bin/bash
for v in 32_128_256 32_256_512 16_128_512; do echo "== W_B_T = $v"; MRBF_LIB=$(pwd)/morbit.jl_amd/variants/libmrbf_$v.so python3 tools/ps_bench2.py 128,256 2>&1 | tail -6 | cut -c1-120; done
