#!/bin/bash
python3 tools/ps_bench2.py 64 2>&1 | tail -3 | cut -c1-120
MRBF_PS_MULTI=0 python3 tools/ps_bench2.py 64 2>&1 | tail -3 | cut -c1-120
