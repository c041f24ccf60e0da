#!/bin/bash
# builds tools/walklab/walklab for gfx950 (run from the repo root)
set -e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/opt/rocm/include -Iinclude tools/walklab/walklab.hip -o tools/walklab/walklab
