#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# Build morbit.jl_amd/libmrbf_<name>.so with one translation unit compiled with extra flags (same-box A/B of a kernel variant:
# MRBF_LIB=.../libmrbf_<name>.so python tools/...).   usage: tools/build_variant.sh <name> <file.hip> <flags...>
set -e
NAME=$1; FILE=$2; shift 2
cd "$(dirname "$0")/../morbit.jl_amd/csrc"
mkdir -p build_var
b=$(basename $FILE .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -I/opt/rocm/include -I. -I../../include "$@" -c $FILE -o build_var/${b}_$NAME.o
OBJS=""
for o in build/*.o; do
  if [ "$(basename $o .o)" = "$b" ]; then OBJS="$OBJS build_var/${b}_$NAME.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libmrbf_$NAME.so $OBJS -L/opt/rocm/lib -lrocblas -lrocsolver -Wl,-rpath,/opt/rocm/lib
ls -la ../libmrbf_$NAME.so
