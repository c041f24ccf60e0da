#!/bin/bash
export MRBF_EXPERIMENTS=1   # the library honours its MRBF_* switches only behind this gate
# Build morbit.jl_amd/libmrbf_prev.so from the committed (HEAD) version of the given csrc files and the current objects of the rest:
# same-box A/B of a kernel change (MRBF_LIB=.../libmrbf_prev.so python tools/...).   usage: tools/build_prev.sh chol_mega.hip [more.hip]
set -e
cd "$(dirname "$0")/../morbit.jl_amd/csrc"
mkdir -p build_prev
OBJS=""
for o in build/*.o; do
  b=$(basename $o .o); use=$o
  for f in "$@"; do
    if [ "$b.hip" = "$f" ]; then
      git show HEAD:morbit.jl_amd/csrc/$f > build_prev/$f   # headers come from the working tree (-I.)
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -I/opt/rocm/include -I. -c build_prev/$f -o build_prev/$b.o
      use=build_prev/$b.o
    fi
  done
  OBJS="$OBJS $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libmrbf_prev.so $OBJS -L/opt/rocm/lib -lrocblas -lrocsolver -Wl,-rpath,/opt/rocm/lib
ls -la ../libmrbf_prev.so
