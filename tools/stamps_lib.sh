#!/bin/bash
# Builds <package>/libafhip_stamps.so: the shipped objects, with the named kernel files recompiled with -DAF_STAMPS (in-kernel
# s_memtime stamps written to the buffer whose address is in AF_STAMP_PTR).  Diagnostics only; run with AF_HIP_LIB=<that file>.
#   bash tools/stamps_lib.sh af_conv133g [af_conv ...]
cd "$(dirname "$0")/.."; PKG=$(ls -d spatiotemporal*_amd); mkdir -p $PKG/build/stamps
bash tools/build.sh > /dev/null || exit 1
OBJS=""
for o in $PKG/build/*.o; do
  b=$(basename $o .o); use=$o
  for f in "$@"; do
    if [ "$f" = "$b" ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-inline-asm -DAF_STAMPS -c $PKG/csrc/$b.hip -o $PKG/build/stamps/$b.o || exit 1
      use=$PKG/build/stamps/$b.o
    fi
  done
  OBJS="$OBJS $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libafhip_stamps.so $OBJS && echo $PKG/libafhip_stamps.so
