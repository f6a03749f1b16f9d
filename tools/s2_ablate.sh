#!/bin/bash
# Timing ablations of the stride-2 kernel: lab libraries with -DS2_ABL=<bits> (conv3x3_s2.hip), tools/s2_probe.py on each.
#   tools/s2_ablate.sh build "1 2 4 8 16 3"    (here: cross-compiles)      tools/s2_ablate.sh run "1 2 ..." [iters]   (on the GPU box)
set -e
R=$(cd $(dirname $0)/.. && pwd)
C=$R/face-recognition-platform_amd/csrc
if [ "$1" = build ]; then
    make -C $C lab > /dev/null
    for a in $2; do
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -Wall -Wno-unused-function -DFRP_LAB ${S2_DEF:--DS2_ABL=}$a -c $C/conv3x3_s2.hip -o /tmp/s2_abl$a.o
        /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls $C/build_lab/*.o | grep -v conv3x3_s2.o) /tmp/s2_abl$a.o -o $R/face-recognition-platform_amd/libfrp_lab_abl$a.so
    done
else
    for a in $2; do
        echo "## S2_ABL=$a"
        FRP_LIB=$R/face-recognition-platform_amd/libfrp_lab_abl$a.so python3 $R/tools/s2_probe.py ${3:-10}
    done
fi
