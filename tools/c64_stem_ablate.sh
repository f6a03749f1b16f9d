#!/bin/bash
# Timing ablations of the fused stem of conv3x3_c64.hip (STEM): lab libraries with -DC64S_ABL=<bits> (1: no stem phase, 2: no stores of the
# stem map, 4: no patch writes, 8: no output stores), then the fused launch's average duration in a kernel trace of the bench under each.
#   tools/c64_stem_ablate.sh build "1 2 8 10"   (here: cross-compiles)        tools/c64_stem_ablate.sh run "base 1 2 8 10"   (on the GPU box)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/face-recognition-platform_amd/csrc
if [ "$1" = build ]; then
    make -C $C lab > /dev/null
    for a in $2; do
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -Wall -Wno-unused-function -DFRP_LAB -DC64S_ABL=$a -c $C/conv3x3_c64.hip -o /tmp/c64s_$a.o
        /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls $C/build_lab/*.o | grep -v conv3x3_c64.o) /tmp/c64s_$a.o -o $R/face-recognition-platform_amd/libfrp_lab_abl$a.so
    done
    exit 0
fi
cd /tmp && export TMPDIR=/tmp
for a in $2; do
    if [ $a = base ]; then export FRP_LIB=$R/face-recognition-platform_amd/libfrp_lab.so; else export FRP_LIB=$R/face-recognition-platform_amd/libfrp_lab_abl$a.so; fi
    rm -rf /tmp/kt_$a
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$a -o kt -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-frames 0 --lanes 1 --pcie-steps 0 --threshold-steps 0 > /dev/null 2>&1
    echo "C64S_ABL=$a $(grep 'c64_kernel<2, false, true, 16, true>' $(find /tmp/kt_$a -name '*kernel_stats.csv') | head -1 | awk -F, '{print $(NF-4)/1000 " us avg"}')"
done
