#!/bin/bash
# Build the library from a git revision's native sources into face-recognition-platform_amd/libfrp_base.so
# (same-box A/B partner: FRP_LIB=face-recognition-platform_amd/libfrp_base.so python bench.py ...).
#   tools/ab_lib.sh [rev]     (default HEAD)
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" face-recognition-platform_amd/csrc include | tar -x -C "$tmp"
make -s -j8 -C "$tmp/face-recognition-platform_amd/csrc"
cp "$tmp/face-recognition-platform_amd/libfrp.so" "$root/face-recognition-platform_amd/libfrp_base.so"
rm -rf "$tmp"
echo "built $root/face-recognition-platform_amd/libfrp_base.so from $rev"
