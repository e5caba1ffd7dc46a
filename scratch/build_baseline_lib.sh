#!/bin/bash
# Build libwu_kernels.so of another git revision next to the tree's own, for same-box interleaved A/B timing
# (scratch/ab_lib.py): scratch/_oldlib/libwu_old.so.  Usage: scratch/build_baseline_lib.sh [rev]   (default HEAD)
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/scratch/_oldlib
rm -rf "$OUT" && mkdir -p "$OUT/a/csrc" "$OUT/include"
cd "$ROOT"
git show $REV:include/wu_kernels.h > "$OUT/include/wu_kernels.h"
for f in $(git ls-tree --name-only $REV weather-unet_amd/csrc/); do git show $REV:$f > "$OUT/a/csrc/$(basename $f)"; done
cd "$OUT/a/csrc"
ls *.hip | xargs -P 6 -I{} sh -c '/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -c {} -o {}.o'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libwu_old.so" *.o
rm -f *.o
echo "$REV" > "$OUT/rev.txt"
echo "built $OUT/libwu_old.so from $REV"
