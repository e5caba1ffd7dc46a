#!/bin/bash
# Builds the two diagnostic libraries of scratch/micro/img3_variants.hip (build container; the .so files travel with gpurun) and
# writes their ISA signatures to profiles/r04_hazard_isa.txt.
set -e
cd "$(dirname "$0")"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
mkdir -p /tmp/i3
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared img3_variants.hip -o libimg3v_packed.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $NOPK img3_variants.hip -o libimg3v_scalar.so 2> >(grep -v "not a recognized feature" >&2)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only img3_variants.hip -o /tmp/i3/packed.s 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 $NOPK -S --cuda-device-only img3_variants.hip -o /tmp/i3/scalar.s 2>/dev/null
python3 isa_img3_report.py /tmp/i3/packed.s /tmp/i3/scalar.s
