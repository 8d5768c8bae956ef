#!/bin/bash
# Builds oracle/_ref/ref_driver: the reference's own translation units that need neither SDL nor OpenEXR, compiled
# UNMODIFIED from where they lie under /root/reference/src (no stand-in headers, nothing copied into this repo), plus
# our driver oracle/ref_driver.cpp.  Build-container only; outputs go to oracle/_ref/ (git-ignored).
# Flags = the reference's own build (CMakeLists.txt: C++17, default optimisation) as the survey measured it:
# g++ -std=c++17 -O2, baseline x86-64 (SSE2 doubles, no FMA).
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
[ -d "$REF/src" ] || { echo "ref_build: $REF/src not present (GPU box?) -- nothing to do"; exit 0; }
mkdir -p "$OUT/obj"
UNITS="vector matrix color colors transform geometry utils KDTree bbox camera light shading shadinghelper texture environment scene"
for u in $UNITS; do
    if [ ! -f "$OUT/obj/$u.o" ] || [ "$REF/src/$u.cpp" -nt "$OUT/obj/$u.o" ]; then
        g++ -std=c++17 -O2 -c "$REF/src/$u.cpp" -o "$OUT/obj/$u.o"
    fi
done
OBJS=$(for u in $UNITS; do echo "$OUT/obj/$u.o"; done)
# symbols defined only in the units that cannot be built here (Raytrace, Mesh, Heightfield, Bitmap, GetFrameWidth ...)
# stay unresolved; the driver never reaches them (see ref_driver.cpp)
g++ -std=c++17 -O2 -no-pie -I"$REF/src" "$HERE/ref_driver.cpp" $OBJS -Wl,--unresolved-symbols=ignore-all -o "$OUT/ref_driver"
echo "ref_build: built $OUT/ref_driver"
