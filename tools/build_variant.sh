#!/bin/bash
# tools/build_variant.sh NAME "FLAGS": a copy of zdw_amd/lib in zdw_amd/lib_ab_NAME whose pass-1 kernels are compiled with FLAGS
# (A/B switches of k_pass1.hip and its headers).  tools/ab_variants.sh times every such copy beside the shipped build.
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
out=$here/zdw_amd/lib_ab_$1
rm -rf "$out"; cp -r "$here/zdw_amd/lib" "$out"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $2 -c "$here/zdw_amd/csrc/k_pass1.hip" -o "$out/obj/k_pass1.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libzdw_hip.so" "$out"/obj/*.o -ldl
echo "built $out"
