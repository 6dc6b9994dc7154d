#!/bin/bash
# Dev tool: build an experimental variant of the library with extra -D flags for ONE source file.
#   tools/build_variant.sh <name> <source.hip> [-DFOO=1 ...]   ->  geoformer_amd/lib/exp/<name>.so
# then run with GF_LIB_PATH=geoformer_amd/lib/exp/<name>.so
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/geoformer_amd/lib/exp; mkdir -p $out/obj_$name
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-gpu-rdc -Wno-unused-result \
  -I$root/include -I$root/geoformer_amd/csrc "$@" -c $root/geoformer_amd/csrc/$base.hip -o $out/obj_$name/$base.o
objs=""
for o in $root/geoformer_amd/lib/obj/*.o; do b=$(basename $o); if [ "$b" == "$base.o" ]; then objs="$objs $out/obj_$name/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $objs
echo $out/$name.so
