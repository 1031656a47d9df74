#!/bin/bash
# Builds dffinthewild_amd/libdffw_<tag>.so = the production objects with ONE source recompiled under extra defines (dev-only ablations and A/B variants;
# select with DFFW_LIB_PATH).  usage: tools/build_variant_lib.sh <tag> <source.hip> -DNAME=VALUE ...      (run `make` in csrc/ first)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; src=$2; shift 2
c=$root/dffinthewild_amd/csrc
obj=${TMPDIR:-/tmp}/dffw_variant_$tag.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -x hip -c $c/$src -o $obj
objs=""
for o in $(sed -n 's/^OBJS = //p' $c/Makefile); do
    if [ "$o" = "${src%.*}.o" ]; then objs="$objs $obj"; else objs="$objs $c/$o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -o $root/dffinthewild_amd/libdffw_$tag.so
echo built dffinthewild_amd/libdffw_$tag.so
