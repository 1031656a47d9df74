#!/bin/bash
# Builds dffinthewild_amd/libdffw_trace.so: the same sources with the phase timelines compiled in (make TRACE=1), in a scratch copy of
# csrc/ so that the production objects stay untouched.  Select it with DFFW_LIB_PATH=dffinthewild_amd/libdffw_trace.so.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=${TMPDIR:-/tmp}/dffw_trace_build
mkdir -p $tmp/pkg/csrc $tmp/include
cp -u $root/dffinthewild_amd/csrc/*.hip $root/dffinthewild_amd/csrc/*.cpp $root/dffinthewild_amd/csrc/*.h $root/dffinthewild_amd/csrc/Makefile $tmp/pkg/csrc/
cp -u $root/include/*.h $tmp/include/
make -s -C $tmp/pkg/csrc -j8 TRACE=1
cp $tmp/pkg/libdffw.so $root/dffinthewild_amd/libdffw_trace.so
echo built $root/dffinthewild_amd/libdffw_trace.so
