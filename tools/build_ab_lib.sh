#!/bin/bash
# Builds build_ab/libdffw_<name>.so from the csrc/ of a git ref (default HEAD) in a scratch directory: an A/B partner for the working tree's
# library on the same GPU box (select it with DFFW_LIB_PATH=build_ab/libdffw_<name>.so; tools/ab_layers_min.py takes it as a variant's env).
# build_ab/ is git-ignored; delete it before the round ends so that it does not travel with every gpurun call.
set -e
name=${1:-base}; ref=${2:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=${TMPDIR:-/tmp}/dffw_ab_$name
rm -rf $tmp; mkdir -p $tmp
git -C $root archive $ref dffinthewild_amd/csrc include | tar -x -C $tmp
make -s -C $tmp/dffinthewild_amd/csrc -j8
mkdir -p $root/build_ab
cp $tmp/dffinthewild_amd/libdffw.so $root/build_ab/libdffw_$name.so
echo built $root/build_ab/libdffw_$name.so from $ref
