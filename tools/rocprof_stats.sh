#!/bin/bash
# usage: tools/rocprof_stats.sh <tag>   kernel-trace + stats of the default bench command (no CPU leg), CSV output
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rocprof_$tag -o $tag -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/rocprof_$tag.json 2> gpurun_out/rocprof_$tag.err
ls gpurun_out/rocprof_$tag
