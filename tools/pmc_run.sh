#!/bin/bash
# usage: tools/pmc_run.sh <tag> <counters...>   (one rocprofv3 --pmc pass over a small bench run)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o pmc -- python3 bench.py --steps 1 --warmup 1 --batch ${PMC_BATCH:-8} --no-cpu-baseline --no-roofline > gpurun_out/pmc_$tag.log 2>&1
ls gpurun_out/pmc_$tag | head
