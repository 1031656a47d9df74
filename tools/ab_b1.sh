#!/bin/bash
# A/B of environment variants on the batch-1 / batch-2 forwards: tools/ab_b1.sh NAME=ENV=V[,ENV=V] ...   (ms per forward, 3 alternating repetitions)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 0"
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}; envs=${envs//,/ }
    a=$(env $envs python bench.py --batch 1 $B 2>/dev/null | ms)
    b=$(env $envs python bench.py --batch 1 $B --slices 5 --size 224 2>/dev/null | ms)
    c=$(env $envs python bench.py --batch 2 $B 2>/dev/null | ms)
    echo "$name rep$rep  10x256x256: $a   5x224x224: $b   batch 2: $c"
  done
done
