mkdir -p gpurun_out/s4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for u in 0 128 256 512 1024 2048 4096; do
DFFW_SMALL_MAX_UNITS=$u python bench.py --batch 1 --steps 100 --warmup 20 --no-cpu-baseline --no-roofline > gpurun_out/s4/b1_$u.json 2>/dev/null
DFFW_SMALL_MAX_UNITS=$u python bench.py --batch 1 --slices 5 --size 224 --steps 100 --warmup 20 --no-cpu-baseline --no-roofline > gpurun_out/s4/b1s_$u.json 2>/dev/null
done
DFFW_SMALL_MAX_UNITS=1024 python bench.py --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --dump-layers gpurun_out/s4/layers_b1_1024.tsv > /dev/null 2>&1
