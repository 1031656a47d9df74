mkdir -p gpurun_out/s4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 0 512 768 1536; do
DFFW_SRD_WGS=$w python bench.py --workload e2e --no-cpu-baseline --steps 5 --warmup 2 --dump-layers gpurun_out/s4/layers_$w.tsv > gpurun_out/s4/bench_$w.json 2>gpurun_out/s4/err_$w.txt
done
