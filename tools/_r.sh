mkdir -p gpurun_out/s3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_e2e.py tests/test_fov_warp.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/s3/t.txt
python bench.py --workload e2e --no-cpu-baseline --dump-layers gpurun_out/s3/layers_e2e.tsv > gpurun_out/s3/bench_e2e.json 2>gpurun_out/s3/bench_e2e.err
