mkdir -p gpurun_out/s5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/s5/t.txt
bash tools/round_measurements.sh > gpurun_out/s5/round.txt 2>&1
