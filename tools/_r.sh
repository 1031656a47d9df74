mkdir -p gpurun_out/s5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|ERROR" > gpurun_out/s5/t.txt
bash tools/round_measurements.sh > gpurun_out/s5/round.txt 2>&1
