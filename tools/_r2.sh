mkdir -p gpurun_out/s4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k test_oracle_parity_fresh 2>&1 | tail -30 > gpurun_out/s4/ta.txt
DFFW_SMALL_MAX_UNITS=0 python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k test_oracle_parity_fresh 2>&1 | tail -5 > gpurun_out/s4/tb.txt
DFFW_SMALL_MAX_UNITS=0 timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/s4/tall.txt
