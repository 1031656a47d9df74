mkdir -p gpurun_out/s4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/lds_bank_probe.hip -o /tmp/lds_bank_probe && /tmp/lds_bank_probe > gpurun_out/s4/lds_probe.txt 2>&1
