#!/bin/bash
# Round-end measurement set on one MI355X (run through gpurun); writes into gpurun_out/final/
set -u
out=gpurun_out/final; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python bench.py --no-cpu-baseline --no-other-configs --dump-layers $out/layers_default.tsv > $out/bench_layers.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $out/bench_torchrun1.json 2> $out/bench_torchrun1.err
for p in fp16 bf16; do python bench.py --precision $p --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_$p.json 2>/dev/null; done
python bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > $out/bench_b1_10x256.json 2>/dev/null
python bench.py --batch 1 --slices 5 --size 224 --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > $out/bench_b1_5x224.json 2>/dev/null
python bench.py --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $out/bench_b8.json 2>/dev/null
python bench.py --input u8 --steps 10 --warmup 3 > $out/bench_u8.json 2>/dev/null
python bench.py --workload e2e --dump-layers $out/layers_e2e.tsv > $out/bench_e2e.json 2>/dev/null
python bench.py --workload e2e --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $out/bench_e2e_b1.json 2>/dev/null
# kernel durations: with the pyramid branches serialised, as in bench.py's own profiled forward (side by side, three small kernels
# share the chip and each one's duration says nothing about the kernel)
export DFFW_NO_CONCURRENT=1 DFFW_NO_PROBE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof -o stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --sustain-seconds 0 > $out/rocprof_bench.json 2> $out/rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof_e2e -o stats -- python3 bench.py --workload e2e --steps 5 --warmup 2 --no-cpu-baseline --sustain-seconds 0 > $out/rocprof_bench_e2e.json 2> $out/rocprof_e2e.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 0 > /dev/null 2>&1
# wave-state / matrix-pipe / LDS counters of the conv kernels (two passes: 8 SQ slots each), batch 32
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq1 -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq2 -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 0 > /dev/null 2>&1
rocprofv3 --pmc TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/pmc_ta -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 0 > /dev/null 2>&1
unset DFFW_NO_CONCURRENT DFFW_NO_PROBE
for f in $out/bench_*.json; do echo "$f: $(cut -c1-110 $f)"; done
