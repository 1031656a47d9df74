#!/bin/bash
# Copy the round-end measurement set (gpurun_out/final, written by tools/round_measurements.sh) into profiles/ under
# round tag $1 (e.g. r01) and derive the HBM-traffic JSON bench.py reads.
set -e
tag=${1:-r01}; src=gpurun_out/final; dst=profiles
cp $src/bench_default.json $dst/${tag}_bench_default.json
cp $src/layers_default.tsv $dst/${tag}_layers_default.tsv
cp $src/rocprof/stats_kernel_stats.csv $dst/${tag}_rocprofv3_kernel_stats.csv
[ -f $src/rocprof_e2e/stats_kernel_stats.csv ] && cp $src/rocprof_e2e/stats_kernel_stats.csv $dst/${tag}_rocprofv3_kernel_stats_e2e.csv
for v in fp16 bf16 b1_10x256 b1_5x224 b8 torchrun1 u8; do cp $src/bench_$v.json $dst/${tag}_bench_$v.json; done
cp $src/bench_e2e.json $dst/${tag}_bench_e2e_b8_480x640.json
cp $src/bench_e2e_b1.json $dst/${tag}_bench_e2e_b1_480x640.json
cp $src/layers_e2e.tsv $dst/${tag}_layers_e2e_b8_480x640.tsv
python tools/hbm_traffic.py $src/pmc_fetch/pmc_counter_collection.csv $src/pmc_write/pmc_counter_collection.csv $src/bench_default.json $dst/${tag}_hbm_traffic.json
python tools/pmc_summary.py $src/pmc_fetch/pmc_counter_collection.csv $src/pmc_write/pmc_counter_collection.csv > $dst/${tag}_pmc_hbm_traffic.txt
python tools/pmc_summary.py $src/pmc_sq1/pmc_counter_collection.csv $src/pmc_sq2/pmc_counter_collection.csv > $dst/${tag}_pmc_conv_kernels.txt
python tools/pmc_derive.py $src/pmc_sq1/pmc_counter_collection.csv $src/pmc_sq2/pmc_counter_collection.csv >> $dst/${tag}_pmc_conv_kernels.txt
