#!/bin/bash
# gpurun_out/r3 (scratch, merged back from the GPU box) -> profiles/ (tracked)
S=gpurun_out/r3
cp $S/bench.json profiles/r3_bench.json
cp $S/bench_b24.json profiles/r3_bench_b24.json
cp $S/bench_b44.json profiles/r3_bench_b44.json
cp $S/bench_fp32_mfma.json profiles/r3_bench_fp32_mfma.json
cp $S/bench_kernel_stats.csv profiles/r3_bench_kernel_stats.csv
cp $S/train_step_kernel_stats.csv profiles/r3_train_step_kernel_stats.csv
cp $S/kernel_trace_dominant.json profiles/r3_kernel_trace_dominant.json
cp $S/pmc_summary.csv profiles/r3_pmc_summary.csv
cp $S/pmc_derived.csv profiles/r3_pmc_derived.csv
cp $S/pmc_derived_b24.csv profiles/r3_pmc_derived_b24.csv
cp $S/pmc_derived_b44.csv profiles/r3_pmc_derived_b44.csv
cp $S/traffic.json profiles/traffic.json
cp $S/latency_tile_forms.txt profiles/r3_latency_tile_forms.txt
cp $S/b8_kernel_stats.csv profiles/r3_b8_kernel_stats.csv
cp $S/stem_bench.txt profiles/r3_stem_bench.txt
