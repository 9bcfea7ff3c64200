#!/bin/bash
# gpurun_out/r5 (scratch, merged back from the GPU box) -> profiles/ (tracked)
S=gpurun_out/r5
for f in bench bench_b47 bench_b24 bench_streams2 bench_mlr3 kernel_trace_dominant; do cp $S/$f.json profiles/r5_$f.json; done
for f in bench_kernel_stats train_step_kernel_stats mlr3_kernel_stats pmc_summary pmc_derived pmc_derived_b47; do cp $S/$f.csv profiles/r5_$f.csv; done
cp $S/traffic.json profiles/traffic.json
