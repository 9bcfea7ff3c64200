#!/bin/bash
# gpurun_out/r4 (scratch, merged back from the GPU box) -> profiles/ (tracked)
S=gpurun_out/r4
for f in bench bench_b24 bench_b47 bench_streams2 bench_mlr3 kernel_trace_dominant; do cp $S/$f.json profiles/r4_$f.json; done
for f in bench_kernel_stats train_step_kernel_stats mlr3_kernel_stats pmc_summary pmc_derived pmc_derived_b24 pmc_derived_b47; do cp $S/$f.csv profiles/r4_$f.csv; done
cp $S/traffic.json profiles/traffic.json
cp $S/wino_out_tpb.txt profiles/r4_wino_out_tpb.txt
