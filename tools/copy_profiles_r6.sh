#!/bin/bash
# gpurun_out/r6 (scratch, merged back from the GPU box) -> profiles/ (tracked)
set -eu
S=gpurun_out/r6
for f in bench bench_b47 bench_b190 bench_streams2 bench_mlr3 bench_pingpong bench_store32 kernel_trace_dominant; do cp "$S/$f.json" "profiles/r6_$f.json"; done
for f in bench_kernel_stats train_step_kernel_stats mlr3_kernel_stats pmc_summary pmc_derived pmc_derived_b47 gemm_ab; do cp "$S/$f.csv" "profiles/r6_$f.csv"; done
for f in gemm_ab pair_clk dropin_breakdown b1_timeline stem_pmc stem12_wgs; do cp "$S/$f.txt" "profiles/r6_$f.txt"; done
cp "$S/traffic.json" profiles/traffic.json
