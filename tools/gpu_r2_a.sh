# round-2 measurement batch A: phase timings of the K=512 GEMM kernels, per-op times at small and large batches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
for args in "--k 1 --B 44" "--k 1 --B 44 --stats" "--k 1 --B 44 --H 10 --W 15 --z 64"; do
  echo "== conv_bench $args" ; python tools/conv_bench.py $args --iters 20 2>&1 | tail -2
  echo "== XL_CONV_CLK conv_bench $args"; XL_CONV_CLK=1 python tools/conv_bench.py $args --iters 1 2>&1 | grep clk | tail -2
done > gpurun_out/r2a/clk.log 2>&1
for B in 3 5 8 44; do
  echo "== bench verbose B=$B"; XL_BENCH_VERBOSE=1 python bench.py --batch $B --no-secondary --no-cpu-baseline --steps 10 2>&1 | grep -v "^{" | tail -120
done > gpurun_out/r2a/verbose.log 2>&1
tail -30 gpurun_out/r2a/clk.log
