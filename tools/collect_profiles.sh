set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
python bench.py > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err
tail -c 600 gpurun_out/prof/bench.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python bench.py > gpurun_out/prof/kt.log 2>&1
rocprofv3 -i tools/pmc_hbm.txt --kernel-trace --output-format csv -d gpurun_out/prof/pmc -- python tools/conv_bench.py --B 24 --iters 3 > gpurun_out/prof/pmc.log 2>&1
rocprofv3 -i tools/pmc_hbm.txt --kernel-trace --output-format csv -d gpurun_out/prof/pmcw -- python tools/conv_bench.py --B 44 --k 1 --H 10 --W 15 --z 64 --iters 3 > gpurun_out/prof/pmcw.log 2>&1
find gpurun_out/prof -name "*.csv" | head -30
