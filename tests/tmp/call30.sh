set -x
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --ops-json gpurun_out/ops_final_en_s0.json > gpurun_out/bench_final_en_s0.json 2>gpurun_out/bench_err.log
python bench.py --config en_s2 --steps 30 --warmup 5 --no-cpu-baseline --ops-json gpurun_out/ops_final_en_s2.json > gpurun_out/bench_final_en_s2.json 2>>gpurun_out/bench_err.log
python bench.py --config mv_s2 --steps 30 --warmup 5 --no-cpu-baseline --ops-json gpurun_out/ops_final_mv_s2.json > gpurun_out/bench_final_mv_s2.json 2>>gpurun_out/bench_err.log
python bench.py --dtype f32 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_final_en_s0_f32.json 2>>gpurun_out/bench_err.log
python bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_final_en_s0_b1.json 2>>gpurun_out/bench_err.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o run -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o run -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
ls -la $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
cat $R/gpurun_out/bench_final_*.json | cut -c1-400
