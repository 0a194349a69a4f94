cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -s -k "16bit_matches_reference_fixtures or csp_fused or bf16_storage_engine or batch64" 2>&1 | grep -v Warning > gpurun_out/r05_parity_table.txt
export TAG=r05
for cfg in mv_s2 en_s0_cdf; do timeout 600 bash profiles/scripts/profile_config.sh $cfg > gpurun_out/profile_$cfg.log 2>&1; done
mkdir -p gpurun_out/variants
b() { name=$1; shift; python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric' | tail -1 > gpurun_out/variants/r05_bench_$name.json; }
b en_s0_cdf_layerwise --config en_s0_cdf --opt csp_fuse=0
b en_s0_cdf_last_level_only --config en_s0_cdf --opt csp_fuse=1
b mv_s2_image_copy --config mv_s2 --opt mv_stem=0
b mv_s2_ffn_one_tile --config mv_s2 --opt ffn_rows2=0
ls gpurun_out/variants | wc -l
