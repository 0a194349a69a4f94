#!/bin/bash
# Everything under profiles/ for one round, on one box: per-config kernel stats + PMC traffic + bench lines (profile_config.sh), the
# bench variants of the headline config, the training-step record.  usage: bash profiles/scripts/round_artifacts.sh   -> gpurun_out/
root="${GRAFT_REPO_ROOT:-/root/repo}"
export TAG=${TAG:-r05}
cd "$root"; mkdir -p gpurun_out/variants
for cfg in en_s0 mv_s2 en_s2 en_s0_pn2 en_s0_cdf; do bash profiles/scripts/profile_config.sh $cfg > gpurun_out/profile_$cfg.log 2>&1; done
b() { name=$1; shift; python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric' | tail -1 > gpurun_out/variants/${TAG}_bench_$name.json; }
b en_s0_dense_radar --dense-radar
b en_s0_storage_bf16 --storage bf16
b en_s0_io_f16 --dtype f16
b en_s0_neck_launches_separate --opt ghost_fuse=0 --opt ds_fuse=0
b en_s0_r03_kernels --storage bf16 --opt ghost_fuse=0 --opt ds_fuse=0
b en_s0_pn2_plain --config en_s0_pn2 --plain
b en_s1 --config en_s1
b en_s0_dense_radar_noskip --dense-radar --opt radar_skip=0
b en_s0_noskip --opt radar_skip=0
b en_s0_cdf_layerwise --config en_s0_cdf --opt csp_fuse=0
b en_s0_cdf_last_level_only --config en_s0_cdf --opt csp_fuse=1
b en_s0_f32 --dtype f32
b en_s0_b1 --batch 1
b en_s0_b8 --batch 8
b en_s0_plain --plain
b en_s0_separate_calls --separate-calls
b en_s0_force_collective --force-collective
b en_s0_head_tile --opt head_rows=0
b en_s0_mlp_tile --opt mlp_band=0
b en_s0_r02_kernels --opt head_rows=0 --opt mlp_band=0 --opt head_fuse=0 --opt radar_compact=0 --opt level_chain=0 --opt sdta_fuse=0
b en_s0_levels_separate --opt level_chain=0
b en_s0_sdta_separate --opt sdta_fuse=0
b en_s0_sdta_all --opt sdta_fuse=2
b en_s0_head_layers_separate --opt head_fuse=0
b en_s0_radar_segments --opt radar_compact=0
b en_s0_radar_copy --opt radar_direct=0
b en_s0_head_band80 --opt head_band=80
b en_s0_spp_launches_separate --opt ghost_fuse=0
b mv_s2_image_copy --config mv_s2 --opt mv_stem=0
b en_s0_b256 --batch 256
b en_s2_b256 --config en_s2 --batch 256
b en_s2_b512_one_gpu --config en_s2 --batch 512 --plain
PYTHONPATH=. python profiles/scripts/train_step.py --batch 8 --steps 5 --baseline > gpurun_out/variants/${TAG}_train_step_b8.json 2>/dev/null
PYTHONPATH=. python profiles/scripts/train_step.py --batch 32 --steps 3 --baseline > gpurun_out/variants/${TAG}_train_step_b32.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd "$root"
PYTHONPATH=. rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train_prof -- python profiles/scripts/train_step.py --batch 8 --steps 2 > gpurun_out/train_prof.log 2>&1
cp $(find gpurun_out/train_prof -name "*kernel_stats.csv" | head -1) gpurun_out/variants/${TAG}_train_kernel_stats.csv; rm -rf gpurun_out/train_prof
ls gpurun_out/variants gpurun_out/prof_*
