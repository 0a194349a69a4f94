#!/bin/bash
# Everything a reader needs to recompute the bench line's fractions for one config, into profiles/ (round tag $TAG, default r05):
#   ${TAG}_kernel_stats_<cfg>.csv  rocprofv3 --kernel-trace --stats of the bench command (per-kernel calls / average duration)
#   ${TAG}_traffic_<cfg>.json      per-launch HBM traffic from two separate PMC passes (FETCH_SIZE, WRITE_SIZE)
#   ${TAG}_bench_<cfg>.json        the bench line itself (+ ${TAG}_ops_<cfg>.json: per-launch isolated times, bytes, flops)
# usage (GPU box, repo root): bash profiles/scripts/profile_config.sh <cfg>      -> files under gpurun_out/prof_<cfg>/ (copy to profiles/)
cfg=${1:-en_s0}
TAG=${TAG:-r05}
root="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
cd "$root"
o=gpurun_out/prof_$cfg
mkdir -p $o
python bench.py --config $cfg --steps 20 --warmup 5 --ops-json $o/${TAG}_ops_$cfg.json > $o/${TAG}_bench_$cfg.json 2> $o/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > $o/stats.log 2>&1
cp $(find $o/stats -name "*kernel_stats.csv" | head -1) $o/${TAG}_kernel_stats_$cfg.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/pmc_fetch -- python profiles/scripts/pmc_forward.py --config $cfg --ops-json $o/pmc_ops.json > $o/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/pmc_write -- python profiles/scripts/pmc_forward.py --config $cfg --ops-json $o/pmc_ops.json > $o/pmc_write.log 2>&1
python profiles/scripts/pmc_ops.py $o/pmc_fetch $o/pmc_write $o/pmc_ops.json $o/${TAG}_traffic_$cfg.json
rm -rf $o/stats $o/pmc_fetch $o/pmc_write
ls -la $o
