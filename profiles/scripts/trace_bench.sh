#!/bin/bash
# rocprofv3 kernel trace (start / end timestamp, queue of every launch) of a short bench run -> gpurun_out/<tag>/...kernel_trace.csv.gz
# usage (on the GPU box, from the repo root): bash profiles/scripts/trace_bench.sh <tag> [bench.py args...]
tag=${1:-trace}; shift
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/$tag.log 2>&1
find gpurun_out/$tag -name "*kernel_trace.csv" -exec gzip -9 {} \;
find gpurun_out/$tag -type f ! -name "*kernel_trace.csv.gz" -delete
tail -2 gpurun_out/$tag.log
