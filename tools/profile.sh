#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + the two PMC passes for
# the bench workload, into gpurun_out/prof_<tag>/.  Counters are collected in their
# own runs (no trace domains combined with --pmc).  Usage: tools/profile.sh <tag> [bench args]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-extras --no-cpu "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 20 --warmup 2 --no-extras --no-cpu --no-graph "$@" > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 2 --no-extras --no-cpu --no-graph "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 20 --warmup 2 --no-extras --no-cpu --no-graph "$@" > $OUT/pmc_sq.log 2>&1
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line.json
echo "profile $TAG done"
