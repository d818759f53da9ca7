#!/bin/bash
# rocprofv3 passes over examples/aesw_check.c (plain C: generate 2^20 blocks with per-block keys, check every constraint on the
# device three times): kernel trace + stats, then the LDS / fetch counters of check_kernel in their own runs.
# Usage (from the repo root): gpurun -- 'bash tools/profile_check.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_check
mkdir -p $OUT
gcc -O2 -std=c11 -D__HIP_PLATFORM_AMD__ -I $R/include -I /opt/rocm/include $R/examples/aesw_check.c -o $OUT/aesw_check \
    -L $R/halo2-aes_amd -laesw -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$R/halo2-aes_amd -Wl,-rpath,/opt/rocm/lib || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $OUT/aesw_check 20 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- $OUT/aesw_check 20 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $OUT/aesw_check 20 > $OUT/pmc_fetch.log 2>&1
echo "profile check done"
