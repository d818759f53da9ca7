#!/bin/bash
# rocprofv3 passes over tools/allocbench pmc: kernel times and the L2 -> fabric write counters per backing (one process per pass,
# so the physical placement differs between passes: compare backings INSIDE a pass, and read times from the same pass' trace).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/allocpmc_${1:-a}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $R/tools/allocbench $R/tools/libaesw_diag.so pmc > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pmc1 -- $R/tools/allocbench $R/tools/libaesw_diag.so pmc > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum --output-format csv -d $OUT/pmc2 -- $R/tools/allocbench $R/tools/libaesw_diag.so pmc > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum --output-format csv -d $OUT/pmc3 -- $R/tools/allocbench $R/tools/libaesw_diag.so pmc > $OUT/pmc3.log 2>&1
find $OUT -name "*.csv" | head -30
echo done
