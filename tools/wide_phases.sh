#!/bin/bash
# cumulative time of the phases of the wide register kernels (ADH_DEBUG_STOP_PHASE stops of adh_fast_body) on the
# transfer-requantification leg
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for p in ${PHASES:-31 32 33 4 5 6 61 62 0}; do
  rm -rf /tmp/widep_stats
  CPU_SAMPLE=1000 ADH_DEBUG_STOP_PHASE=$p rocprofv3 --kernel-trace --stats -d /tmp/widep_stats -o p -- python $REPO/tools/bench_legs.py transfer > /tmp/widep_stats.log 2>&1
  echo "stop $p: $(python $REPO/tools/rocpd_summary.py /tmp/widep_stats/p_results.db | grep 'wide_kernel' | awk -F, '{printf "%s calls %.1f us avg; ", $(NF-5), $(NF-3)/1e3}')"
done
