#!/bin/bash
# PMC of the ion-mobility kernels on the reduced config-4 bench (GPU box, via gpurun)
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for mode in 0 8; do
  rm -rf /tmp/im_pmc$mode
  ADH_DEBUG_IM=$mode ADH_BENCH_NO_CPU=1 STEPS=2 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_LDS -d /tmp/im_pmc$mode -o p -- python $REPO/tools/bench_timstof.py > /tmp/im_pmc$mode.log 2>&1
  echo "== ADH_DEBUG_IM=$mode"
  python $REPO/tools/rocpd_summary.py /tmp/im_pmc$mode/p_results.db | grep "adh_gather_im\|adh_feature_im" | sed 's/(DevTims[^)]*)//' | cut -c1-100
done
