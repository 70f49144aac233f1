#!/bin/bash
# executed instructions of the ion-mobility feature kernel up to each stop point (ADH_DEBUG_IM), per wavefront:
# one rocprofv3 --pmc run per stop on the reduced config-4 bench (GPU box, via gpurun)
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for mode in ${PHASES:-11 1 2 3 4 45 5 6 71 72 0}; do
  rm -rf /tmp/im_pmc$mode
  ADH_DEBUG_IM=$mode ADH_BENCH_NO_CPU=1 STEPS=2 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/im_pmc$mode -o p -- python $REPO/tools/bench_timstof.py > /tmp/im_pmc$mode.log 2>&1
  python $REPO/tools/rocpd_summary.py /tmp/im_pmc$mode/p_results.db | grep "adh_feature_im" | python -c "
import sys
d={}
for l in sys.stdin:
    p=l.rstrip().rsplit(',',4)
    d[p[1]]=d.get(p[1],0)+float(p[4])
w=d.get('SQ_WAVES',1)
print('stop $mode', ' '.join('%s %.0f' % (k.replace('SQ_',''), v/w) for k,v in sorted(d.items()) if k!='SQ_WAVES'))"
done
