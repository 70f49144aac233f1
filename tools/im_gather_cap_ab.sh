export N_PREC=200000 N_CYCLES=2000 SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 STEPS=3 TOUCHED_SAMPLE=5 TOUCHED_SAMPLE_SEL=2 ADH_BENCH_NO_SELECT=1
for v in default ${VARIANTS:-gcc gca gcb}; do
  if [ "$v" = default ]; then unset ADH_LIB_PATH; else export ADH_LIB_PATH=$PWD/alphadia_amd/libalphadia_hip_$v.so; fi
  python tools/bench_timstof.py > /tmp/gcab.json 2> /tmp/gcab.log
  python - /tmp/gcab.json $v <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf=r['roofline']
print(sys.argv[2], 'h2h %.2f gather %.3f ms features %.3f valid %.3f' % (r['ms_per_step'], rf['gather_kernel_ms'], rf['feature_kernel_ms'], r['valid_fraction']))
PY
done
