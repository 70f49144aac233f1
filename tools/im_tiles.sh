# scratch driver: configs[3] at full size with the gather's variants (tile layout on / off, batches, phases)
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "timstof" 2>&1 | tail -5
export N_PREC=200000 N_CYCLES=2000 SCAN_MAX=918 N_TOF=400000 EVENTS_PER_PUSH=30 ADH_BENCH_NO_CPU=1 STEPS=3 TOUCHED_SAMPLE=20
run() { name=$1; shift; env "$@" python tools/bench_timstof.py > gpurun_out/im_v_$name.json 2> gpurun_out/im_v_$name.log; }
run default A=1
run batches ADH_DEBUG_IM=14
run notiles ADH_DEBUG_IM_NO_TILES=1
run notiles_batches ADH_DEBUG_IM_NO_TILES=1 ADH_DEBUG_IM=14
run sel_only ADH_DEBUG_IM=7
for f in gpurun_out/im_v_*.json; do echo $f; python - $f <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(r['ms_per_step'], r['resident']['ms_per_step'], r['roofline']['gather_kernel_ms'], r['roofline']['feature_kernel_ms'], r['stage_seconds'], r['valid_fraction'], r['mean_matched_events'])
PY
done
