# scratch driver: selection kernel, cumulative stops (ADH_DEBUG_SELECT_STOP) and the generic smoothing loop
python -m pytest tests/test_selection.py -x -q -m gpu 2>&1 | tail -2
for st in 0 1 2 3 4 5; do
  echo "stop $st: $(ADH_BENCH_NO_CPU=1 ADH_DEBUG_SELECT_STOP=$st python tools/bench_select.py 2>/dev/null | tail -1 | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["kernel_ms"], r["host_call_ms"], r["candidates_found"])')"
done
echo "lds taps: $(ADH_BENCH_NO_CPU=1 ADH_DEBUG_SELECT_LDS_TAPS=1 python tools/bench_select.py 2>/dev/null | tail -1 | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["kernel_ms"], r["candidates_found"])')"
python tools/bench_select.py 2>/dev/null | tail -1
