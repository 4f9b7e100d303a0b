#!/bin/bash
# end-to-end scaling over ranks on one box: N processes, device r mod device_count
out=gpurun_out/${1:-r03s}
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_words.py tests/test_gpu_replay.py -x -q > $out/pytest_words.log 2>&1
tail -3 $out/pytest_words.log
for n in 2 4 8; do
  timeout 1500 python bench.py --gpus $n --no-cpu > $out/bench_n$n.json 2> $out/bench_n$n.err
  echo "n=$n rc=$?"
  python - <<PY
import json
try:
    d = json.load(open('$out/bench_n$n.json'))
    e = (d.get('e2e') or {}).get('lca', {})
    print($n, 'value', d.get('value'), 'e2e', {k: e.get(k) for k in ('value', 'value_per_rank', 'seconds', 'phases_s', 'streaming_s', 'tokenizer_threads', 'frac_of_config', 'error')})
except Exception as ex:
    print('n=$n', ex)
PY
  tail -3 $out/bench_n$n.err
done
