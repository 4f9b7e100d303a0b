#!/bin/bash
# selected GPU tests + the lca end-to-end leg
out=gpurun_out/${1:-r03q}
mkdir -p $out
timeout 1200 python -m pytest ${2:-tests/test_gpu_dtok.py tests/test_gpu_words.py} -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -25 $out/pytest.log
timeout 900 python bench.py --no-cpu > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"
python - <<PY
import json
d = json.load(open('$out/bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'e2e_value', 'e2e_ordinal_value') if k in d})
for k, v in (d.get('e2e') or {}).items():
    print(k, {x: v.get(x) for x in ('value', 'seconds', 'phases_s', 'streaming_s', 'value_streaming', 'error')})
PY
tail -5 $out/bench.err
