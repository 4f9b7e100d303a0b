#!/bin/bash
# the GPU test-suite, the tokenizer's host scaling and the default bench line
out=gpurun_out/${1:-r03}
mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.log
tail -4 $out/pytest_gpu.log
WOLTKA_TOK_TIMING=1 timeout 900 python tools/tok_scaling.py --threads 8,16,32,64,128 > $out/tok_scaling.txt 2>&1
tail -12 $out/tok_scaling.txt
timeout 1800 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"
python - <<PY
import json
d = json.load(open('$out/bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'e2e_value', 'e2e_ordinal_value') if k in d})
print(d.get('roofline'))
for k, v in (d.get('e2e') or {}).items():
    print(k, {x: v.get(x) for x in ('value', 'seconds', 'phases_s', 'streaming_s', 'value_streaming', 'error', 'text_generated_s')})
PY
tail -5 $out/bench.err
