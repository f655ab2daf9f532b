#!/bin/bash
# round 6, session e: the default bench line with host_pointer / enc_path / step_frac / anscdf4s_chunk4096; multi-device tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_host_layer.py -q -m gpu -k "multi_device or environment" 2>&1 | tail -5 > gpurun_out/r06e_tests.txt
timeout 900 python bench.py > gpurun_out/r06e_bench.json 2> gpurun_out/r06e_bench.err
tail -5 gpurun_out/r06e_bench.err; cat gpurun_out/r06e_tests.txt
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r06e_bench.json") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["flags"])
print(json.dumps(j["roofline"], indent=1)[:1500])
print(json.dumps(j.get("host_pointer"), indent=1))
print(json.dumps(j["configs"].get("anscdf4s_chunk4096"), indent=1))
PY
