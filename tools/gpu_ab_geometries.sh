#!/bin/bash
# split engine: parity suite, bench at the default geometry and with 8-lane groups
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/split_tests.log
cat gpurun_out/split_tests.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_split.json 2> gpurun_out/bench_split.err
python - <<'PY'
import json
for f in ("bench_split",):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["decrypt"]["value"], d["config"]["geometry"])
PY
PHE_HIP_GROUP=8 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_g8.json 2> gpurun_out/bench_g8.err
PHE_HIP_GROUP=4 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_g4.json 2> gpurun_out/bench_g4.err
python - <<'PY'
import json
for f in ("bench_g8","bench_g4"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["decrypt"]["value"], d["config"]["geometry"])
    except Exception as e: print(f, "failed", e)
PY
