#!/bin/bash
# Round 3, first GPU pass (through gpurun): smoke, the GPU tests, the batch-size sweeps (ladder + every rung pinned), one bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
T=${TAG:-r03a}
O=gpurun_out/$T; mkdir -p $O
timeout 300 python __graft_entry__.py smoke; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.txt
for G in 0 4 8 16; do
  timeout 400 python tools/bench_sweep.py --group $G --table > $O/batch_sweep_g$G.json 2> $O/batch_sweep_g$G.txt; echo "sweep g$G rc=$?"
done
cat $O/batch_sweep_g0.txt
timeout 300 python tools/bench_sweep.py --group 2 --ops decrypt --table > $O/batch_sweep_g2_decrypt.json 2> $O/batch_sweep_g2_decrypt.txt; echo "sweep g2 rc=$?"
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_1M.json 2> $O/bench_1M.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_1M.json")); r=d["roofline"]
print(d["value"], d["decrypt"]["value"], r["frac"], d["bit_exact"], d["cpu_baseline"]["value"])
for k,v in d["ops"].items(): print(" ", k, round(v["value"]), v.get("additions_per_s"), v["bit_exact_strided_sample_vs_gmp_oracle"])
print(d["config4"])
PY
