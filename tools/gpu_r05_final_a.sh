#!/bin/bash
# TAG=<name> bash tools/gpu_r05_final_a.sh (through gpurun): first half of the round-5 record on the FINAL tree — the GPU suite, the driver's
# bench line, the 1024-bit (WITH ops) and 3072-bit lines, the measured ladder (calibration + check of its picks against every rung
# pinned), the PMC passes behind roofline.traffic for the headline kernels AND the ops kernels.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
R=$PWD; O=gpurun_out/${TAG:-r05k}; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python __graft_entry__.py smoke; echo "smoke rc=$?"
timeout 400 python tools/calibrate_ladder.py --max 17 > $O/ladder_gfx950.txt 2> $O/ladder.err; echo "calibrate rc=$? lines=$(wc -l < $O/ladder_gfx950.txt)"
cp $O/ladder_gfx950.txt python-paillier_amd/phe/ladder_gfx950.txt
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed" $O/pytest_gpu.txt | tail -2
timeout 400 python tools/calibrate_ladder.py --min 8 --max 15 --check $O/ladder_gfx950.txt > $O/ladder_check.json 2>/dev/null; python -c "import json;print('ladder check: worst picked/best pinned', json.load(open('$O/ladder_check.json'))['worst_picked_over_best_pinned'])"
timeout 400 python bench.py > $O/bench_1M.json 2> $O/bench_1M.err; echo "bench rc=$?"
timeout 300 python bench.py --key-bits 1024 --no-cpu-baseline --no-config4 > $O/bench_1024.json 2>/dev/null; echo "1024 rc=$?"
timeout 300 python bench.py --key-bits 3072 --batch 262144 --steps 1 --warmup 1 --no-cpu-baseline --no-config4 > $O/bench_3072.json 2>/dev/null; echo "3072 rc=$?"
TAG=${TAG:-r05k} bash tools/gpu_pmc_traffic.sh > $O/pmc_traffic.log 2>&1; tail -3 $O/pmc_traffic.log
python -c "
import json
d=json.load(open('$O/bench_1M.json')); print('encrypts/s', d['value'], 'decrypts/s', d['decrypt']['value'], 'frac', d['roofline']['frac'], 'stale', d['roofline'].get('stale'))
print({k: round(v['value']) for k, v in d['ops'].items()})"
