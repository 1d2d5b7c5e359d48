#!/bin/bash
# TAG=<name> bash tools/gpu_r06_first.sh (through gpurun): round 6, first measurement of the tree — the whole GPU suite, the driver's
# bench line, configs[4] in the file's scalar call shape at 1024 and 2048 bits, API-level rates (the decode of decrypt_batch), and
# 3072-bit keys with the 4 x 27 (single-word sweeps) and 8 x 14 (fused sweeps) rungs pinned at 2^20 rows.
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
O=gpurun_out/${TAG:-r06a}; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python __graft_entry__.py smoke; echo "smoke rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 400 python -m pytest tests/test_federated_example.py -m gpu -q -s -k scalar > $O/federated_scalar_api_gpu.txt 2>&1; echo "federated rc=$?"; grep -E "configs\[4\]|Hospital|rounds" $O/federated_scalar_api_gpu.txt | head -20
timeout 600 python bench.py > $O/bench_1M.json 2> $O/bench_1M.err; echo "bench rc=$?"
timeout 300 python tools/bench_api.py 1048576 > $O/bench_api_1M.json 2> $O/bench_api.err; echo "api rc=$?"; cat $O/bench_api_1M.json | head -c 1500
for grp in 0 4 8; do
  timeout 400 python tools/bench_sweep.py --key-bits 3072 --group $grp --min 20 --max 20 --ops encrypt,decrypt,mul --budget-ms 4000 --table > $O/pin3072_g$grp.json 2> $O/pin3072_g$grp.txt
  echo "== 3072 bits, 2^20 rows, --group $grp"; grep -v amdgpu $O/pin3072_g$grp.txt | tail -5
done
