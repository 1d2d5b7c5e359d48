mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r29.log 2>&1; echo "pytest(G=8 pref) rc=$?"; tail -4 gpurun_out/pytest_gpu_r29.log
PHE_HIP_GROUP=16 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_gpu_r29_g16.log 2>&1; echo "pytest(G=16) rc=$?"; tail -4 gpurun_out/pytest_gpu_r29_g16.log
for cfg in "8 2" "8 1" "8 3" "16 2" "16 4" "16 1"; do
  set -- $cfg
  PHE_HIP_GROUP=$1 timeout 300 python bench.py --batch 131072 --steps 1 --warmup 1 --no-cpu-baseline --blocks-per-cu $2 > gpurun_out/bench_g$1_b$2.json 2> gpurun_out/bench_g$1_b$2.err
  echo "G=$1 blocks/CU=$2 rc=$?"
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_g$1_b$2.json"))
print("   enc/s %.0f  dec/s %.0f  geom %s  exact %s" % (d["value"], d["decrypt"]["value"], d["config"]["geometry"], d["bit_exact"]))
PY
done
