mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r6.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu_r6.log
timeout 120 python examples/federated_learning_batched.py 2048 > gpurun_out/federated_2048_lat.log 2>&1; echo "fed rc=$?"; tail -7 gpurun_out/federated_2048_lat.log
timeout 300 python bench.py --batch 262144 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_256k.json 2> gpurun_out/bench_256k.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/bench_256k.json')); print(d['value'], d['decrypt']['value'], d['bit_exact'])"
