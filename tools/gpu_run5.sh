mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r5.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu_r5.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --batch 65536 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; echo "torchrun rc=$?"; cut -c1-400 gpurun_out/bench_torchrun1.json; tail -3 gpurun_out/bench_torchrun1.err
timeout 120 python examples/federated_learning_batched.py 2048 > gpurun_out/federated_2048.log 2>&1; echo "fed rc=$?"; cat gpurun_out/federated_2048.log
timeout 200 python __graft_entry__.py smoke; echo "smoke rc=$?"
