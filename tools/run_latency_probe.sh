#!/bin/bash
# Build (if needed) and run tools/latency_probe.hip; the binary lives beside the library so that it travels with a gpurun snapshot.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
BIN="$ROOT/python-paillier_amd/lib/phe_latency_probe"
if [ "$1" = "--build" ] || [ ! -x "$BIN" ] || [ "$ROOT/tools/latency_probe.hip" -nt "$BIN" ] || [ "$ROOT/python-paillier_amd/csrc/split_core.h" -nt "$BIN" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -o "$BIN" "$ROOT/tools/latency_probe.hip" || exit 1
fi
[ "$1" = "--build" ] || "$BIN"
