#!/bin/bash
# TAG=<name> bash tools/gpu_r06_every_row.sh (through gpurun): EVERY row against libgmp, not samples — the configs[2] results of a
# 2^20-row batch at 2048 bits (add, chains, both scalar-multiply branches, obfuscate, key owner), the encrypts of a 2^20-row batch at
# 1024 bits and of a 2^18-row batch at 3072 bits.  Not bench lines (one step each; the oracle dominates the wall time).
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
O=gpurun_out/${TAG:-r06e}; mkdir -p $O
timeout 1500 python bench.py --steps 1 --warmup 0 --oracle-sample 4096 --ops-sample 1048576 --no-config4 --no-cpu-baseline > $O/every_row_ops_2048_1M.json 2> $O/ops.err; echo "ops rc=$?"
timeout 600 python bench.py --key-bits 1024 --steps 1 --warmup 0 --oracle-sample 1048576 --no-ops --no-config4 --no-cpu-baseline > $O/every_row_vs_libgmp_1024_1M.json 2>/dev/null; echo "1024 rc=$?"
timeout 900 python bench.py --key-bits 3072 --batch 262144 --steps 1 --warmup 0 --oracle-sample 262144 --no-ops --no-config4 --no-cpu-baseline > $O/every_row_vs_libgmp_3072_256k.json 2>/dev/null; echo "3072 rc=$?"
python - <<PY
import json
for f in ("every_row_ops_2048_1M","every_row_vs_libgmp_1024_1M","every_row_vs_libgmp_3072_256k"):
    p=json.load(open("$O/%s.json"%f))
    print(f, p["bit_exact"], {k:(v["bit_exact_strided_sample_vs_gmp_oracle"],v["rows_checked"]) for k,v in (p.get("ops") or {}).items()})
PY
