"""bench.py's bookkeeping, checked without a GPU: the algorithmic MAC32 counts are exactly SURVEY.md 8(d)'s table, the
executed multiply-add model follows the window sizes key_setup.h picks, and the peak is the calibrated half-rate one."""
import os
import sys

from conftest import ROOT

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_canonical_mac32_counts_match_survey_table():
    assert bench.mac32_counts(1024) == (10023808, 2569600)
    assert bench.mac32_counts(2048) == (79250560, 20056960)
    assert bench.mac32_counts(3072) == (266342976, 67133952)


def test_executed_multiply_adds_model():
    info = {"lane_limbs_pub": 418, "lane_limbs_priv": 218, "engine_pub": "split", "engine_priv": "split"}
    enc, dec = bench.executed_mads(2048, info)
    H = 72
    # 2048 + 1 squarings (the ladder's and base^2 of the table) at (3 + 10/18) H^2 — round 5: the symmetric half of X0*X0, 10 of the
    # 18 limbs of a lane per row (csrc/split_core.h sq_row; 4 H^2 before) —, ceil(2048/7) + 32 products at 5 H^2 (6-bit windows; one
    # of them was the square: - 4 + ...), entry 4 H^2, exit 10 H^2, nude factor 2 H^2
    sq = 3 + 10 / 18
    assert enc == round((sq * 2049 - 4 + 5 * (293 + 32) + 4 + 10 + 2) * H * H)
    # decrypt: two half-width exponentiations over 1024-bit exponents on H = 36 (2 lanes x 18), inputs of 4 chunks (+ one product)
    h = 36
    assert dec == 2 * round((sq * 1025 - 4 + 5 * (147 + 32) + 4 * 4 + 5 + 10) * h * h)
    assert enc < bench.mac32_counts(2048)[0]                   # the split engine issues fewer multiplies than the canonical count
    full = {"lane_limbs_pub": 436, "lane_limbs_priv": 236, "engine_pub": "full", "engine_priv": "full"}
    enc_full, _ = bench.executed_mads(2048, full)
    assert enc_full == (2048 + 325 + 3) * 2 * 144 * 144


def test_peak_is_the_calibrated_half_rate():
    peak, sustained, src = bench.valu_peak_mac32()
    assert abs(peak - 256 * 4 * 64 * 2.4e9 / 4) < 1
    assert sustained is None or 0.7 * peak < sustained < peak
    assert os.path.exists(os.path.join(ROOT, "profiles", src)) if src else True


def test_exact_counts_back_the_model():
    """profiles/executed_mads_r*.json (tools/count_executed_mads.py: wave::mad64 calls counted by the CPU wave emulator)
    is what roofline.frac is computed from; the closed-form model must stay within 1 % of it"""
    info = {"lane_limbs_pub": 418, "lane_limbs_priv": 218, "engine_pub": "split", "engine_priv": "split"}
    counted, src = bench.counted_mads(2048, info)
    assert counted and os.path.exists(os.path.join(ROOT, "profiles", src))
    enc, dec = bench.executed_mads(2048, info)
    assert abs(counted["encrypt"] / enc - 1) < 0.01 and abs(counted["decrypt"] / dec - 1) < 0.01
    assert counted["encrypt"] < bench.mac32_counts(2048)[0]
    assert counted["raw_add_two_montgomery_products"] == 4 * 144 * 144       # two full-width 144-limb Montgomery products
    # ... which is not what a large batch runs any more (round 4): one plain product + one fold against the key's table, by tiles
    # with one element per lane (mul_tile.h) — about half the multiply-adds (padding of the column blocks included)
    assert counted["raw_add_form"] == "tiles" and counted["raw_add"] == counted["raw_add_tiles"]
    assert 0.95 * 2 * 144 * 144 < counted["raw_add_table_in_lds"] < counted["raw_add_tiles"] < 1.1 * 2 * 144 * 144
    # 1024-bit keys: by tiles too since round 5 (the 8-wave shape: n^2 fills 8 x 9 columns; 11,160 multiply-adds against 20,736)
    small = bench.counted_mads(1024, dict(info, lane_limbs_pub=218))[0]
    assert small["raw_add_form"] == "tiles" and small.get("raw_add_tile_waves") == 8 and small["raw_add"] * 1.8 < small["raw_add_two_montgomery_products"]
    # the count per geometry of the CRT halves (the ladder may end on another rung than the narrowest: 3072 bits run the halves
    # on 4 x 14): same work on 2 x 18 and 4 x 9 (both 36 limbs), more on 56 limbs than on 54
    assert counted["decrypt_by_halves_geometry"]["218"] == counted["decrypt"] == counted["decrypt_by_halves_geometry"]["409"]
    wide, _ = bench.counted_mads(3072, dict(info, lane_limbs_pub=427))
    assert wide["decrypt_by_halves_geometry"]["227"] == wide["decrypt"] < wide["decrypt_by_halves_geometry"]["414"]
    assert wide["decrypt_by_halves_geometry"]["414"] / wide["decrypt"] < (56 / 54) ** 2
    # a geometry the count was not made for falls back to the model
    assert bench.counted_mads(2048, dict(info, lane_limbs_pub=236))[0] is None
    assert bench.counted_mads(2048, dict(info, engine_pub="full"))[0] is None


def test_ops_canonical_counts_match_survey_table():
    ops = bench.mac32_ops(2048)
    assert ops["raw_add"] == 32768 and ops["raw_mul_float56"] == 2763264
    assert bench.mac32_ops(3072)["raw_mul_float56"] == 6209280 and bench.mac32_ops(1024)["raw_add"] == 8192


def _run_bench(cmd, timeout=600):
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


SELFTEST = ["--selftest-emu", "--key-bits", "256", "--batch", "6", "--steps", "2", "--warmup", "1",
            "--config4-key-bits", "256", "--config4-total", "9"]


def _check_two_rank_line(out):
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["bit_exact"] == {"roundtrip_full_batch": True, "strided_sample_vs_gmp_oracle": True, "strided_sample_rows": 6}
    assert set(out["ops"]) >= {"raw_add", "raw_mul_float56", "raw_mul_int64", "raw_mul_float56_neg10pct", "obfuscate"}
    assert all(rec["bit_exact_strided_sample_vs_gmp_oracle"] is True and rec["value"] > 0 for rec in out["ops"].values())
    cfg4 = out["config4"]
    assert cfg4["total"] == 9 and cfg4["rows_per_gpu"] == 5 and cfg4["bit_exact_boundaries_and_sample_vs_gmp_oracle"] is True
    assert cfg4["all_gather"]["bytes_received_per_gpu"] == 9 * 16 * 4
    assert out["cpu_baseline"] is None                           # rank 0 at N = 1 only
    assert "SELFTEST" in out["data"]                             # never mistaken for a measurement


def test_bench_gpus_2_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` started bare must become two ranks (round-1 verdict: --gpus was parsed and ignored).
    CPU stand-in: gloo + the wave emulator (--selftest-emu); the N > 1 plumbing is the same code as on the GPUs."""
    _check_two_rank_line(_run_bench([sys.executable, "bench.py", "--gpus", "2"] + SELFTEST))


def test_bench_under_torchrun_is_a_rank():
    """the driver's launch form for N > 1: torch.distributed.run starts the ranks, bench.py must not re-launch"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", "bench.py", "--gpus", "2"] + SELFTEST
    _check_two_rank_line(_run_bench(cmd))


def test_bench_strong_scaling_shards_one_job_over_the_ranks():
    """--scaling strong: --batch is the WHOLE job (7 rows over two ranks: 4 + 3), configs[3] keeps its total at every N, and the
    line says which kind of scaling it is"""
    args = [a for a in SELFTEST]
    args[args.index("--batch") + 1] = "7"
    out = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--scaling", "strong"] + args)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["rccl_ranks"] == 2
    assert out["config"]["batch_whole_job"] == 7 and out["config"]["batch_per_gpu"] == 4
    assert out["bit_exact"]["roundtrip_full_batch"] is True
    cfg4 = out["config4"]
    assert cfg4["scaling"] == "strong" and cfg4["total"] == 9 and cfg4["rows_per_gpu"] == 5
    assert cfg4["bit_exact_boundaries_and_sample_vs_gmp_oracle"] is True
    assert cfg4["memory_budget_bytes_rank0"]["ciphertext_shard"] == 5 * 16 * 4
    weak = _run_bench([sys.executable, "bench.py", "--gpus", "2"] + SELFTEST)
    assert weak["scaling"] == "weak" and weak["config"]["batch_whole_job"] == 12 and weak["config4"]["scaling"] == "weak"


def test_bench_single_process_line():
    out = _run_bench([sys.executable, "bench.py"] + SELFTEST + ["--config4"])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1
    assert out["config4"]["rows_per_gpu"] == 9 and out["config4"]["bit_exact_boundaries_and_sample_vs_gmp_oracle"] is True
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["matches_gpu"] is True
    roof = out["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "canonical_frac"} <= set(roof)


def _run_bench_raw(cmd, timeout=600):
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    return res.returncode, (json.loads(lines[0]) if len(lines) == 1 else None), res.stderr[-3000:]


def test_a_wrong_row_in_the_cpu_baseline_sample_fails_the_run():
    """`cpu_baseline.matches_gpu` — the largest libgmp sample of the line — used to be reported and
    ignored.  One flipped bit in a ciphertext that ONLY that leg looks at (--oracle-sample 1 checks row 0, the fault sits in
    row 1) must make bench.py exit non-zero, with the line saying which comparison failed."""
    rc, out, err = _run_bench_raw([sys.executable, "bench.py"] + SELFTEST + ["--oracle-sample", "1", "--inject-fault", "cpu_baseline"])
    assert rc != 0, err
    assert out["bit_exact"] == {"roundtrip_full_batch": True, "strided_sample_vs_gmp_oracle": True, "strided_sample_rows": 1}
    assert out["cpu_baseline"]["matches_gpu"] is False
    assert all(rec["bit_exact_strided_sample_vs_gmp_oracle"] is True for rec in out["ops"].values())
    # the same command without the fault exits 0
    rc, out, err = _run_bench_raw([sys.executable, "bench.py"] + SELFTEST + ["--oracle-sample", "1"])
    assert rc == 0 and out["cpu_baseline"]["matches_gpu"] is True, err


def test_a_wrong_row_in_an_ops_result_or_in_the_gathered_vector_fails_the_run():
    rc, out, err = _run_bench_raw([sys.executable, "bench.py"] + SELFTEST + ["--inject-fault", "raw_add"])
    assert rc != 0 and out["ops"]["raw_add"]["bit_exact_strided_sample_vs_gmp_oracle"] is False, err
    assert out["ops"]["raw_add"]["rows_checked"] == 6            # every row of the 6-row selftest batch (default: 4,096 strided rows)
    assert out["ops"]["obfuscate"]["bit_exact_strided_sample_vs_gmp_oracle"] is True
    rc, out, err = _run_bench_raw([sys.executable, "bench.py"] + SELFTEST + ["--inject-fault", "config4"])
    assert rc != 0 and out["config4"]["bit_exact_boundaries_and_sample_vs_gmp_oracle"] is False, err
    assert out["config4"]["rows_checked"] == 9


def test_a_library_allgather_that_hangs_does_not_cost_the_line():
    """The library's own RCCL gather (default for N > 1) is the one leg no box with more than one GPU has run: it runs last,
    under a watchdog.  With the leg made to hang (test hook) both ranks must still end with exit code 0 and rank 0 must print the
    whole line, `config4.all_gather_by_library_rccl.error` saying what happened."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PHE_BENCH_TEST_LIB_GATHER_HANG="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--lib-allgather-timeout", "3"] + SELFTEST, env=env,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    import json
    out = json.loads(lines[0])
    _check_two_rank_line(out)
    assert "no answer within 3 s" in out["config4"]["all_gather_by_library_rccl"]["error"]
    assert "did not finish" in res.stderr


def test_sample_sizes_of_the_default_line():
    """what a default run checks against libgmp: 4,096 strided rows of the headline batch, of every ops result and of the
    configs[3] job (plus its shard boundaries)"""
    a = bench.parse_args([])
    assert (a.oracle_sample, a.ops_sample, a.config4_sample) == (4096, 4096, 4096) and a.inject_fault is None


def test_eight_ranks_weak_and_strong():
    """the shape of the driver's 8-GPU run (python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8), on gloo + the
    wave emulator: eight ranks, ragged configs[3] shards (9 rows over 8 ranks: 2 1 1 1 1 1 1 1), one all-gather, one line"""
    out = _run_bench([sys.executable, "bench.py", "--gpus", "8"] + SELFTEST, timeout=1200)
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["scaling"] == "weak"
    assert out["config"]["batch_whole_job"] == 48 and out["config"]["batch_per_gpu"] == 6
    assert out["bit_exact"]["roundtrip_full_batch"] is True and out["bit_exact"]["strided_sample_vs_gmp_oracle"] is True
    cfg4 = out["config4"]
    assert cfg4["total"] == 9 and cfg4["rows_per_gpu"] == 2 and cfg4["bit_exact_boundaries_and_sample_vs_gmp_oracle"] is True
    assert cfg4["rows_checked"] == 9 and cfg4["all_gather"]["bytes_received_per_gpu"] == 9 * 16 * 4
    assert out["cpu_baseline"] is None
    args = [a for a in SELFTEST]
    args[args.index("--batch") + 1] = "19"                      # 19 rows over 8 ranks: 3 3 3 2 2 2 2 2
    out = _run_bench([sys.executable, "bench.py", "--gpus", "8", "--scaling", "strong"] + args, timeout=1200)
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["scaling"] == "strong"
    assert out["config"]["batch_whole_job"] == 19 and out["config"]["batch_per_gpu"] == 3
    assert out["bit_exact"]["roundtrip_full_batch"] is True
    assert out["config4"]["total"] == 9 and out["config4"]["bit_exact_boundaries_and_sample_vs_gmp_oracle"] is True


def test_memory_budget_of_the_real_eight_gpu_job():
    """BASELINE configs[3] as the driver will run it: 3072-bit key, 8M plaintexts over 8 GPUs (2^20 per GPU), the gathered
    vector held once by torch's all-gather and once more by the library's own RCCL gather — far inside 288 GB of HBM, also
    with configs[1]'s operands still resident and also for the strong form's single-GPU point (the whole 8M job on one GPU)"""
    from phe.sharding import shard_bounds
    t1, t2, total = 96, 192, 1 << 23
    resident_cfg1 = (2 * 64 + 128 + 64) * 4 << 20               # m, r, c, m_back of the 2^20-row headline batch
    for world in (1, 2, 4, 8):
        for rank in range(world):
            lo, hi = shard_bounds(total, world, rank)
            b = bench.config4_memory_budget(hi - lo, total, t1, t2, world > 1, world > 1)
            assert b["ciphertext_shard"] == (hi - lo) * 768 and b["operands_m_r"] == (hi - lo) * 768
            assert b["gathered_vector_torch"] == (total * 768 if world > 1 else 0) == b["gathered_vector_library_rccl"]
            assert sum(b.values()) + resident_cfg1 < 0.1 * 288e9, (world, rank, sum(b.values()))
    b8 = bench.config4_memory_budget(1 << 20, total, t1, t2, True, True)
    assert b8["gathered_vector_torch"] == 6442450944            # the 6.4 GB of SURVEY 8(a) row a4, twice on every GPU
    assert 15.5e9 < sum(b8.values()) + resident_cfg1 < 18e9
    # weak form (the default line): 2^20 rows per GPU at every N, the gathered vector grows with N
    bw = bench.config4_memory_budget(1 << 20, 8 << 20, t1, t2, True, True)
    assert bw == b8
