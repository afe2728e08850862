"""bench.py's multi-rank launch contract on CPU: `python bench.py --gpus N` must start its own ranks (one process per GPU) from a
parent that never touches the GPU, relay ONE JSON line, and fail cleanly when the node has fewer GPUs than ranks. The ranks run
`--dry-run` here (gloo, host arithmetic through the same M-shard planner / pipelined all-gather driver): no GPU in this container."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(*argv, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=e, timeout=600)


def _strict(text):
    """json.loads that refuses the bare NaN / Infinity tokens Python's own writer emits by default: the line must be STRICT JSON."""
    def bad(tok):
        raise ValueError(f"non-standard JSON constant {tok}")
    return json.loads(text, parse_constant=bad)


def test_self_launch_two_ranks_dry_run(tmp_path):
    detail = tmp_path / "sub" / "detail.json"
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--detail", str(detail))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # exactly one JSON line on stdout
    import bench
    assert len(lines[0]) <= bench.MAX_LINE_BYTES == 8192
    line = _strict(lines[0])
    # the full record: in the sidecar (directory created on demand) and once more on stderr; the line says where
    full = _strict(detail.read_text())
    assert full["ranks_detail"] == line["ranks_detail"] and full["value"] == line["value"] and line["detail"] == str(detail)
    copies = [ln for ln in r.stderr.splitlines() if "[bench detail] {" in ln]  # (torchrun may prefix the ranks' stderr lines)
    assert len(copies) == 1 and _strict(copies[0][copies[0].index("{"):]) == full
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["config"]["ranks"] == 2 and "m-shard x2" in line["config"]["parallelism"]
    assert line["config"]["all_gather_bytes_per_step"] == 64 * 96 * 4
    assert line["scaling"] == "strong" and line["higher_is_better"] is True
    assert "dry-run" in line["data"]
    # what a multi-rank line says about its CU split and about the rank count the collective library itself reports (none in a dry run)
    assert line["config"]["comm_compute_units"] in (0, 4, 8) and line["config"]["rccl_reported_ranks"] == 0
    # the self-diagnosing part of a multi-rank line: every rank's own ms per step and clock, every panel's exchange-wait time
    _check_ranks_detail(line, world=2, npanels=3)


def _check_ranks_detail(line, world, npanels):
    d, c = line["ranks_detail"], line["config"]
    assert len(d["ms_per_step"]) == world and all(x > 0 for x in d["ms_per_step"])
    assert len(d["clock_ghz"]) == world
    assert len(d["panel_wait_ms_max_over_ranks"]) == npanels and len(d["panel_wait_rank_of_max"]) == npanels and len(d["wait_ms_per_step"]) == world
    assert all(x >= 0 for x in d["panel_wait_ms_max_over_ranks"]) and all(0 <= r < world for r in d["panel_wait_rank_of_max"])
    assert c["rank_ms_per_step_min"] == min(d["ms_per_step"]) and c["rank_ms_per_step_max"] == max(d["ms_per_step"])
    assert c["rank_ms_per_step_max"] <= line["ms_per_step"] * 1.001 + 1e-3, "a rank's own time cannot exceed the max-over-ranks time of the line"
    assert 0 <= c["rank_slowest"] < world
    for k in ("rank_clock_ghz_min", "rank_clock_ghz_max", "exchange_wait_ms_per_step_max", "exchange_wait_ms_last_panel_max", "exchange_wait_ms_before_last_panel_max"):
        assert k in c
    assert abs(c["exchange_wait_ms_per_step_max"] - max(d["wait_ms_per_step"])) < 1e-3


def test_ranks_detail_tells_a_slow_link_from_a_slow_rank():
    """bench.ranks_detail on made-up inputs: per-panel means over the stamped steps, max over ranks with the rank that holds it, the last panel's
    wait apart from the others'."""
    import bench
    per_rank = [{"ms_per_step": 13.1, "clock_ghz": 1.52, "waits": [(0, 0.0), (1, 0.0), (2, 0.4), (0, 0.0), (1, 0.0), (2, 0.6)]},
                {"ms_per_step": 14.9, "clock_ghz": 1.31, "waits": [(0, 0.2), (1, 0.0), (2, 0.1), (0, 0.4), (1, 0.0), (2, 0.1)]}]
    d, c = bench.ranks_detail(per_rank, 3)
    assert d["panel_wait_ms_max_over_ranks"] == [0.3, 0.0, 0.5] and d["panel_wait_rank_of_max"][0] == 1 and d["panel_wait_rank_of_max"][2] == 0
    assert d["wait_ms_per_step"] == [0.5, 0.4]
    assert c["rank_slowest"] == 1 and c["rank_clock_ghz_min"] == 1.31 and c["rank_ms_per_step_max"] == 14.9
    assert c["exchange_wait_ms_last_panel_max"] == 0.5 and c["exchange_wait_ms_before_last_panel_max"] == 0.3 and c["exchange_wait_ms_per_step_max"] == 0.5
    d, c = bench.ranks_detail([{"ms_per_step": 1.0, "clock_ghz": None, "waits": []}], 1)
    assert c["rank_clock_ghz_min"] is None and d["panel_wait_ms_max_over_ranks"] == [0.0]


def test_the_line_stays_small_whatever_rides_in_the_record(tmp_path):
    """bench.emit on a made-up record the size of round 5's (27 secondary workloads' objects, 8 ranks x 64 panels of wait times, a non-finite
    value): the stdout line keeps the contract's keys + roofline + cpu_baseline + targets inside MAX_LINE_BYTES as strict JSON with `targets`
    last, the rest goes to the sidecar untouched."""
    import bench
    rf = {"bound": "mfma", "kernel": "k", "achieved": 1.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.5, "traffic": None, "note": "x" * 300}
    cpu = {"value": 1.0, "unit": "TFLOP/s", "cores": 16, "kind": "port", "sample": "s" * 250}
    others = [{"workload": f"w{i}", "metric": "gemm_tflops", "value": float(i), "roofline": rf, "cpu_baseline": cpu} for i in range(27)]
    targets = {f"c{i}_some_workload_key_tflops": 1234.5 for i in range(70)}
    big_ranks = {"ms_per_step": [1.0] * 8, "panel_wait_ms_max_over_ranks": [0.1234] * 512}
    full = {"metric": "gemm_tflops", "value": 1.0, "unit": "TFLOP/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": float("nan"),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": dict({"workload": "gemm_f16_32768"}, **targets), "roofline": rf, "cpu_baseline": cpu, "checks": {"u": float("inf")},
            "ranks_detail": big_ranks, "others": others, "targets": targets}
    out = []
    text = bench.emit(full, str(tmp_path / "d.json"), out.append)
    assert out == [text] and len(text) <= bench.MAX_LINE_BYTES
    line = _strict(text)
    assert list(line)[-1] == "targets" and line["targets"] == targets and "others" not in line and line["ranks_detail"] is None
    assert line["ms_per_step"] is None and line["checks"]["u"] is None  # non-finite -> null, never a bare NaN
    assert line["roofline"] == rf and line["cpu_baseline"] == cpu and line["config"]["workload"] == "gemm_f16_32768"
    side = _strict((tmp_path / "d.json").read_text())
    assert len(side["others"]) == 27 and side["ranks_detail"] == big_ranks and side["config"] == full["config"]
    # a short ranks_detail (2 ranks, 3 panels) stays in the line; no sidecar requested -> "detail": null
    small = dict(full, others=[], ranks_detail={"ms_per_step": [1.0, 2.0]}, ms_per_step=1.0, checks={})
    line = _strict(bench.emit(small, "-", out.append))
    assert line["ranks_detail"] == {"ms_per_step": [1.0, 2.0]} and line["detail"] is None


def test_rccl_engine_variants_leave_4_and_8_compute_units():
    import bench
    assert bench.ENGINES == ["rccl", "rccl_cus4", "staged"] and bench.ENGINE_COMM_CUS == {"rccl": 8, "rccl_cus4": 4}


def test_self_launch_fails_cleanly_without_enough_gpus():
    ndev = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True).stdout.strip() or 0)
    if ndev >= 2:  # (counted in a child: the pytest process must not load torch's bundled HIP runtime next to the library's)
        return  # a multi-GPU box: the real launch is what runs there
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary")
    assert r.returncode == 2, r.stdout + r.stderr
    assert "GPU(s) visible" in r.stderr and "Traceback" not in r.stderr
    assert r.stdout.strip() == ""


def test_failing_rank_fails_the_launch():
    # a rank that dies (here: an impossible world/--gpus combination inside the children) must surface as a non-zero exit, no line
    r = _run("--gpus", "2", "--dry-run", "--steps", "0", "--warmup", "0")  # steps 0 -> division by zero in every rank
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_hanging_engine_trial_times_out_and_the_next_engine_runs():
    """--gather auto with N > 1 ranks: every exchange engine's trial runs as a fresh child group with a time limit, started by ranks that
    have not touched the GPU. Here (dry run) rank 1 of the FIRST engine's trial sleeps forever -- what a collective that never completes
    looks like -- and the second engine's trial (the RCCL engine again, with 4 CUs left to its kernels) fails on rank 0: the launch must still
    print ONE JSON line, from the third engine, well inside the limits, with both failures recorded."""
    import time
    t0 = time.time()
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--gather", "auto",
             env={"WG_BENCH_DRY_HANG": "rccl:1", "WG_BENCH_DRY_FAIL": "rccl_cus4:0", "WG_BENCH_TRIAL_TIMEOUT": "25", "WG_BENCH_LAUNCH_TIMEOUT": "300"})
    assert r.returncode == 0, r.stdout + r.stderr
    assert time.time() - t0 < 240
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    cfg = line["config"]
    assert cfg["gather_engine"] == "staged" and cfg["chosen"] == "staged"
    assert cfg["engine_trials"]["rccl"] == {"ms_per_step": None, "error": "timeout"}
    assert cfg["engine_trials"]["rccl_cus4"]["ms_per_step"] is None and cfg["engine_trials"]["rccl_cus4"]["error"]
    assert cfg["engine_trials"]["staged"]["ms_per_step"] > 0
    assert set(cfg["engine_trials"]) == {"rccl", "rccl_cus4", "staged"}  # the RCCL engine is tried with 8 and with 4 CUs left to its kernels
    assert cfg["comm_compute_units"] == 0 and cfg["rccl_reported_ranks"] == 0  # staged engine: every CU computes; the dry run has no RCCL
    assert cfg["ranks"] == 2 and cfg["all_gather_bytes_per_step"] == 64 * 96 * 4  # the contract's fields are still there


def test_every_engine_hanging_fails_the_launch_in_bounded_time():
    import time
    t0 = time.time()
    r = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", "--gather", "auto",
             env={"WG_BENCH_DRY_HANG": "rccl:0,rccl_cus4:1,staged:0", "WG_BENCH_TRIAL_TIMEOUT": "10", "WG_BENCH_LAUNCH_TIMEOUT": "200"})
    assert r.returncode != 0
    assert time.time() - t0 < 150
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "no exchange engine works" in r.stderr


def test_panel_planner_whole_rounds():
    import bench
    # 4 ranks of the 32768^3 problem on 256 CUs: 32 tile rows -> 8 tile columns (2048) fill exactly one round; 16 panels
    assert bench.plan_panel_cols(8192, 32768, 256) == 2048
    # 8 ranks: 16 tile rows -> 16 tile columns per round -> 4096-column panels, 8 of them
    assert bench.plan_panel_cols(4096, 32768, 256) == 4096
    # CU-masked stream (224 CUs): 32 tile rows x 7 tile columns = one round
    assert bench.plan_panel_cols(8192, 32768, 224) % (7 * 256) == 0
    # tiny problems: at least two panels when N allows, never wider than N
    assert bench.plan_panel_cols(256, 512, 256) == 256
    assert bench.plan_panel_cols(256, 256, 256) == 256


def test_panel_planner_properties():
    """plan_panel_cols over a sweep of shard heights / widths / CU counts: panels are whole tiles, never wider than N, at least two when N
    allows, and no candidate with fewer idle CUs in its last round was passed over."""
    import bench
    for tile in (128, 256):
        for cus in (224, 248, 256):
            for mg in (256, 1024, 4096, 8192, 16384, 32768, 5000):
                for n in (256, 512, 2048, 8192, 32768, 12800):
                    pc = bench.plan_panel_cols(mg, n, cus, tile=tile)
                    assert 0 < pc <= n and (pc % tile == 0 or pc == n), (mg, n, cus, tile, pc)
                    if n >= 2 * tile:
                        assert -(-n // pc) >= 2
                    tiles_m = -(-mg // tile)

                    def waste(c):
                        t = tiles_m * c
                        return (-(-t // cus)) * cus / t
                    best = min(round(waste(c), 3) for c in range(1, n // tile + 1) if -(-(n // tile) // c) >= 2 or n // tile < 2)
                    assert round(waste(pc // tile), 3) <= best + 1e-9, (mg, n, cus, tile, pc)
