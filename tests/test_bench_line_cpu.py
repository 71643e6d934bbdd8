"""CPU: the record bench.py prints last is what the driver parses — one compact JSON object, well inside an
8,191-byte stream tail, carrying the contract keys + ``roofline`` + ``cpu_baseline``; the tables go to an EARLIER
line.  Checked on a full record of a real run (profiles/r5/r5z_bench_default_final.json) and on a synthetic N > 1 one."""
import io
import json
import os
import contextlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAMPLE = os.path.join(ROOT, "profiles", "r5", "r5z_bench_default_final.json")

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "launch_mode", "ranks")
TAIL = 8191


def _sample():
    lines = [l for l in open(SAMPLE).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def _emit(line, tmp_path):
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.emit(line, str(tmp_path / "d" / "bench_detail.json"))
    return buf.getvalue()


def test_last_stdout_line_is_a_compact_record(tmp_path):
    full = _sample()
    assert len(json.dumps(full)) > TAIL, "the sample must be a record that used to overflow the tail"
    out = _emit(full, tmp_path)
    lines = out.splitlines()
    assert len(lines) == 2
    last = lines[-1]
    assert len(last.encode()) <= 4000 < TAIL
    rec = json.loads(last)
    for k in CONTRACT:
        assert k in rec, k
    assert rec["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4)
    assert rec["roofline"]["bound"] == "hbm" and rec["roofline"]["peak"] == 8000.0
    assert rec["cpu_baseline"]["value"] == pytest.approx(full["cpu_baseline"]["value"], rel=1e-4)
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cores"] >= 1
    assert rec["value"] == pytest.approx(full["value"], rel=1e-4)
    assert rec["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-4)
    assert "model" not in rec["config"] and "workload" in rec["config"]
    # the tail of the stream, cut the way the driver cuts it, still holds the whole record as its last line
    tail = out[-TAIL:]
    assert json.loads(tail.splitlines()[-1]) == rec
    # nothing after it, and no other line of the tail starts like JSON
    assert out.endswith(last + "\n")
    # the full record is the earlier line and the file
    assert json.loads(lines[0])["bench_detail"]["variants"].keys() == full["variants"].keys()
    assert json.load(open(tmp_path / "d" / "bench_detail.json"))["kernels"] == full["kernels"]


def test_compact_record_of_a_multi_rank_line_fits(tmp_path):
    full = _sample()
    for k in ("cpu_baseline", "vs_cpu_baseline", "parity", "variants", "multi_gpu_model", "multi_gpu_model_bf16"):
        full.pop(k, None)
    full.update(n_gpus=8, ranks=8,
                rank_skew=dict(per_rank_ms_per_step=[1.0] * 8, max_minus_min_ms=0.01, note="x" * 200),
                collective=dict(backend="nccl (RCCL)", ranks=8, op="all_gather_into_tensor", shard_bytes=5120000,
                                us_per_call_max_over_ranks=80.0, calls_per_step=1),
                frames_in_parallel=dict(scaling="weak", frames_per_step=8, ms_per_step=4.2, value=7.6e7, unit="BEV queries/s",
                                        note="y" * 100),
                strong_scaling=dict(north_star_target=0.85, statement="z" * 500, t1_ms_one_untiled_frame_per_gpu=4.2,
                                    efficiency_t1_over_N_TN=0.4, amdahl_bound_efficiency=0.5))
    rec = json.loads(_emit(full, tmp_path).splitlines()[-1])
    assert rec["n_gpus"] == 8 and rec["frames_in_parallel"]["value"] == pytest.approx(7.6e7)
    assert rec["strong_scaling"]["efficiency_t1_over_N_TN"] == 0.4 and "statement" not in rec["strong_scaling"]
    assert rec["collective"]["us_per_call_max_over_ranks"] == 80.0


def test_oversized_optional_tables_are_dropped_not_truncated(tmp_path):
    import bench
    full = _sample()
    full["kernels"] = {f"kernel_{i:04d}_{'k' * 40}": dict(avg_us=1.0, launches=1, alg_bytes=1, GBs=1.0) for i in range(400)}
    rec = bench.compact_line(full)
    assert len(json.dumps(rec, separators=(",", ":"))) <= bench.COMPACT_LIMIT
    assert "kernels_avg_us" not in rec and "roofline" in rec and "cpu_baseline" in rec


def test_default_single_gpu_run_creates_no_process_group():
    """The N = 1 default path never calls init_process_group (its c10d warning line broke the round-5 record)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'os.environ.setdefault("TORCH_CPP_LOG_LEVEL", "ERROR")' in src.split("import torch")[0]
    assert "if args.ddp_eager:" in src
