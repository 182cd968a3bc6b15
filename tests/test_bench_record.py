"""The ONE JSON line bench.py prints must stay small enough for the driver's capture (BENCH_r05.json: a 20 kB line came back
as `parsed: null`).  The formatter is exercised on canned full records -- every committed `profiles/*bench_line*.json` of the
last two rounds plus a synthetic worst case -- and the contract keys of SURVEY 8(d) must survive the compaction."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")
FULL_RECORDS = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[456]*bench_line*.json")))


def _is_full(rec):
    return "roofline_other_timed_kernels" in rec or "roofline_serial_replay" in rec


@pytest.mark.parametrize("path", FULL_RECORDS, ids=[os.path.basename(p) for p in FULL_RECORDS])
def test_stdout_record_is_compact_and_complete(path):
    full = json.load(open(path))
    if not _is_full(full):                      # a compact record of this round: must already obey the limit
        assert len(json.dumps(full, separators=(",", ":"))) < 6000
        return
    line = bench.compact_record(full)
    assert "\n" not in line and len(line) < 6000, len(line)
    rec = json.loads(line)
    for k in CONTRACT:
        assert k in rec, k
    assert rec["value"] == pytest.approx(full["value"], rel=1e-4) and rec["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-4)
    roof = rec["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "selection"):
        assert k in roof, k
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-3)
    assert isinstance(roof["flop_convention"], str) and roof["flop_convention"] in bench.FLOP_CONVENTIONS      # a key, not a paragraph
    assert "groups" not in rec.get("step_roofline", {}) and "frac" in rec["step_roofline"]
    assert "note" not in rec.get("gradient_exchange", {})
    cb = rec["cpu_baseline"]
    if cb is not None:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in cb, k
    assert "roofline_other_timed_kernels" not in rec and "roofline_serial_replay" not in rec


def test_worst_case_record_sheds_optional_keys_instead_of_growing():
    full = json.load(open(os.path.join(ROOT, "profiles", "r5z_bench_line.json")))
    # many tied kernels with long template names + a bloated exchange object: the line must still fit
    tie = dict(full["roofline"]["selection"]["tied_within_5_percent"][0])
    full["roofline"]["selection"]["tied_within_5_percent"] = [dict(tie, kernel=f"some_very_long_kernel_symbol_name<{i}, 2, 3, 4, 5, 6>")
                                                               for i in range(8)]
    full["gradient_exchange"]["padding"] = "x" * 5000
    line = bench.compact_record(full)
    assert len(line) <= bench.RECORD_LIMIT
    rec = json.loads(line)
    for k in CONTRACT:
        assert k in rec, k


def test_emit_writes_detail_file_and_one_line(tmp_path, monkeypatch):
    full = json.load(open(os.path.join(ROOT, "profiles", "r5z_bench_line.json")))
    monkeypatch.setenv("SSBEV_BENCH_DETAIL", str(tmp_path / "detail.json"))
    r, w = os.pipe()
    bench.emit_record(full, w)
    os.close(w)
    data = os.read(r, 1 << 20).decode()
    os.close(r)
    assert data.count("\n") == 1 and len(data) < 6000
    detail = json.load(open(tmp_path / "detail.json"))
    assert "roofline_serial_replay" in detail and "groups" in detail["step_roofline"]
