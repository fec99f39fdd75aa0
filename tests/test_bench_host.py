"""Host-side pieces of bench.py that need no GPU: the algorithmic byte count, the rule that a PMC
summary of an older kernel source is never reported as this kernel's traffic, and the CPU leg's thread count."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_follow_the_survey_formula():
    # 36 B read + 64 B matrix + 1 visibility bit + 20 B per emitted command
    assert bench.algorithmic_bytes_per_instance(0.0) == pytest.approx(100.125)
    assert bench.algorithmic_bytes_per_instance(1.0) == pytest.approx(120.125)
    assert bench.algorithmic_bytes_per_instance(0.268553) == pytest.approx(100.125 + 20 * 0.268553)


def test_committed_pmc_summary_belongs_to_the_committed_kernel():
    """profiles/r02_cfg3_pmc_summary.json must have been collected from the kernel sources in this tree; otherwise
    bench.py would print traffic = null ("stale") on the driver's run."""
    row, stale = bench.pmc_traffic(3, 1_000_000)
    if row is None:  # mid-development state: visible, not fatal (bench.py then says "stale", never a wrong number)
        pytest.skip(f"no PMC summary matches the kernel source hash {bench.kernel_source_sha()} (stale: {stale}): "
                    "re-run tools/r02_profiles.sh on the GPU box and copy the summary into profiles/")
    assert 0.98 < row["hbm_bytes_per_launch"] / (1_000_000 * bench.algorithmic_bytes_per_instance(0.268553)) < 1.10


def test_a_summary_of_another_kernel_is_reported_stale_not_as_traffic(monkeypatch):
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: "0" * 16)
    row, stale = bench.pmc_traffic(3, 1_000_000)
    assert row is None and stale and stale.endswith("pmc_summary.json")


def test_cpu_leg_thread_count_is_bounded(monkeypatch):
    monkeypatch.setenv("MIP_BENCH_MAX_THREADS", "3")
    assert 1 <= bench.host_cores() <= 3


def test_committed_bench_line_carries_the_contract_fields():
    line = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_default.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert key in line, key
    assert line["config"]["workload"].startswith("mixed 64-mesh scene, 1 M instances") and line["vs_baseline"] is None
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert line["cpu_baseline"]["kind"] == "port" and "unpinned" in line["parity"]
