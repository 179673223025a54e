"""bench.py quotes counter (`roofline.traffic`, `pmc_classes`) and sampler-parity records only when they were taken on the build it
runs: every record under profiles/ carries `kernel_src_sha` = od_build_source_sha() of the library that produced it."""
import json
import os

import pytest


@pytest.fixture
def bench_mod(monkeypatch, tmp_path):
    import importlib
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(repo)
    bench = importlib.import_module("bench")
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    monkeypatch.setattr(bench, "_lib_sha", lambda: "aaaa000011112222")
    return bench, tmp_path / "profiles"


def test_traffic_record_needs_matching_build(bench_mod):
    bench, prof = bench_mod
    rec = {"B": 32, "L": 8192, "od_flash_attn_bwd": {"read_bytes": 10.0, "write_bytes": 5.0}, "classes": {"gemm_nt": {"mfma_busy": 0.4}}}
    (prof / "r02_traffic.json").write_text(json.dumps({**rec, "kernel_src_sha": "ffff"}))          # another build: never quoted
    traffic, classes, src = bench.load_pmc(32, 8192)
    assert traffic is None and classes is None and src["source"] is None and src["kernel_src_sha"] == "aaaa000011112222"
    (prof / "r03_traffic.json").write_text(json.dumps({**rec, "kernel_src_sha": "aaaa000011112222"}))
    traffic, classes, src = bench.load_pmc(32, 8192)
    assert traffic == 15 and classes == rec["classes"] and src["source"] == os.path.join("profiles", "r03_traffic.json")
    assert bench.load_pmc(8, 32768)[0] is None                                                   # another workload: not quoted either
    (prof / "r04_traffic.json").write_text("{ not json")                                          # a damaged newer record is skipped
    assert bench.load_pmc(32, 8192)[0] == 15


def test_sampler_parity_record_needs_matching_build(bench_mod):
    bench, prof = bench_mod
    modes = {"fp32": {"meets_1e-4": True, "rel_l2_vs_reference_fp32": 3e-7}}
    (prof / "r03_sampler_parity.json").write_text(json.dumps({"kernel_src_sha": "0123", "modes": modes}))
    assert bench.load_sampler_parity() == (None, None)
    (prof / "r03_sampler_parity.json").write_text(json.dumps({"kernel_src_sha": "aaaa000011112222", "modes": modes}))
    rec, src = bench.load_sampler_parity()
    assert rec["modes"] == modes and src.endswith("r03_sampler_parity.json")


def test_library_reports_its_source_hash():
    """od_build_source_sha(): 16 hex digits, and equal to what csrc/source_sha.sh computes from the tree the library was built from."""
    import subprocess
    from osu_dreamer_amd import _lib
    if not os.path.exists(_lib.DEFAULT_SO):
        pytest.skip("the HIP library has not been built")
    import ctypes
    cdll = ctypes.CDLL(_lib.DEFAULT_SO)
    cdll.od_build_source_sha.restype = ctypes.c_char_p
    sha = cdll.od_build_source_sha().decode()
    assert len(sha) == 16 and all(c in "0123456789abcdef" for c in sha)
    tree = subprocess.check_output(["bash", os.path.join(os.path.dirname(_lib.DEFAULT_SO), "csrc", "source_sha.sh")], text=True).strip()
    assert sha == tree, "the in-tree library is stale: run csrc/build.sh"


def test_power_sampler_energy(bench_mod, tmp_path):
    """joules_per_step: the hwmon energy counter's delta where the board has one, otherwise the 100 Hz power trace integrated over the timed
    region (trapezoid, the first / last reading held to the region's ends); None — never a made-up number — without readings."""
    bench, _ = bench_mod
    ps = bench.PowerSampler.__new__(bench.PowerSampler)
    ps.period, ps.dir = 0.01, None
    ps.samples, ps.times = [], []
    ps.t_enter, ps.t_exit, ps.e_enter, ps.e_exit = 10.0, 12.0, None, None
    assert ps.joules()[0] is None
    # a trace: 1000 W for the first second, ramping to 1400 W over the second one
    ps.times = [10.0 + 0.01 * i for i in range(1, 200)]
    ps.samples = [((1000.0 if t <= 11.0 else 1000.0 + 400.0 * (t - 11.0)), 2000.0) for t in ps.times]
    j, src = ps.joules()
    assert j == pytest.approx(1000.0 + 1200.0, rel=2e-3) and "integral" in src
    # an energy counter wins over the trace
    ps.e_enter, ps.e_exit = 5.0e6, 2.405e9
    j, src = ps.joules()
    assert j == pytest.approx(2400.0) and "energy1_input" in src
    # a hwmon directory without the files: every field None
    hw = tmp_path / "hwmon0"
    hw.mkdir()
    ps2 = bench.PowerSampler.__new__(bench.PowerSampler)
    ps2.dir = str(hw)
    assert ps2._num("energy1_input") is None and ps2._read() == (None, None)
