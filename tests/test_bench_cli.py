"""bench.py's launcher logic that needs no GPU: `--gpus N` without WORLD_SIZE must resolve to an answer (its own ranks, or a
clear error), never to "launch with torch.distributed.run" plumbing advice the driver cannot act on."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DESIRE_BENCH_ONE_GPU")}


def test_gpus_flag_without_enough_devices_is_a_clear_error():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        return
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "64"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "--gpus 64" in (p.stdout + p.stderr) and "visible" in (p.stdout + p.stderr)


def test_mismatched_world_size_names_both_ways_to_launch():
    env = _env()
    env["WORLD_SIZE"] = "3"
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stdout + p.stderr)


def test_committed_traffic_selects_the_launch_class_and_is_never_below_the_algorithmic_bytes():
    """the driver's line carried 43 MB of `traffic` for a launch whose algorithmic bytes are 219 MB (a mixed-size
    average out of a per-symbol summary).  The selection is now per launch class (workgroups = ceil(R / 32)) and a figure below
    the algorithmic bytes is refused."""
    sys.path.insert(0, ROOT)
    import bench
    from desire_amd.spec import Dims
    for windows in (128, 512):
        d = Dims(n_scenes=windows, mno=32, K=20, T_obs=8, T_pred=40, H=128, L=128, n_grids=1, grid_size=4)
        algorithmic = d.R * (2 * d.T_pred * 2 * 4 + 4) + d.A * d.H * 4
        t = bench.committed_traffic(d)
        assert t is not None and bench.committed_traffic.source, windows
        assert algorithmic <= t < 4 * algorithmic, (windows, t, algorithmic, bench.committed_traffic.source)
    # a launch size nobody profiled has no figure (rather than somebody else's)
    d = Dims(n_scenes=16, mno=32, K=20, T_obs=8, T_pred=40, H=128, L=128, n_grids=1, grid_size=4)
    assert bench.committed_traffic(d) is None and bench.committed_traffic.source is None


def test_profile_summaries_of_this_round_are_per_launch_class():
    """profiles/r04_*_pmc_per_kernel.json entries carry their grid; SQ_WAVES (where collected) equals workgroups x waves per workgroup."""
    import glob
    import json
    paths = glob.glob(os.path.join(ROOT, "profiles", "r04_*pmc_per_kernel.json"))
    for p in paths:
        for name, c in json.load(open(p)).items():
            assert "workgroups" in c and "workgroup_size" in c and "[wgs=" in name, (p, name)
            if "SQ_WAVES" in c:
                assert int(round(c["SQ_WAVES"])) == c["waves_expected"], (p, name, c["SQ_WAVES"], c["waves_expected"])


def test_the_drivers_line_stays_small():
    """the 23 KB one-line record of round 5 (committed: profiles/r05_bench_default_full.json) did not parse on the driver's side.
    benchlib/emit.py reduces it to the contract keys + a few scalars per leg, under 4 KB, and prints it LAST; the whole record goes to a `#full` line
    and to bench_full.json."""
    import contextlib
    import io
    import json
    import tempfile
    from benchlib.emit import LINE_BUDGET, compact, emit
    with open(os.path.join(ROOT, "profiles", "r05_bench_default_full.json")) as f:
        rec = json.load(f)
    assert len(json.dumps(rec)) > 20000
    line = compact(rec)
    assert len(json.dumps(line)) <= LINE_BUDGET
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "step_ms_median", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "accuracy"):
        assert k in line, k
    assert line["value"] == rec["value"] and line["ms_per_step"] == rec["ms_per_step"] and line["config"] == rec["config"]
    r = line["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - rec["roofline"]["frac"]) < 1e-4 and r["traffic"] > 0 and r["peak"] == 157.3
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 16 and c["value"] > 0 and "sample" in c
    assert line["alt_ms"]["few_windows"]["windows_1"] > 0 and line["sdd"]["compact_rows_and_ioc"]["ms_per_step"] > 0
    # a record bloated with error strings still fits
    fat = dict(rec, alt={("leg%d" % i): {"error": "x" * 500} for i in range(60)})
    assert len(json.dumps(compact(fat))) <= LINE_BUDGET
    with tempfile.TemporaryDirectory() as tmp:
        os.environ["DESIRE_BENCH_FULL"] = os.path.join(tmp, "full.json")
        try:
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                emit(rec)
        finally:
            del os.environ["DESIRE_BENCH_FULL"]
        lines = buf.getvalue().splitlines()
        assert len(lines) == 2 and lines[0].startswith("#full {") and lines[1].startswith("{") and len(lines[1]) <= LINE_BUDGET + 100
        assert json.loads(lines[0][6:])["alt"] == rec["alt"]
        with open(os.path.join(tmp, "full.json")) as f:
            assert json.load(f)["sdd"] == rec["sdd"]
