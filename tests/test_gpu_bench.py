"""bench.py's contract, exercised on the GPU box: one JSON line with the fields the driver reads, and the multi-rank code
path (barrier / max-over-ranks timing / sharded work) run as two ranks that share the single GPU over gloo."""
import json
import os

import numpy as np
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _line(out):
    """the driver's line: the LAST stdout line, the only one that starts with `{`, small enough for the driver's parser"""
    all_lines = [l for l in out.splitlines() if l.strip()]
    lines = [l for l in all_lines if l.startswith("{")]
    assert len(lines) == 1 and all_lines[-1] == lines[0], out[-2000:]
    assert len(lines[0]) < 6000, len(lines[0])
    o = json.loads(lines[0])
    for k in REQUIRED:
        assert k in o, k
    return o


def _last_json(out):
    """the complete record (every leg with its notes and per-kernel tables): the `#full ` line printed before the driver's line; its contract keys
    must be the ones the driver's line carries"""
    line = _line(out)
    full = [l for l in out.splitlines() if l.startswith("#full ")]
    assert len(full) == 1, out[-2000:]
    o = json.loads(full[0][len("#full "):])
    for k in REQUIRED:
        assert o[k] == line[k], k
    for k in ("roofline", "cpu_baseline", "accuracy"):
        assert (k in o) == (k in line), k
    if "roofline" in o and o["roofline"].get("frac") is not None:
        assert abs(line["roofline"]["frac"] - o["roofline"]["frac"]) < 1e-4 * abs(o["roofline"]["frac"])
        assert abs(line["roofline"]["achieved"] - o["roofline"]["achieved"]) < 1e-4 * abs(o["roofline"]["achieved"])
    return o


def test_default_contract_fields():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    p = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--windows", "16"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _line(p.stdout)
    for k in REQUIRED + ("roofline", "cpu_baseline", "accuracy", "step_ms_median"):
        assert k in line, k
    lr, lc = line["roofline"], line["cpu_baseline"]
    assert lr["bound"] == "mfma" and 0 < lr["frac"] < 1 and abs(lr["frac"] - lr["achieved"] / lr["peak"]) < 1e-3 and lr["unit"] == "TFLOP/s" and "traffic" in lr
    assert lc["kind"] == "port" and lc["value"] > 0 and lc["cores"] >= 1 and lc["sample"] and lc["unit"]
    assert line["accuracy"]["max_abs_err_Y"] < line["accuracy"]["gate"] == 1e-3
    assert line["sdd"]["ms_per_step"] > 0 and line["alt_ms"]["few_windows"]["windows_1"] > 0 and line["alt_ms"]["training_step"]["fp32"] > 0
    assert 0 < line["alt_figures"]["bf16_config2_mno128_ioc_frac_of_bf16_peak"] < 1
    assert os.path.exists(os.path.join(ROOT, line["full_record"]))
    with open(os.path.join(ROOT, line["full_record"])) as f:
        assert json.load(f)["value"] == line["value"]
    o = _last_json(p.stdout)
    for k in ("roofline", "cpu_baseline", "accuracy"):
        assert k in o, k
    assert o["n_gpus"] == 1 and o["steps"] == 2 and o["warmup"] == 1 and o["vs_baseline"] is None and o["scaling"] == "weak"
    r = o["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert o["accuracy"]["max_abs_err_Y"] < o["accuracy"]["gate"] == 1e-3 and o["accuracy"]["max_abs_err_Y0"] < 1e-3
    c = o["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert abs(o["value"] - o["config"]["rows_per_gpu"] * 2 / (o["ms_per_step"] * 2e-3)) < 1e-6 * o["value"]
    # round 2: the real-SDD leg next to the dense headline, executed-flops fraction, labelled traffic, host core count
    assert o["sdd"]["value"] > 0 and "bookstore" in o["sdd"]["data"] and o["sdd"]["ms_per_step"] > 0
    # round 5: the same windows with padding skipped -- present rows bit-identical with DESIRE_FLAG_COMPACT_ROWS, equal to fp32 summation order with the slot
    # classes on top, every compacted form faster than the padded one (9 of 32 slots present)
    sd = o["sdd"]
    assert sd["compact_rows"]["present_rows_bit_identical_to_uncompacted"] is True
    assert sd["compact_rows_and_ioc"]["max_abs_diff_present_rows_vs_uncompacted"] < 1e-5
    # (16 windows here: the slot classes fold into the handle's own below 8192 rows, so only the row compaction is expected to show)
    assert sd["compact_rows"]["ms_per_step"] < sd["ms_per_step"] and sd["compact_rows_and_ioc"]["ms_per_step"] < 1.25 * sd["compact_rows"]["ms_per_step"]
    assert sd["split_bf16x3_compact_rows_and_ioc"]["value_present_agents_only"] > sd["value_present_agents_only"]
    assert sd["bf16_compact_rows_and_ioc"]["ms_per_step"] > 0 and sd["bf16"]["ms_per_step"] > 0
    assert 0 < r["whole_path_frac_executed"] < r["whole_path_frac"] < 1
    assert "traffic_source" in r and (r["traffic"] is None or r["traffic"] >= r["algorithmic_hbm_bytes_per_launch"])
    assert 0 < r["kernel_ms"] <= o["ms_per_step"]
    assert c["host_cores"] >= c["threads"] >= 1
    # the opt-in forms measured beside the headline: row-compacted pooling, split-bf16 IOC (same results to ~1e-5)
    sp = o["alt"]["split_bf16x3_ioc"]
    assert sp["value"] > 0 and sp["ioc_ms"] > 0 and 0 < sp["max_abs_diff_vs_fp32_kernel"] < 1e-4
    assert o["alt"]["row_compacted_pooling"]["value"] > 0
    assert o["alt"]["reference_defaults"]["batch_size_10"]["value"] > 0 and o["alt"]["reference_defaults"]["windows_128"]["value"] > 0
    fw = o["alt"]["few_windows"]                              # literal configs[1]: one window per call
    assert 0 < fw["windows_1"]["ms_per_call"] <= fw["windows_8"]["ms_per_call"] < 10
    wl = o["alt"]["with_loader"]                              # round 4: the loader in the loop (host loader / device builder feeding whole steps)
    assert wl["host_loader"]["next_batch_into_pinned_f32_windows_per_s"] > 0 and wl["host_loader"]["next_batch_windows_per_s"] > 0
    assert wl["device_builder"]["windows_per_s"] > 0 and wl["resident"]["value"] > 0
    assert 0.5 < wl["fed_by_host_loader"]["fraction_of_resident"] < 1.1 and 0.5 < wl["fed_by_device_builder"]["fraction_of_resident"] < 1.1
    c3 = o["alt"]["config3_shape"]                            # BASELINE configs[3] per-GPU shape: fp32 and the step-wise split forms at H = 256
    assert c3["fp32"]["ioc_ms"] > c3["split_bf16x3"]["ioc_ms"] > 0 and c3["split_bf16x6"]["ioc_ms"] > c3["split_bf16x3"]["ioc_ms"]
    assert c3["split_bf16x3"]["ioc_max_abs_diff_vs_fp32_kernel"] < 1e-4 and c3["split_bf16x6"]["ioc_max_abs_diff_vs_fp32_kernel"] < 2e-6
    tr = o["alt"]["training_step"]                            # configs[4]'s per-GPU work, fp32 and split operands
    assert tr["fp32"]["value"] > 0 and tr["split_bf16x3"]["value"] > tr["fp32"]["value"] and np.isfinite(tr["split_bf16x3"]["loss"])
    assert tr["fp32"]["roofline"]["algorithmic_flops_per_step"] > 0 and 0 < tr["fp32"]["roofline"]["frac"] < 1            # VERDICT r04 missing 3: a roofline on the training leg
    ts = tr["sdd"]                                            # real SDD windows: the compacted training steps are the faster ones, same loss
    assert ts["split_bf16x3_compact_rows_and_ioc"]["ms_per_step"] < ts["split_bf16x3_compact_rows"]["ms_per_step"] < ts["split_bf16x3"]["ms_per_step"]
    assert abs(ts["fp32_compact_rows_and_ioc"]["loss"] - ts["fp32"]["loss"]) < 1e-5 * abs(ts["fp32"]["loss"])
    sk = wl["skip_padding"]                                   # the loaders against the 3x shorter compacted step
    assert sk["resident_ms_per_step"] < wl["resident"]["ms_per_step"] and sk["device_builder_fraction_of_resident"] > 0.5
    tp = tr["split_bf16x3_two_piece_forward"]                 # DESIRE_FLAG_TRAIN_FWD_3P: faster, and the same loss to the step's rounding
    assert tp["value"] > tr["split_bf16x3"]["value"] and abs(tp["loss"] - tr["split_bf16x3"]["loss"]) < 1e-3 * abs(tr["split_bf16x3"]["loss"])
    assert o["accuracy"]["x6_max_abs_err_Y0"] < 2e-6 and o["accuracy"]["x6_max_abs_err_Y"] < 2e-6       # the fp32 kernels' own class
    v64 = o["accuracy"]["vs_float64_oracle"]                  # against exact (float64) arithmetic: not further than the fp32 implementations
    for key in ("Y0", "Y"):
        worst32 = max(v64[key]["fp32_kernels"]["max"], v64[key]["fp32_numpy_oracle"]["max"])
        assert v64[key]["six_products"]["max"] < max(1.5 * worst32, 5e-7), v64[key]
    s6 = o["alt"]["split_bf16x6_ioc"]
    assert s6["value"] > 0 and s6["ioc_ms"] > 0 and s6["max_abs_diff_vs_fp32_kernel"] < 2e-6 and s6["max_abs_diff_vs_fp32_kernel"] < sp["max_abs_diff_vs_fp32_kernel"]
    # round 3: BASELINE configs[2] (128 agents per scene, bf16 operands) is measured by the default command, with its own roofline
    # fraction and its accuracy against the rounding oracle
    c2 = o["alt"]["bf16_config2"]
    for tag, mno in (("mno128", 128), ("mno32", 32)):
        assert c2[tag]["agents_per_scene"] == mno and c2[tag]["samples_per_step"] == 81920
        assert c2[tag]["value"] > 0 and 0 < c2[tag]["ioc_frac_of_bf16_peak"] < 1 and 0 < c2[tag]["whole_path_frac_of_bf16_peak"] < 1
    assert c2["accuracy"]["decoder_max_abs_err_vs_fp32_oracle"] < 1e-3
    assert c2["accuracy"]["ioc_max_abs_err_vs_rounding_oracle"] < 7e-3 * c2["accuracy"]["refinement_scale"]


def test_headline_only_is_the_profiled_command_and_its_traffic_is_a_real_launch():
    """`bench.py --headline-only` (what profiles/collect_r04.sh runs under rocprofv3): the timed region and nothing else, so every kernel
    symbol is launched at one size; at a profiled size the line's traffic comes from that launch class and is >= the algorithmic bytes."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    p = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--windows", "128", "--headline-only"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    o = _last_json(p.stdout)
    assert "alt" not in o and "sdd" not in o and "cpu_baseline" not in o
    r = o["roofline"]
    assert r["traffic"] is not None and r["traffic"] >= r["algorithmic_hbm_bytes_per_launch"]
    assert abs(r["traffic_over_algorithmic"] - r["traffic"] / r["algorithmic_hbm_bytes_per_launch"]) < 1e-9 and r["traffic_over_algorithmic"] < 4
    assert 0 < r["kernel_ms"] <= o["ms_per_step"] and 0.5 < r["frac"] < 1


@pytest.mark.parametrize("extra", [[], ["--train"], ["--shard", "agents", "--mno", "16"]])
def test_two_ranks_on_one_gpu(extra):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    # (the agent-sharded / configs[3] / configs[4] legs of the plain N > 1 line are test_gpus_flag_starts_its_own_ranks' business: switched off here)
    env = dict(os.environ, DESIRE_BENCH_ONE_GPU="1", DESIRE_BENCH_NO_AGENT_LEG="1", DESIRE_BENCH_NO_CONFIG3_LEG="1", DESIRE_BENCH_NO_CONFIG4_LEG="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "8"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    o = _last_json(p.stdout)
    assert o["n_gpus"] == 2 and o["value"] > 0
    if not extra:
        assert o["config"]["rows_per_gpu"] == 8 * 32 * 20
        assert abs(o["value"] - 2 * o["config"]["rows_per_gpu"] * 2 / (o["ms_per_step"] * 2e-3)) < 1e-6 * o["value"]


def _check_multi_rank_legs(o, world):
    """the plain `bench.py --gpus N` line carries the peer-buffer pass (on by default), BASELINE configs[3] at its named shape
    (32 scenes x 64 agents sharded 64/N per rank, K = 50, H = 256, fp32 and split) and configs[4] (512 agents per step over the ranks, training)."""
    pb = o["agent_sharded"]["peer_buffers"]
    assert "error" not in pb and "skipped" not in pb and pb["ioc_ms"] > 0, pb
    c3 = o["alt"]["config3"]
    assert "error" not in c3, c3
    assert ("%d slots per rank over %d ranks" % (64 // world, world)) in c3["shape"]
    for tag in ("fp32", "split_bf16x3"):
        assert c3[tag]["finite"] and c3[tag]["ms_per_step"] > 0 and c3[tag]["exposed_comm_ms"] >= 0, c3[tag]
        assert c3[tag]["peer_buffers"]["ioc_ms"] > 0, c3[tag]
    c4 = o["alt"]["config4_train"]
    assert "error" not in c4, c4
    for tag in ("fp32", "split_bf16x3"):
        assert c4[tag]["finite"] and c4[tag]["ms_per_step"] > 0 and c4[tag]["allreduce_ms"] > 0 and c4[tag]["allreduce_bytes"] > 4_000_000, c4[tag]
    assert abs(c4["fp32"]["loss"] - c4["split_bf16x3"]["loss"]) < 1e-3 * abs(c4["fp32"]["loss"])


@pytest.mark.slow
def test_eight_ranks_on_one_gpu_carry_every_leg():
    """The driver's SCALE command at N = 8, all ranks sharing this box's GPU over gloo (DESIRE_BENCH_ONE_GPU): 8 slots per rank of 64-agent scenes
    at configs[3], 2 windows per rank at configs[4]."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DESIRE_BENCH_ONE_GPU"] = "1"
    env["DESIRE_BENCH_LEG_TIMEOUT"] = "600"
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--windows", "8"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    o = _last_json(p.stdout)
    assert o["n_gpus"] == 8 and o["value"] > 0
    assert "error" not in o["agent_sharded"], o["agent_sharded"]
    _check_multi_rank_legs(o, 8)


def test_a_rank_dying_in_an_extra_leg_does_not_cost_the_line():
    """A GPU fault in one of the multi-rank legs is an abort of that rank, and the launcher then SIGTERMs the others: rank 0 must still print the
    line with the headline and the legs that had finished (benchlib/legs_dist.py: wake-up pipe + helper thread)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DESIRE_BENCH_ONE_GPU"] = "1"
    env["DESIRE_BENCH_FAULT_AT"] = "config3"
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "8"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    o = _last_json(p.stdout)
    assert o["n_gpus"] == 2 and o["value"] > 0
    assert "error" not in o["agent_sharded"] and o["agent_sharded"]["finite"]
    assert "error" in o["alt"]["config3"] and "config4_train" not in o["alt"]


def test_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE) must not die on plumbing -- it re-executes
    itself under torch.distributed.run, keeps the one-JSON-line contract, reports the rank count the collective library saw and
    carries the agent-sharded leg (per-step neighbour all-gather) in the same line."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DESIRE_BENCH_ONE_GPU"] = "1"
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "8"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    o = _last_json(p.stdout)
    assert o["n_gpus"] == 2 and o["value"] > 0 and o["config"]["rows_per_gpu"] == 8 * 32 * 20
    leg = o["agent_sharded"]
    assert "error" not in leg, leg
    assert leg["finite"] and leg["agents_per_scene_over_all_ranks"] == 64 and leg["bytes_received_per_rank_per_ioc_step"] == 2 * leg["bytes_sent_per_rank_per_ioc_step"]
    assert leg["ioc_ms_with_collectives"] > 0 and leg["exposed_comm_ms"] >= 0
    _check_multi_rank_legs(o, 2)
    # asking for more GPUs than the node has is an error message, not a hang (without the one-GPU test switch)
    env.pop("DESIRE_BENCH_ONE_GPU")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "64"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "visible" in (p.stdout + p.stderr)


def test_compact_flag_is_labelled():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    p = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--windows", "16", "--compact", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    o = _last_json(p.stdout)
    assert "row-compacted" in o["config"]["workload"] and "note" in o["roofline"] and o["value"] > 0


def test_split_flag_is_labelled_and_gated():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    p = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--windows", "16", "--split"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    o = _last_json(p.stdout)
    assert "split" in o["metric"] and o["dtype"].startswith("bf16x3") and "k_ioc_x3" in o["roofline"]["kernel"]
    assert abs(o["roofline"]["peak"] - 2500.0 / 3) < 1e-6 and 0 < o["roofline"]["frac"] < 1
    assert o["accuracy"]["max_abs_err_Y"] < 1e-4          # the HIP path with split operands against the fp32 oracle (gate 1e-3)
    p = subprocess.run([sys.executable, "bench.py", "--split", "--bf16"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
