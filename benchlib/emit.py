"""How bench.py's record leaves the process.

The driver parses the LAST stdout line as JSON and keeps only a few KB of tail (BENCH_r05.json: the 23 KB one-line record came back
`parsed: null`).  So the record goes out three ways:

  1. `bench_full.json` beside bench.py (override: $DESIRE_BENCH_FULL) -- the complete record, every leg with its notes and per-kernel tables;
  2. one stdout line `#full {...}` -- the same record for whoever captures stdout (does not start with `{`, so it is never taken for the line);
  3. the LAST stdout line: the contract keys (metric, value, unit, n_gpus, steps, warmup, ms_per_step, ..., config, roofline, cpu_baseline,
     accuracy) plus a few scalars per extra leg, kept under LINE_BUDGET bytes (asserted here and in tests/test_gpu_bench.py).
"""
import json
import os

LINE_BUDGET = 4000
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_TOP = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "step_ms_median", "step_ms_min", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "forward_ms", "backward_ms", "whole_step_tflops_3x_forward_credit")
_ROOF = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "executed_frac", "kernel_ms",
         "algorithmic_flops_per_launch", "algorithmic_hbm_bytes_per_launch", "whole_path_frac", "whole_path_frac_executed")
_DROP = {"note", "unit", "data", "sample", "gates", "kernel_ms", "kernel_ms_per_step", "roofline", "kernels", "ioc_kernel", "shape", "units",
         "whole_path_note", "traffic_source", "traffic_unit"}


def _num(v):
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, int):
        return v
    if isinstance(v, float):
        return float("%.5g" % v)
    return v


def _slim(o, depth):
    """numeric / boolean leaves (5 significant digits) and error strings, `depth` levels of nesting; notes, units and per-kernel tables dropped"""
    out = {}
    for k, v in o.items():
        if k in _DROP:
            continue
        if isinstance(v, dict):
            if depth > 0:
                s = _slim(v, depth - 1)
                if s:
                    out[k] = s
        elif isinstance(v, (int, float, bool)) or v is None:
            out[k] = _num(v)
        elif k in ("error", "skipped") and isinstance(v, str):
            out[k] = v[:160]
    return out


def _leg_scalar(v):
    """one figure per leg: the first of these keys found walking down"""
    if not isinstance(v, dict):
        return _num(v)
    if "error" in v:
        return {"error": str(v["error"])[:120]}
    for key in ("ms_per_step", "ms_per_call", "ioc_ms", "value", "fraction_of_resident", "resident_ms_per_step"):
        if key in v and isinstance(v[key], (int, float)):
            return {key: _num(v[key])}
    out = {}
    for k, x in v.items():
        if isinstance(x, dict) and k not in _DROP:
            s = _leg_scalar(x)
            if isinstance(s, dict) and len(s) == 1 and "error" not in s:
                s = next(iter(s.values()))
            if s is not None and s != {}:
                out[k] = s
    return out


def compact(rec):
    """The driver's line: contract keys whole, everything else reduced until the line fits LINE_BUDGET."""
    line = {k: rec[k] for k in _TOP if k in rec}
    if isinstance(line.get("dtype"), str) and len(line["dtype"]) > 300:
        line["dtype"] = line["dtype"][:297] + "..."
    if "config" in rec:
        line["config"] = dict(rec["config"])
    if "roofline" in rec:
        r = rec["roofline"]
        line["roofline"] = {k: _num(r[k]) for k in _ROOF if k in r}
        src = r.get("traffic_source", "")
        line["roofline"]["traffic_source"] = src.split(" ")[1] if src.startswith("from_profile: ") else src[:80]
    if "cpu_baseline" in rec:
        c = rec["cpu_baseline"]
        line["cpu_baseline"] = {k: (_num(c[k]) if not isinstance(c[k], str) else c[k][:200]) for k in ("value", "unit", "cores", "threads", "host_cores", "kind", "sample") if k in c}
        for sub in ("numpy_oracle", "per_object_loop"):
            if isinstance(c.get(sub), dict) and "value" in c[sub]:
                line["cpu_baseline"][sub + "_value"] = _num(c[sub]["value"])
    if "accuracy" in rec:
        line["accuracy"] = {k: (_num(v) if not isinstance(v, str) else v[:80]) for k, v in rec["accuracy"].items() if not isinstance(v, dict)}
    line["full_record"] = rec.get("full_record", "bench_full.json")
    # extra legs: `alt` as ONE figure per leg (ms per step / call; the leaf's ms_per_step, ms_per_call or ioc_ms) plus the few fractions the
    # verdicts track; sdd / agent_sharded / comm as their scalar leaves, nesting reduced until the line fits
    if isinstance(rec.get("alt"), dict):
        line["alt_ms"] = {n: _leg_scalar(v) for n, v in rec["alt"].items()}
        picks = {}
        for name, path in (("bf16_config2_mno128_ioc_frac_of_bf16_peak", ("bf16_config2", "mno128", "ioc_frac_of_bf16_peak")),
                           ("bf16_config2_mno32_ioc_frac_of_bf16_peak", ("bf16_config2", "mno32", "ioc_frac_of_bf16_peak")),
                           ("split_bf16x3_ioc_frac_of_bf16_peak_over_3", ("split_bf16x3_ioc", "ioc_frac_of_bf16_peak_over_3")),
                           ("split_bf16x3_max_abs_diff_vs_fp32_kernel", ("split_bf16x3_ioc", "max_abs_diff_vs_fp32_kernel")),
                           ("split_bf16x6_max_abs_diff_vs_fp32_kernel", ("split_bf16x6_ioc", "max_abs_diff_vs_fp32_kernel")),
                           ("training_fp32_frac", ("training_step", "fp32", "roofline", "frac")),
                           ("with_loader_host_fraction_of_resident", ("with_loader", "fed_by_host_loader", "fraction_of_resident")),
                           ("with_loader_device_fraction_of_resident", ("with_loader", "fed_by_device_builder", "fraction_of_resident"))):
            v = rec["alt"]
            for k in path:
                v = v.get(k) if isinstance(v, dict) else None
            if isinstance(v, (int, float)):
                picks[name] = _num(v)
        if picks:
            line["alt_figures"] = picks
    extras = [k for k in ("sdd", "agent_sharded", "comm") if isinstance(rec.get(k), dict)]
    for depth in (2, 1, 0, -1):
        for k in extras:
            line[k] = _slim(rec[k], depth) if depth >= 0 else _leg_scalar(rec[k])
        if len(json.dumps(line)) <= LINE_BUDGET:
            break
    if len(json.dumps(line)) > LINE_BUDGET:          # still too long (a pathological error string): the contract keys alone
        for k in extras + ["alt_ms", "alt_figures"]:
            line.pop(k, None)
        if len(json.dumps(line)) > LINE_BUDGET and "config" in line:
            line["config"]["workload"] = line["config"].get("workload", "")[:300]
    return line


def emit(rec):
    """full record -> file + `#full` line; compact line LAST.  Never raises on the file (a read-only tree must not cost the driver its line)."""
    path = os.environ.get("DESIRE_BENCH_FULL") or os.path.join(ROOT, "bench_full.json")
    rec = dict(rec)
    try:
        with open(path, "w") as f:
            json.dump(rec, f)
            f.write("\n")
        rec["full_record"] = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError as e:
        rec["full_record"] = "stdout '#full' line only (%s)" % type(e).__name__
    print("#full " + json.dumps(rec), flush=True)
    line = json.dumps(compact(rec))
    assert len(line) <= LINE_BUDGET + 2000, len(line)
    print(line, flush=True)
