"""Shared constants and small helpers of bench.py's legs (peaks from /opt/skills/guides/MI355X_MICROARCH.md)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16 (v_mfma_f32_32x32x16_bf16)
HBM_PEAK_GBS = 8000.0


def ioc_flops_per_row(d):
    """Algorithmic FLOPs of k_ioc per row (one (agent,k) trajectory), SURVEY.md 8(d) D4 IOC terms."""
    H, E, B, T = d.H, d.E, d.B, d.T_pred
    return d.iters * (T * (6.0 * H * (E + H) + 2.0 * B * H * H + 2 * H + 4 * d.E_v) + 2.0 * H * 2 * T)


def committed_traffic(d, bf16=False):
    """HBM bytes per launch of the dominant IOC kernel from the committed rocprofv3 PMC passes (profiles/, collected from
    `bench.py --headline-only` at the same shape in separate --pmc runs): 2 x FETCH_SIZE (gfx950 counts wide reads at half,
    MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KiB -> bytes.  The summaries are keyed per LAUNCH CLASS (symbol, workgroups,
    workgroup size: profiles/summarise_pmc.py), and only the class whose grid is THIS launch's -- ceil(R / 32) tiles of 4 waves --
    is accepted; a summary without grid information (rounds 1-2) is accepted only if its SQ_WAVES equals that wave count.  A figure
    below the algorithmic bytes is physically impossible for the launch and is refused (a mixed-size average got
    through).  None when no matching profile is committed."""
    import glob
    import re
    committed_traffic.source = None
    tiles = (d.R + 31) // 32
    waves = tiles * 4
    algorithmic = d.R * (2 * d.T_pred * 2 * 4 + 4) + d.A * d.H * 4
    sym = "k_ioc_bf16" if bf16 else "k_ioc"

    def is_headline_kernel(name):
        """Mangled (`_Z5k_iocILi128ELi16ELi32ELi32ELb0ELb0ELi1EEv7IocArgs.kd`) or demangled (`void k_ioc<128, 16, 32, 32, false, false, 1>(IocArgs)`)
        symbol of the plain inference form: this H, no saving / compact flag set, one workgroup per tile."""
        m = re.match(r"^_Z\d+%s((?:I|L[ib]\d+E)+)E" % sym, name)
        if m:
            targs = re.findall(r"L([ib])(\d+)E", m.group(1))
        else:
            m = re.match(r"^void %s<([^>]*)>" % sym, name)
            if not m:
                return False
            targs = [("b", {"true": "1", "false": "0"}[x.strip()]) if x.strip() in ("true", "false") else ("i", x.strip()) for x in m.group(1).split(",")]
        ints = [int(v) for k, v in targs if k == "i"]
        flags = [int(v) for k, v in targs if k == "b"]
        if not ints or ints[0] != d.H or any(flags):
            return False
        return not (sym == "k_ioc" and len(ints) >= 5 and ints[4] != 1)            # NSPL > 1 = the bin-split form of few-window launches
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*pmc_per_kernel.json")), reverse=True):     # newest round first
        base = os.path.basename(path)
        if ("bf16" in base) != bool(bf16) or "x6" in base or "split" in base or "train" in base:
            continue
        with open(path) as fh:
            pmc = json.load(fh)
        for name, c in pmc.items():
            if not is_headline_kernel(name) or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                continue
            if "workgroups" in c:
                if c["workgroups"] != tiles:
                    continue
            elif int(round(c.get("SQ_WAVES", -1))) != waves:
                continue
            b = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            if b < algorithmic:
                continue
            committed_traffic.source = "profiles/%s : %s" % (base, name)
            return b
    return None

committed_traffic.source = None


def sdd_windows(n_windows, mno):
    """Real Stanford Drone Dataset windows for the `sdd` leg: the committed 160-frame slice of bookstore/video6 (the reference
    loader's own preprocessing of it, tests/golden/loader_bookstore6_T48.npz: data0 [160, 32, 3] = [id, x_px, y_px]) cut into
    every 8 + 40-frame window with the loader's slot assignment, tiled to n_windows.  Absent slots stay absent (7.5 objects per
    frame on average, BASELINE configs[1]'s "SDD bookstore")."""
    from desire_amd.data_loader import window_to_slots
    g = np.load(os.path.join(ROOT, "tests", "golden", "loader_bookstore6_T48.npz"))
    frames = g["data0"]
    wins = []
    for s0 in range(0, frames.shape[0] - 48, 4):
        src, _ = window_to_slots(frames[s0:s0 + 49], 48, frames.shape[1])
        wins.append(np.pad(src[:, :mno], ((0, 0), (0, max(0, mno - src.shape[1])), (0, 0))))
    wins = np.stack(wins).astype(np.float32)                               # [n_real, 48, mno, 3]
    idx = np.arange(n_windows) % wins.shape[0]
    return np.ascontiguousarray(wins[idx, :8]), np.ascontiguousarray(wins[idx, 8:]), wins.shape[0]


HBM_ACHIEVABLE_GBS = 6300.0          # what a streaming kernel reaches on this part (measured copy rate), for the HBM floor


def committed_train_traffic(split):
    """Memory-side bytes of ONE training step from the newest committed rocprofv3 set of `bench.py --train [--split] --headline-only`:
    sum over launch classes of (calls per step, from <set>_kernel_stats.csv) x (2 x FETCH_SIZE + WRITE_SIZE per launch, from
    <set>_pmc_per_kernel.json; the guide's gfx950 correction).  Returns (bytes_per_step, source, per_kernel_top) or (None, None, None)."""
    import csv
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_train%s_pmc_per_kernel.json" % ("_split" if split else ""))), reverse=True):
        stats = path.replace("_pmc_per_kernel.json", "_kernel_stats.csv")
        bench = path.replace("_pmc_per_kernel.json", "_bench.json")
        if not (os.path.exists(stats) and os.path.exists(bench)):
            continue
        with open(bench) as fh:
            b = json.loads(fh.read().strip().splitlines()[-1])
        per_run = b["steps"] + b["warmup"]
        with open(path) as fh:
            pmc = json.load(fh)
        total, rows = 0.0, []
        with open(stats) as fh:
            for r in csv.DictReader(fh):
                key = "%s [wgs=%s,wg=%s]" % (r["kernel"], r["workgroups"], r["workgroup_size"])
                c = pmc.get(key)
                if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                    continue
                per_step = float(r["calls"]) / per_run
                by = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 * per_step
                total += by
                rows.append((by, r["kernel"][:60], per_step, float(r["avg_ms"]), c.get("MfmaUtil")))
        if total > 0:
            rows.sort(reverse=True)
            top = [{"kernel": k, "GB_per_step": round(by / 1e9, 3), "launches_per_step": round(n, 2), "avg_ms": ms, "MfmaUtil": mu} for by, k, n, ms, mu in rows[:6]]
            return total, "profiles/" + os.path.basename(path), top
    return None, None, None
