#!/usr/bin/env python3
"""Per-LAUNCH-CLASS mean of every PMC counter per dispatch from rocprofv3 `--pmc ... -o name` sqlite databases.  A class is
(kernel symbol, workgroups in the grid, workgroup size): the JSON key is "<symbol> [wgs=N,wg=M]" and every entry carries
`workgroups`, `workgroup_size`, `dispatches`, so a reader (bench.py committed_traffic) can pick the launch it means and check
SQ_WAVES against it.  Usage: summarise_pmc.py out.json a_results.db [b_results.db ...]  (counters of several passes merge per class)"""
import collections
import json
import sqlite3
import sys


def pmc(path):
    db = sqlite3.connect(path)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda s: [t for t in tabs if s in t][0]
    pt, kd, ks, pi = T("pmc_event"), T("kernel_dispatch"), T("kernel_symbol"), T("info_pmc")
    wgs = "(d.grid_size_x / d.workgroup_size_x) * (d.grid_size_y / d.workgroup_size_y) * (d.grid_size_z / d.workgroup_size_z)"
    q = (f"select s.kernel_name, {wgs} as wgs, d.workgroup_size_x * d.workgroup_size_y * d.workgroup_size_z as wg, i.name, avg(v), count(*) "
         f"from (select e.event_id eid, e.pmc_id pid, sum(e.value) v from {pt} e group by e.event_id, e.pmc_id) x "
         f"join {kd} d on d.event_id=x.eid join {ks} s on s.id=d.kernel_id join {pi} i on i.id=x.pid "
         f"group by s.kernel_name, wgs, wg, i.name")
    out = collections.defaultdict(dict)
    for k, nwg, wg, n, v, cnt in c.execute(q):
        e = out["%s [wgs=%d,wg=%d]" % (k, nwg, wg)]
        e[n] = v
        e["workgroups"], e["workgroup_size"], e["dispatches"] = int(nwg), int(wg), int(cnt)
    return out


def main():
    res = collections.defaultdict(dict)
    for p in sys.argv[2:]:
        for k, v in pmc(p).items():
            res[k].update(v)
    for k, v in res.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE"):
            v["MfmaUtil"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["HBM_bytes_2xFETCH_plus_WRITE"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        if "SQ_WAVES" in v:
            v["waves_expected"] = v["workgroups"] * ((v["workgroup_size"] + 63) // 64)
    json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
