#!/usr/bin/env python3
"""Per-kernel mean of every PMC counter per dispatch from rocprofv3 `--pmc ... -o name` sqlite databases.
Usage: summarise_pmc.py out.json a_results.db [b_results.db ...]   (counters from several passes are merged per kernel)"""
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
    q = (f"select s.kernel_name, i.name, avg(v) from (select e.event_id eid, e.pmc_id pid, sum(e.value) v from {pt} e "
         f"group by e.event_id, e.pmc_id) x join {kd} d on d.event_id=x.eid join {ks} s on s.id=d.kernel_id "
         f"join {pi} i on i.id=x.pid group by s.kernel_name, i.name")
    out = collections.defaultdict(dict)
    for k, n, v in c.execute(q):
        out[k][n] = v
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
    json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
