#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run (rocpd sqlite): average counter value per kernel.  Usage: rocpd_pmc.py db [out.md]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
def cols(t): return [r[1] for r in cur.execute(f"pragma table_info({t})")]
cand = [v for v in ("counters_collection", "pmc_events") if v in views]
lines = []
for v in cand:
    c = cols(v)
    lines.append(f"<!-- {v}: {c} -->")
    name = next((x for x in c if x in ("kernel_name", "name")), None)
    cname = next((x for x in c if x in ("counter_name", "pmc_name", "symbol")), None)
    val = next((x for x in c if x in ("value", "counter_value")), None)
    if not (name and cname and val):
        continue
    rows = cur.execute(f"select {name}, {cname}, count(*), avg({val}), min({val}), max({val}) from {v} group by {name}, {cname}").fetchall()
    lines += ["| kernel | counter | dispatches | avg | min | max |", "|---|---|---|---|---|---|"]
    for k, cn, n, a, mn, mx in sorted(rows, key=lambda r: -r[3] * r[2]):
        k = re.sub(r"\(.*", "", k).replace("void jh::", "").replace("jh::", "")
        if "at::native" in k or "rocclr" in k: continue
        lines.append(f"| {k} | {cn} | {n} | {a:.1f} | {mn:.1f} | {mx:.1f} |")
    break
out = "\n".join(lines); print(out)
if len(sys.argv) > 2: open(sys.argv[2], "w").write(out + "\n")
