#!/usr/bin/env python3
"""Timeline of consecutive decode tokens out of a rocprofv3 (rocpd sqlite) kernel trace: where the time of one token goes --
inside kernels (begin..end of each dispatch) or between them (end of one dispatch .. begin of the next on the same stream).
A token = the dispatches between two finish_token_kernel dispatches; only tokens of a graph-replayed decode loop are used
(as many gate+up GEMVs as the model has layers: 5 dispatches per layer; 6 in reference order with the three-launch attention).
Usage: tools/rocpd_timeline.py <results.db> [n_tokens=4] [reference-order|order-free]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ntok = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()


def short(name):
    s = re.sub(r"\(.*", "", name).replace("void jh::", "").replace("jh::", "")
    return s


rows = [(short(n), s, e) for n, s, e in rows]
# token boundaries.  Two replayed loops can be in a trace: the reference-order one (jh_p16.h / jh_t16.h kernels: 6 dispatches per layer --
# q|k|v, scores, softmax + values, o, gate|up, down) and the order-free one (5 per layer); `want` picks which (argv[3], default: the
# reference-order loop if the trace has one)
cuts = [i for i, r in enumerate(rows) if r[0].startswith("finish_token_kernel")]
by_kind = {"reference-order": [], "order-free": []}
for a, b in zip(cuts, cuts[1:]):
    seg = rows[a + 1:b + 1]          # dispatches after the previous finish, up to and including this token's finish
    names = [r[0] for r in seg]
    gu = sum(1 for n in names if re.match(r"gemv_t16_kernel<1, 2,", n) or re.match(r"gemv_i8q4_kernel<1, 2,", n))
    strict = any("p16" in n for n in names)
    per_layer = 6 if strict and not any(n.startswith("attn_p16_fused") for n in names) else 5   # one-launch reference-order attention: 5
    if gu == 0 or len(seg) != per_layer * gu + 2:              # + LM head + finish: the replayed decode graph only
        continue
    by_kind["reference-order" if strict else "order-free"].append((a, seg))
want = sys.argv[3] if len(sys.argv) > 3 else ("reference-order" if by_kind["reference-order"] else "order-free")
tokens = by_kind[want]
print(f"{want} decode loop ({len(by_kind['reference-order'])} reference-order and {len(by_kind['order-free'])} order-free replayed tokens in the trace)\n")
# the longest run of consecutive tokens, taken from its middle
runs, cur_run = [], []
for t in tokens:
    if cur_run and t[0] != cur_run[-1][0] + len(cur_run[-1][1]):
        runs.append(cur_run)
        cur_run = []
    cur_run.append(t)
if cur_run:
    runs.append(cur_run)
if not runs:
    print("no replayed decode tokens found in the trace")
    sys.exit(1)
run = max(runs, key=len)
mid = max(0, len(run) // 2 - ntok // 2)
pick = run[mid:mid + ntok]
print(f"{len(tokens)} replayed decode tokens in the trace, longest consecutive run {len(run)}; tokens {mid}..{mid + len(pick) - 1} of that run below\n")
print("| token | dispatches | first begin -> last end (us) | sum of kernel durations (us) | sum of gaps (us) | largest gap (us) | previous finish end -> first begin (us) |")
print("|---|---|---|---|---|---|---|")
agg = {}
for j, (a, seg) in enumerate(pick):
    span = (seg[-1][2] - seg[0][1]) / 1e3
    dur = sum(e - s for _, s, e in seg) / 1e3
    gaps = [(seg[i + 1][1] - seg[i][2]) / 1e3 for i in range(len(seg) - 1)]
    lead = (seg[0][1] - rows[a][2]) / 1e3
    print(f"| {mid + j} | {len(seg)} | {span:.1f} | {dur:.1f} | {sum(gaps):.1f} | {max(gaps):.2f} | {lead:.2f} |")
    for i, (n, s, e) in enumerate(seg):
        g = (s - (seg[i - 1][2] if i else rows[a][2])) / 1e3
        x = agg.setdefault(n, [0, 0.0, 0.0])
        x[0] += 1; x[1] += (e - s) / 1e3; x[2] += g
print("\nper kernel over those tokens (gap = end of the previous dispatch -> begin of this one):\n")
print("| kernel | dispatches / token | avg duration (us) | avg gap before (us) | us / token in kernel | us / token in gaps |")
print("|---|---|---|---|---|---|")
n = len(pick)
for k, x in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| {k} | {x[0] / n:.0f} | {x[1] / x[0]:.2f} | {x[2] / x[0]:.2f} | {x[1] / n:.1f} | {x[2] / n:.1f} |")
tot_d = sum(x[1] for x in agg.values()) / n
tot_g = sum(x[2] for x in agg.values()) / n
print(f"\nper token: {tot_d:.1f} us inside kernels + {tot_g:.1f} us between them = {tot_d + tot_g:.1f} us")
