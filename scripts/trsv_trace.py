"""after scripts/prof_ilu.sh <tag>: per-launch durations of the triangular solves (one factor = k_fill_not_ready ...
last k_trsv_syncfree), split into far SpMV phases and near launches:  python scripts/trsv_trace.py gpurun_out/prof_ilu_<tag>"""
import csv, glob, sys, collections
d = sys.argv[1]
tr = sorted(glob.glob(d + "/*/*_kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cm::", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
        int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
solves, cur = [], None
for k, dur, t0, t1 in seq:
    if k.startswith("k_fill_not_ready"):
        if cur: solves.append(cur)
        cur = []
        continue
    if cur is None: continue
    if k.startswith(("k_trsv", "k_pb_phase")): cur.append((k, dur, t0, t1))
    elif k.startswith(("k_half", "k_full", "k_update_p", "k_check", "k_perm", "k_init")):
        solves.append(cur); cur = None
if cur: solves.append(cur)
# cut each solve after its last k_trsv launch (what follows is the loop's own SpMV)
clean = []
for s in solves:
    last = max((i for i, e in enumerate(s) if e[0].startswith("k_trsv")), default=-1)
    s = s[:last + 1]
    if s and all(e[1] > 3.0 for e in s if e[0].startswith("k_trsv")): clean.append(s)      # (frozen launches are no-ops of ~2 us)
print("factor solves traced:", len(clean))
if not clean: sys.exit(0)
L = collections.Counter(len(s) for s in clean).most_common(1)[0][0]
sel = [s for s in clean if len(s) == L]
for which, name in ((0, "first factor of a pair (L)"), (1, "second (U)")):
    grp = sel[which::2]
    if not grp: continue
    wall = sum(s[-1][3] - s[0][2] for s in grp) / len(grp) / 1e3
    near = sum(sum(e[1] for e in s if e[0].startswith("k_trsv")) for s in grp) / len(grp)
    p1 = sum(sum(e[1] for e in s if e[0].startswith("k_pb_phase1")) for s in grp) / len(grp)
    p2 = sum(sum(e[1] for e in s if e[0].startswith("k_pb_phase2")) for s in grp) / len(grp)
    print("--- %s: %d solves; wall %.1f us = near %.1f + far phase 1 %.1f + far phase 2 %.1f + gaps %.1f" % (name, len(grp), wall, near, p1, p2, wall - near - p1 - p2))
    print("    per launch:", "  ".join("%s %.0f" % (grp[0][i][0].replace("k_trsv_syncfree", "near").replace("k_pb_phase", "p")[:12], sum(s[i][1] for s in grp) / len(grp)) for i in range(L)))
