"""Timeline summary of the 20 timed steps in a rocprofv3 kernel trace of `bench.py --steps 20 --warmup 5`."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'mn_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
S = lambda r: int(r['Start_Timestamp']) / 1e3
E = lambda r: int(r['End_Timestamp']) / 1e3
signs = [i for i, r in enumerate(rows) if 'mn_cc_sign' in r['Kernel_Name']]
# groups of sweeps separated by idle gaps of the whole GPU > 150 us
groups, cur = [], [signs[0]]
for a, b in zip(signs, signs[1:]):
    busy_end = max(E(r) for r in rows[a:b])
    if S(rows[b]) - busy_end > 150:
        groups.append(cur); cur = []
    cur.append(b)
groups.append(cur)
print("groups of sweeps (by idle gaps):", [len(g) for g in groups[-8:]])
for g in groups:
    if len(g) == 20:
        t0 = S(rows[g[0]])
        last = g[-1]
        end = max(E(r) for r in rows[g[0]:] if S(r) < S(rows[last]) + 2000 and (groups.index(g) == len(groups) - 1 or S(r) < S(rows[groups[groups.index(g) + 1][0]])))
        print("20-step window: first sweep starts at 0, last sweep starts at %.1f us, everything done at %.1f us" % (S(rows[last]) - t0, end - t0))
        starts = [S(rows[i]) - t0 for i in g]
        print("sweep starts:", " ".join("%.0f" % s for s in starts))
        mq = rows[g[0]]['Queue_Id']
        mainq = [r for r in rows[g[0]:] if r['Queue_Id'] == mq and S(r) <= S(rows[last]) + 200]
        busy = sum(E(r) - S(r) for r in mainq)
        print("main queue: %d kernels, busy %.0f us until its last kernel ends at %.0f us" % (len(mainq), busy, max(E(r) for r in mainq) - t0))
