"""Per-stream timeline of one training step from a rocprofv3 --kernel-trace database."""
import sqlite3, sys, re, collections
db = sys.argv[1]; step_sel = int(sys.argv[2]) if len(sys.argv) > 2 else -2
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
ks = [t for t in tabs if 'kernel_symbol' in t][0]; kd = [t for t in tabs if 'kernel_dispatch' in t][0]
rows = list(c.execute(f"select d.start, d.end, d.queue_id, d.stream_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I", n)
    if m: return m.group(1)
    return n[:60]
# steps delimited by adamw_tick
ticks = [i for i, r in enumerate(rows) if 'adamw_tick' in r[4]]
a, b = ticks[step_sel], ticks[step_sel + 1]
step = rows[a:b]
t0, t1 = step[0][0], step[-1][1]
print(f"step wall {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels")
byq = collections.defaultdict(list)
for r in step: byq[(r[2], r[3])].append(r)
for q, rs in byq.items():
    busy = sum(r[1] - r[0] for r in rs)
    gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
    print(f"queue/stream {q}: {len(rs)} kernels, busy {busy / 1e6:.3f} ms, first {(rs[0][0]-t0)/1e6:.3f} last {(rs[-1][1]-t0)/1e6:.3f}; small gaps(<20us) sum {sum(g for g in gaps if 0 < g < 20000)/1e6:.3f} ms n={sum(1 for g in gaps if 0<g<20000)}; big gaps sum {sum(g for g in gaps if g >= 20000)/1e6:.3f}")
# union busy time (any stream)
ev = sorted([(r[0], 1) for r in step] + [(r[1], -1) for r in step])
cur = 0; last = t0; idle = 0; both = 0
for t, d in ev:
    if cur == 0: idle += t - last
    if cur >= 2: both += t - last
    cur += d; last = t
print(f"GPU idle (no kernel running) {idle / 1e6:.3f} ms; >=2 kernels overlapping {both / 1e6:.3f} ms")
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    k = short(r[4]) + f" q{r[2]}"; agg[k][0] += 1; agg[k][1] += r[1] - r[0]
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  {t / 1e3:9.1f} us  n={n:4d}  avg {t / n / 1e3:7.1f}  {k}")
if len(sys.argv) > 3:
    print("---- gaps >= 8us on main stream, and timeline excerpt")
    mq = max(byq, key=lambda q: len(byq[q])); rs = byq[mq]
    for i in range(len(rs) - 1):
        g = rs[i + 1][0] - rs[i][1]
        if g >= 8000: print(f"  gap {g/1e3:7.1f} us at t={(rs[i][1]-t0)/1e6:.3f} ms after {short(rs[i][4])[:30]} before {short(rs[i+1][4])[:30]}")
    lo, hi = [float(x) for x in sys.argv[3].split(",")]
    for r in step:
        ts = (r[0] - t0) / 1e6
        if lo <= ts <= hi: print(f"  {ts:8.3f} +{(r[1]-r[0])/1e3:7.1f}us q{r[2]} {short(r[4])[:40]}")
