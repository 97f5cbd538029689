"""Per-kernel MFMA-pipe utilisation from a rocprofv3 --pmc counter_collection.csv (scratch/pmc_mfma.sh).
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3
sums the counter over the 8 XCDs: MI355X_MICROARCH.md 'DVFS give-back').  The profiler serialises dispatches while it
collects counters, so these are stand-alone per-kernel figures, weighted here by launch count x duration."""
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n[:96]
rows = []
for kn, cs in agg.items():
    n = len(cs.get("GRBM_GUI_ACTIVE", []))
    if not n: continue
    gui = sum(cs["GRBM_GUI_ACTIVE"]) / 8.0
    mfma = sum(cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))
    wave = sum(cs.get("SQ_WAVE_CYCLES", [0])); wait = sum(cs.get("SQ_WAIT_ANY", [0])); winst = sum(cs.get("SQ_WAIT_INST_ANY", [0]))
    act = sum(cs.get("SQ_ACTIVE_INST_ANY", [0]))
    rows.append((gui, short(kn), n, mfma / (gui * 1024) if gui else 0.0, wait / wave if wave else 0, winst / wave if wave else 0, act / wave if wave else 0))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("# kernel | launches | share of GPU cycles | MFMA pipe busy | wave cycles: parked (s_waitcnt/barrier) | issue-stalled | issuing")
for gui, kn, n, mu, wa, wi, ac in rows[:30]:
    print(f"{kn:96s} | {n:5d} | {gui / tot:6.1%} | {mu:6.1%} | {wa:6.1%} | {wi:6.1%} | {ac:6.1%}")
w = sum(r[0] * r[3] for r in rows) / tot if tot else 0
print(f"# whole step, cycle-weighted MFMA pipe busy: {w:.1%}  (dense bf16 peak = 100 %: every SIMD issuing one 32x32x16 MFMA per 32 cycles)")
