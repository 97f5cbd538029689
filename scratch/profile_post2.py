"""Post-process scratch/profile_round2.sh output into the files committed under profiles/ (usage: tag name)."""
import collections, csv, os, shutil, sys
tag, name = sys.argv[1], sys.argv[2]   # e.g. a r2_a_cls_bs64
src = f"gpurun_out/prof_{tag}"
shutil.copy(f"{src}/stats/s_kernel_stats.csv", f"profiles/{name}_kernel_stats.csv")
shutil.copy(f"{src}/bench.json", f"profiles/{name}_bench.json")
shutil.copy(f"{src}/timeline.txt", f"profiles/{name}_timeline.txt")
shutil.copy(f"{src}/mfma_util.txt", f"profiles/{name}_pmc_mfma_util.txt")
def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        agg[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg
f = load(f"{src}/fetch/f_counter_collection.csv", "FETCH_SIZE")
w = load(f"{src}/write/w_counter_collection.csv", "WRITE_SIZE")
rows = []
for key, v in f.items():
    ww = w.get(key, [0.0])
    rows.append((sum(v) * 2 / 1024, key[0], key[1], len(v), sum(v) / len(v) * 2 / 1024, sum(ww) / len(ww) / 1024))
rows.sort(reverse=True)
with open(f"profiles/{name}_pmc_hbm_traffic.txt", "w") as o:
    o.write("# HBM traffic per kernel launch (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, bench.py --steps 3 --warmup 2).\n"
            "# FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section); counter unit KB.\n"
            "# Columns: kernel | grid(threads) | launches | fetch MB per launch (x2 corrected) | write MB per launch\n")
    for tot, kn, grid, n, fm, wm in rows[:40]:
        o.write(f"{kn[:84]:84s} | {grid:8d} | {n:4d} | {fm:9.1f} | {wm:9.1f}\n")
print(open(f"profiles/{name}_pmc_hbm_traffic.txt").read()[:2500])
print(open(f"profiles/{name}_pmc_mfma_util.txt").read()[:3000])
