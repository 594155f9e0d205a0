"""Print the kernel timeline of one steady-state step from a rocprofv3 kernel-trace CSV: start offset, duration, queue, grid, name."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(ev) if "logmel" in r["Kernel_Name"]]
# bench.py: priming steps, warm-up, the timed graph replays, then FIVE eagerly re-issued steps for the per-kernel events - take a
# step from the middle of the run (a graph replay), not from its tail
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a, b = starts[k], starts[k + 1]
t0 = int(ev[a]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.search(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)(I|E)", n)
    return (m.group(1) if m else n.split("(")[0])[:34]
for r in ev[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if r.get("Grid_Size_X"):                       # total work-items of the launch (all three grid dimensions)
        grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1)
    else:
        grid = r.get("Grid_Size") or ""
    print(f"{s/1e3:9.1f} {(e-s)/1e3:8.1f} q{r.get('Queue_Id','?'):>3} g{grid:>9} {short(r['Kernel_Name'])}")
print("step wall %.1f us" % ((int(ev[b]["Start_Timestamp"]) - t0) / 1e3))
