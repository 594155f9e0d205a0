"""Print the last burst of kernels (bursts are separated by > 2 ms of idle) of a rocprofv3 kernel-trace CSV."""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
bursts, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 2_000_000:
        bursts.append(cur); cur = []
    cur.append(b)
bursts.append(cur)
ev = bursts[-1]
t0 = int(ev[0]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.search(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)(I|E)", n)
    return (m.group(1) if m else n.split("(")[0])[:34]
for r in ev:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {(e-s)/1e3:8.1f} q{r.get('Queue_Id','?'):>3} {short(r['Kernel_Name'])}")
print("burst wall %.1f us, %d kernels" % ((int(ev[-1]["End_Timestamp"]) - t0) / 1e3, len(ev)))
