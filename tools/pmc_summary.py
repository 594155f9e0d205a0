"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected in separate runs) into per-kernel HBM bytes.

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
Units / corrections follow MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, so the read side is doubled.
"""
import collections, csv, json, re, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


SKIP_STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # leading steps to drop (eager priming + capture of a graph-mode run)
STEPS = {}


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    seen = set()
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    # one log-mel launch per step: everything dispatched before the (SKIP_STEPS + 1)-th one is dropped
    marks = sorted({int(r["Dispatch_Id"]) for r in rows if "logmel" in r["Kernel_Name"]})
    first = marks[SKIP_STEPS] if SKIP_STEPS and len(marks) > SKIP_STEPS else -1
    STEPS[counter] = len([m for m in marks if m >= first])
    for r in rows:
        if int(r["Dispatch_Id"]) < first:
            continue
        k = short(r["Kernel_Name"])
        acc[k][1] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key)
            acc[k][0] += 1
    return acc


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    nf, f = fetch.get(k, [0, 0.0])
    nw, w = write.get(k, [0, 0.0])
    out[k] = {"launches": max(nf, nw),
              "read_bytes_per_launch": round(2.0 * 1024.0 * f / nf) if nf else None,      # x2: gfx950 correction
              "write_bytes_per_launch": round(1024.0 * w / nw) if nw else None}
rd = sum((v["read_bytes_per_launch"] or 0) * fetch.get(k, [0])[0] for k, v in out.items())
wr = sum((v["write_bytes_per_launch"] or 0) * write.get(k, [0])[0] for k, v in out.items())
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KiB -> bytes, FETCH_SIZE x2 on gfx950"
                     + (f"; the first {SKIP_STEPS} steps (eager priming + graph capture) dropped" if SKIP_STEPS else ""),
           "steps": STEPS, "read_bytes_per_step": round(rd / max(STEPS.get("FETCH_SIZE", 1), 1)),
           "write_bytes_per_step": round(wr / max(STEPS.get("WRITE_SIZE", 1), 1)),
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
print(f"per step: read {rd / max(STEPS.get('FETCH_SIZE', 1), 1) / 1e9:.3f} GB, written {wr / max(STEPS.get('WRITE_SIZE', 1), 1) / 1e9:.3f} GB "
      f"({STEPS} steps counted)")
for k, v in sorted(out.items(), key=lambda kv: -((kv[1]["read_bytes_per_launch"] or 0) * kv[1]["launches"]))[:25]:
    print(f"{v['launches']:5d}  rd {(v['read_bytes_per_launch'] or 0) / 1e6:9.2f} MB  wr {(v['write_bytes_per_launch'] or 0) / 1e6:9.2f} MB  {k[:90]}")
