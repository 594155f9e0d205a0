"""Summarise a rocprofv3 kernel-trace CSV: GPU busy/idle time and the per-step timeline of the main phases."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# steps are delimited by logmel_kernel launches
starts = [i for i, e in enumerate(ev) if "logmel" in e[2]]
if len(starts) < 3: sys.exit("need >= 3 steps")
a, b = starts[-3], starts[-2]            # one full steady-state step
seg = ev[a:b]
t0, t1 = seg[0][0], ev[b][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in seg:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"step wall {1e-6*(t1-t0):.3f} ms, GPU busy (union of kernels) {1e-6*busy:.3f} ms, idle {1e-6*(t1-t0-busy):.3f} ms, kernels {len(seg)}")
marks = [("logmel", "front end"), ("conv1_moments", "encoder fwd (q then k)"), ("l2norm_fwd", "moco + heads"), ("maxmean_bwd", "encoder bwd"), ("sgd", "sgd")]
last = None
for s, e, n in seg:
    for key, label in marks:
        if key in n and (last is None or last[1] != label):
            if last: print(f"  {last[1]:28s} {1e-6*(s-last[0]):7.3f} ms")
            last = (s, label)
print(f"  {last[1]:28s} {1e-6*(t1-last[0]):7.3f} ms")
agg = collections.defaultdict(float)
for s, e, n in seg: agg[n.split("(")[0][-60:]] += e - s
for n, t in sorted(agg.items(), key=lambda kv: -kv[1])[:14]: print(f"    {1e-6*t:7.3f} ms  {n}")
