"""Diagnostic (GPU box): run a tool under rocprofv3 --kernel-trace --stats and print the per-kernel averages.
usage: python tools/kstats.py tools/conv_bench.py [pattern]"""
import csv, glob, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp(prefix="kstats", dir="/tmp")
env = dict(os.environ, TMPDIR="/tmp")
subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, sys.argv[1])],
               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"] and "at::native" not in r["Name"]:
            print(f"{r['Name'][:90]:90s} {int(r['Calls']):5d} {float(r['AverageNs']) / 1000:9.1f} us")
