"""Diagnostic (GPU box): repeat the two-rank graph-phase step of tests/test_gpu_ddp.py and print the loss trajectories."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-ssl_amd"), os.path.join(ROOT, "tests")]
os.environ["PYTHONPATH"] = os.pathsep.join(sys.path[:3] + [os.environ.get("PYTHONPATH", "")])     # spawned ranks import the test module
import numpy as np
import test_gpu_ddp as T
def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "delores_s"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    graph = (sys.argv[3] != "eager") if len(sys.argv) > 3 else True
    ref = None
    for r in range(reps):
        g0, g1 = T._run(which, steps=6, graph=graph)
        same = all(np.array_equal(g0["w"][n], g1["w"][n]) for n in g0["w"])
        l = np.array(g0["losses"])
        if ref is None: ref = l
        print(f"rep {r}: replicas_equal={same} max|dl|={np.abs(l - ref).max():.2e} losses={np.round(l, 5)}", flush=True)


if __name__ == "__main__":
    main()
