"""Where the seconds before the first cycle go: ruge_stuben(setup=...) stage times (AMG_SETUP_TIMING) and the upload /
schedule construction per level (AMGH_VERBOSE).   usage: python tools/verbose_build.py [N=256] [setup=gpu] [overlap=0|1]
(overlap=1: the schedules of level l are built on a second thread while the host splits level l+1, as bench.py does)"""
import os, sys, time
os.environ.setdefault("AMG_SETUP_TIMING", "1")
os.environ.setdefault("AMGH_VERBOSE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = sys.argv[2] if len(sys.argv) > 2 else "gpu"
AMG.ruge_stuben(AMG.poisson((12, 12, 12)), setup=mode).device()     # library load / HIP context outside the timings
print("---- timed from here", flush=True)
t0 = time.time(); A = AMG.poisson((N, N, N)); print("poisson", round(time.time() - t0, 2), flush=True)
overlap = len(sys.argv) > 3 and sys.argv[3] == "1"
print("clock %.2f" % (time.perf_counter() % 1000), flush=True)
t0 = time.time(); ml = AMG.ruge_stuben(A, setup=mode, device=0 if overlap else None); print("setup" + (" + upload + schedules (overlapped)" if overlap else ""), round(time.time() - t0, 2), "clock %.2f" % (time.perf_counter() % 1000), flush=True)
t0 = time.time(); dev = ml.device(); print("upload", round(time.time() - t0, 2), flush=True)
