"""setup_s of bench.py (poisson + ruge_stuben(setup="gpu", device=0): hierarchy, upload and schedules pipelined) with the round-5
layouts switched off one by one (tunables read when the schedules are built / at amgh_finalize).   usage: python tools/setup_ab.py [N=256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = AMG.hip_lib()
AMG.ruge_stuben(AMG.poisson((16, 16, 16)), setup="gpu", device=0)
for dict_on, code_on in ((1, 1), (0, 1), (1, 0), (0, 0), (1, 1)):
    lib.amgh_debug_set_tunable(b"gs_bw_dict", dict_on); lib.amgh_debug_set_tunable(b"stream_code", code_on)
    t0 = time.perf_counter()
    A = AMG.poisson((N, N, N))
    ml = AMG.ruge_stuben(A, setup="gpu", device=0)
    t1 = time.perf_counter()
    dev = ml.device()
    t2 = time.perf_counter()
    print(f"gs_bw_dict = {dict_on} stream_code = {code_on}: setup_s {t1 - t0:.2f} (+ {t2 - t1:.3f} s for ml.device()), device bytes {dev.device_bytes() / 1e9:.2f} GB", flush=True)
    del dev, ml, A
lib.amgh_debug_set_tunable(b"gs_bw_dict", 1); lib.amgh_debug_set_tunable(b"stream_code", 1)
