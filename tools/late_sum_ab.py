#!/usr/bin/env python3
"""A/B of the relayed walk's row sum on the headline hierarchy (256^3, ruge_stuben defaults): stored-order sum (tunable
gs_bw_inorder = 1: the scalar loop's bits) against the dependency-aware one (0, the default: the far half of a row summed above
the hand-over, csrc/hip/gs_relay.hpp LATE) — V-cycle, the per-level smoother times of the cycle as it runs, and the stand-alone
pre-smoother of levels 0 / 1.   usage: python tools/late_sum_ab.py [N = 256] [rounds = 3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from bench import uniform

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ml = AMG.ruge_stuben(AMG.poisson((N, N, N)), setup="gpu", device=0)
dev = ml.device(0, 1)
lib = dev.lib
n = ml.levels[0].A.m
bd = AMG.DeviceBuffer(n, 0, uniform(n, 0)); zd = AMG.DeviceBuffer(n, 0)
print(f"{N}^3: levels {[l.A.m for l in ml.levels]}; late-capable levels (records split around the padding): "
      f"{[l for l in range(len(ml.levels)) if (lib.amgh_debug_set_tunable(b'gs_bw_inorder', 0), lib.amgh_debug_bw_late(dev.h, l))[1] == 1]}", flush=True)
z = {}
for r in range(rounds):
    for inorder in (1, 0):
        lib.amgh_debug_set_tunable(b"gs_bw_inorder", inorder)
        for _ in range(3): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        assert lib.amgh_dev_sync(0) == 0
        t0 = time.perf_counter()
        for _ in range(10): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        assert lib.amgh_dev_sync(0) == 0
        ms = 1e3 * (time.perf_counter() - t0) / 10
        z[inorder] = zd.download()
        dev.profile(True)
        for _ in range(3): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
        lib.amgh_dev_sync(0)
        prof = dev.profile_read(); dev.profile(False)
        sm = [(prof["Presmoother"][l] + prof["Postsmoother"][l]) / 3 for l in range(3)]
        pre = [dev.bench_op(l, 4, 20, 3) for l in (0, 1)]
        print(f"round {r} gs_bw_inorder = {inorder}: V-cycle {ms:.3f} ms; smoothers per cycle level 0 / 1 / 2: {sm[0]:.3f} / {sm[1]:.3f} / {sm[2]:.3f} ms; "
              f"stand-alone symmetric pre-smoother level 0 / 1: {pre[0]:.3f} / {pre[1]:.3f} ms", flush=True)
lib.amgh_debug_set_tunable(b"gs_bw_inorder", 0)
d = np.linalg.norm(z[0] - z[1]) / np.linalg.norm(z[1])
print(f"||z_late - z_inorder|| / ||z_inorder|| = {d:.3e}  (same iterate, one reassociation per row)")
