#!/usr/bin/env python3
"""Smoother-pass time per level for several values of a SWEEP-time tunable (amgh_debug_set_tunable: gs_xcd_map,
gs_block_pipe, gs_flip, gs_slots, ...) on one resident hierarchy.  Tunables that are read when a schedule is
built (gs_merge, gs_super, gs_bigslot, gs_block_inverse) need tools/tunable_sweep.py instead.
usage: python tools/gs_tune.py [N=256] [name v1 v2 ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
name = (sys.argv[2] if len(sys.argv) > 2 else "gs_xcd_map").encode()
values = [int(v) for v in sys.argv[3:]] or [0, 1, 0, 1]
A = AMG.poisson((N, N, N)); ml = AMG.ruge_stuben(A); dev = ml.device(); lib = dev.lib
print("levels", [l.A.m for l in ml.levels])
lv = [l for l in range(len(ml.levels)) if ml.levels[l].A.m >= 256]
for v in values:
    lib.amgh_debug_set_tunable(name, v)
    ts = [dev.bench_op(l, 4, 3, 1) for l in lv]
    print(f"{name.decode()}={v}: " + "  ".join(f"L{l} {t:7.3f}" for l, t in zip(lv, ts)) + f"   sum {sum(ts):7.3f} ms", flush=True)
