#!/usr/bin/env python3
"""Smoother-pass time per level for several settings of the SWEEP-time tunables (amgh_debug_set_tunable:
gs_xcd_map, gs_block_pipe, gs_flip, gs_slots, gs_nnz_per_wg, gs_block_target, ...) on one resident hierarchy.
Tunables that are read when a schedule is built (gs_merge, gs_super, gs_bigslot, gs_block_inverse) need
tools/tunable_sweep.py instead.
usage: python tools/gs_tune.py [N=256] [name=v[,name=v...] ...]      each argument = one configuration"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
configs = sys.argv[2:] or ["gs_xcd_map=0", "gs_xcd_map=1"]
A = AMG.poisson((N, N, N)); ml = AMG.ruge_stuben(A); dev = ml.device(); lib = dev.lib
print("levels", [l.A.m for l in ml.levels])
lv = [l for l in range(len(ml.levels)) if ml.levels[l].A.m >= 256]
for cfg in configs:
    for kv in cfg.split(","):
        k, v = kv.split("=")
        lib.amgh_debug_set_tunable(k.encode(), int(v))
    ts = [dev.bench_op(l, 4, 3, 1) for l in lv]
    print(f"{cfg}: " + "  ".join(f"L{l} {t:7.3f}" for l, t in zip(lv, ts)) + f"   sum {sum(ts):7.3f} ms", flush=True)
