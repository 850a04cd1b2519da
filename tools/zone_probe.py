#!/usr/bin/env python3
"""Per-LAUNCH durations of one symmetric Gauss-Seidel pass on levels 0..2 of the 256^3 hierarchy, for group sizes
capped at 2, 3, 4, 5 (uniform m per level): the input for deciding whether groups of DIFFERENT depth along one sweep
(deep where dependency levels are small, shallow where they are large) would pay.
Run under rocprofv3 --kernel-trace; this script then reads the trace CSV and writes a compact JSON.
usage: rocprofv3 --kernel-trace -d DIR -o zp -- python tools/zone_probe.py run [N=256]
       python tools/zone_probe.py parse DIR/.../zp_kernel_trace.csv OUT.json"""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CAPS = (2, 3, 4, 5)
LEVELS = (0, 1, 2)


def run(N):
    import ctypes as C
    import numpy as np
    import amg_amd as AMG
    from amg_amd.device import DeviceHierarchy
    A = AMG.poisson((N, N, N)); ml = AMG.ruge_stuben(A, setup="gpu")
    lib = AMG.hip_lib()
    mark_x = AMG.DeviceBuffer(4096, 0, np.ones(4096)); sc = AMG.DeviceBuffer(2048, 0); out = C.c_double(0)

    def marker(k):
        for _ in range(k):
            lib.amgh_dot_d(0, 4096, mark_x.ptr, mark_x.ptr, sc.ptr, C.byref(out), None)

    meta = []
    for cap in CAPS:
        lib.amgh_debug_set_tunable(b"gs_merge_force", cap)
        dev = DeviceHierarchy(ml)
        for l in LEVELS:
            dev.bench_op(l, 4, 1, 1)               # warm: schedules grown, graphs not involved
            marker(3)
            dev.bench_op(l, 4, 1, 0)
            marker(2)
            meta.append(dict(cap=cap, level=l, fwd=dev.gs_sweep_stats(l, False), bwd=dev.gs_sweep_stats(l, True),
                             nlev=dev.gs_dependency_levels(l)))
        del dev                                     # one hierarchy resident at a time
        import gc; gc.collect()
    json.dump(meta, open(os.environ.get("ZP_META", "gpurun_out/zp_meta.json"), "w"))


def parse(path, outp):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    print("kernels in trace:", len(rows), sorted(set(n.split("(")[0][:60] for _, _, n in rows))[:40])
    seq = [(d, n.split("(")[0].split("<")[0].replace("amgh::", "")) for _, d, n in rows]
    # phases: ... [3 x dot marker] <measured pass> [2 x dot marker] ...
    phases, i = [], 0
    isdot = lambda k: "dot_partial" in seq[k][1]
    while i < len(seq):
        if isdot(i):
            j = i
            while j < len(seq) and (isdot(j) or "reduce_final" in seq[j][1] or "copyBuffer" in seq[j][1]):
                j += 1
            ndot = sum(1 for k in range(i, j) if isdot(k))
            if ndot == 3:      # a measured pass follows
                k = j
                while k < len(seq) and not isdot(k):
                    k += 1
                phases.append([(d, n) for d, n in seq[j:k]])
                i = k
                continue
            i = j
        else:
            i += 1
    json.dump(phases, open(outp, "w"))
    print(len(phases), "measured passes;", [len(p) for p in phases])


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
    else:
        parse(sys.argv[2], sys.argv[3])
