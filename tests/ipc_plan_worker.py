"""Worker of tests/test_distributed_cpu.py::test_ipc_plans_*: ONE rank (= one process) of the row-sharded hierarchy
driven through libamghip's `amgh_dist_*` C ABI itself, IPC transport, plans only (device = -1: no GPU).  The
collective setup — halo needs exchanged between the processes, send / receive plans, interior ranges, collapse of the
coarse levels onto rank 0, host all-reduce and barrier — is checked against an independent numpy computation on the
global matrices.

    python ipc_plan_worker.py RANK NRANKS /shm_name [die]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import amg_amd as AMG  # noqa: E402
from amg_amd import sharded as SH  # noqa: E402


def off_range_cols(csr, r0, r1, c0, c1):
    rp, ci, _ = csr
    cols = np.asarray(ci[int(rp[r0]):int(rp[r1])], dtype=np.int64)
    return cols[(cols < c0) | (cols >= c1)]


def expected_plan(needs_by_rank, cuts, me):
    """What rank `me` must hold for a vector partitioned by `cuts` when rank p reads the global entries needs_by_rank[p]."""
    N = len(cuts) - 1
    halo = np.unique(needs_by_rank[me])
    recv = np.array([np.count_nonzero((halo >= cuts[p]) & (halo < cuts[p + 1])) for p in range(N)])
    send_idx, send = [], np.zeros(N, dtype=np.int64)
    for p in range(N):
        if p == me:
            continue
        h = np.unique(needs_by_rank[p])
        mine = h[(h >= cuts[me]) & (h < cuts[me + 1])] - cuts[me]
        send[p] = mine.size
        send_idx.append(mine)
    send_idx = np.concatenate(send_idx) if send_idx else np.zeros(0, dtype=np.int64)
    return halo, recv, send, send_idx


def interior(csr, r0, r1, c0, c1):
    """Longest middle run of local rows without an off-range column (what runs while the halo is in flight)."""
    rp, ci, _ = csr
    n = r1 - r0
    has = np.zeros(n, dtype=bool)
    for i in range(n):
        cols = np.asarray(ci[int(rp[r0 + i]):int(rp[r0 + i + 1])])
        has[i] = np.any((cols < c0) | (cols >= c1))
    i0, i1 = 0, n
    for i in range(n):
        if not has[i]:
            continue
        if i < n // 2:
            i0 = i + 1
        else:
            i1 = i
            break
    return min(i0, i1), i1


def main():
    rank, nranks, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    die = len(sys.argv) > 4 and sys.argv[4] == "die"
    A = AMG.poisson((12, 10, 9))
    ml = AMG.ruge_stuben(A, max_coarse=20)
    sizes = [l.A.m for l in ml.levels] + [ml.final_A.m]
    lc = SH.num_sharded_levels(sizes, nranks, shard_min_rows=60)
    assert lc >= 2, (sizes, lc)
    levels = SH.level_arrays(ml, lc)
    if die and rank == nranks - 1:
        # attach (so that the others get past the rendezvous), then vanish before the collective setup
        h = SH.C.c_void_p()
        lib = AMG.hip_lib()
        assert lib.amgh_dist_create_ipc(SH.C.byref(h), -1, rank, nranks, name.encode()) == 0
        os._exit(7)
    try:
        sh = SH.ShardedHierarchy(levels, sizes[lc], None, rank, nranks, -1, ("ipc", name))
    except AMG.AMGError as e:
        if die:
            assert "invalid state" in str(e), str(e)
            print(f"IPC_PLAN_RANK_{rank}_SAW_DEAD_PEER", flush=True)
            return
        raise
    assert not die
    cuts = [SH.row_cuts(d["n"], nranks) for d in levels] + [np.array([0] + [sizes[lc]] * nranks, dtype=np.int64)]
    for l in range(lc + 1):
        c = cuts[l]
        # x_l is read by A_l (and S_l) on the level's own rows and by P_{l-1} on the rows of level l-1
        needs = []
        for p in range(nranks):
            nd = []
            if l < lc:
                nd.append(off_range_cols(levels[l]["A"], int(c[p]), int(c[p + 1]), int(c[p]), int(c[p + 1])))
            if l >= 1:
                cp = cuts[l - 1]
                nd.append(off_range_cols(levels[l - 1]["P"], int(cp[p]), int(cp[p + 1]), int(c[p]), int(c[p + 1])))
            needs.append(np.concatenate(nd))
        halo, recv, send, send_idx = expected_plan(needs, c, rank)
        pi = sh.plan_info(l, 0)
        assert pi["nloc"] == c[rank + 1] - c[rank], (l, pi["nloc"])
        assert np.array_equal(pi["halo"], halo), (l, "halo")
        assert np.array_equal(pi["recv_cnt"], recv) and np.array_equal(pi["send_cnt"], send), (l, "counts")
        assert np.array_equal(pi["send_idx"], send_idx), (l, "send_idx")
        if l < lc:
            assert pi["interior"] == interior(levels[l]["A"], int(c[rank]), int(c[rank + 1]), int(c[rank]), int(c[rank + 1])), l
            # res_l is read by R_l on the rows of level l+1
            cn = cuts[l + 1]
            rneeds = [off_range_cols(levels[l]["R"], int(cn[p]), int(cn[p + 1]), int(c[p]), int(c[p + 1])) for p in range(nranks)]
            halo, recv, send, send_idx = expected_plan(rneeds, c, rank)
            pr = sh.plan_info(l, 1)
            assert np.array_equal(pr["halo"], halo) and np.array_equal(pr["send_idx"], send_idx), (l, "res plan")
            assert np.array_equal(pr["recv_cnt"], recv) and np.array_equal(pr["send_cnt"], send), (l, "res counts")
    # collapse: the first collapsed level lives on rank 0 alone; every other rank receives what its P rows read from it
    pt = sh.plan_info(lc, 0)
    assert pt["nloc"] == (sizes[lc] if rank == 0 else 0)
    if rank > 0:
        assert pt["recv_cnt"][0] == pt["halo"].size > 0 and pt["send_cnt"].sum() == 0
    else:
        assert pt["halo"].size == 0 and pt["send_cnt"][1:].sum() == pt["send_idx"].size > 0
    # host collectives of the transport
    v = sh.allreduce([float(rank + 1), 10.0 * rank], "sum")
    assert v[0] == nranks * (nranks + 1) / 2 and v[1] == 10.0 * nranks * (nranks - 1) / 2
    assert sh.allreduce([float(rank)], "max")[0] == nranks - 1
    for _ in range(50):
        sh.barrier()
    # plans only: the data path says so instead of touching a GPU
    rc = sh.lib.amgh_dist_precond_apply_d(sh.h, None, None, 0)
    assert rc != 0
    sh.close()
    print(f"IPC_PLAN_RANK_{rank}_OK", flush=True)


if __name__ == "__main__":
    main()
