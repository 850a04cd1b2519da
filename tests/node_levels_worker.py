"""Worker of tests/test_distributed_cpu.py::test_node_levels_gloo: the once-per-node hierarchy hand-over of bench_dist.py
(rank 0 builds and exports, the others map the files and slice their rows) under a gloo world of 2 on CPU."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import amg_amd as AMG  # noqa: E402
from amg_amd import sharded as SH  # noqa: E402
from bench_dist import node_levels  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    def bcast(obj):
        box = [obj]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    build = lambda: AMG.ruge_stuben(AMG.poisson((30, 30, 28)), setup="host")  # noqa: E731
    levels, info, tail, shm = node_levels(rank, world, bcast, build)
    # (shard threshold 200 000 rows: this problem is all "tail"; lower it through the sizes directly)
    assert info["lc"] == 0 and (tail is not None) == (rank == 0)
    # now with sharded levels: patch the threshold by calling the pieces the way node_levels does
    orig = SH.num_sharded_levels
    SH.num_sharded_levels = lambda sizes, nranks, shard_min_rows=200_000: orig(sizes, nranks, 2000)
    try:
        levels, info, tail, shm = node_levels(rank, world, bcast, build)
    finally:
        SH.num_sharded_levels = orig
    assert info["lc"] >= 2 and len(levels) == info["lc"]
    ml = build()        # every rank rebuilds here only to CHECK what it mapped
    sums = []
    for l, d in enumerate(levels):
        cuts = SH.row_cuts(d["n"], world)
        r0, r1 = int(cuts[rank]), int(cuts[rank + 1])
        rp, ci, va = SH._rows(d["A"], r0, r1)
        Arp, Aci, Ava = ml.levels[l].A.csr_arrays()
        assert rp[0] == 0, (l, rp[0])
        assert np.array_equal(ci, Aci[Arp[r0]:Arp[r1]]), (l, "col", rank, ci.shape, Aci[Arp[r0]:Arp[r1]].shape)
        assert np.array_equal(va, Ava[Arp[r0]:Arp[r1]]), (l, "val", rank, float(np.abs(va - Ava[Arp[r0]:Arp[r1]]).max()))
        prp, pci, pva = SH._rows(d["P"], r0, r1)
        R = ml.levels[l].R        # CSR of P = CSC arrays of R
        assert np.array_equal(pci, R.rowval[R.colptr[r0]:R.colptr[r1]]) and np.array_equal(pva, R.nzval[R.colptr[r0]:R.colptr[r1]])
        assert d["pre"] == (1, 2, 1, 1.0) and d["nc"] == ml.levels[l].P.n
        sums.append(float(va.sum()))
    assert info["n_tail"] == ml.levels[info["lc"]].A.m if info["lc"] < len(ml.levels) else ml.final_A.m
    assert (tail is not None) == (rank == 0)
    dist.barrier()
    if rank == 0:
        import shutil
        shutil.rmtree(shm, ignore_errors=True)
        print("NODE_LEVELS_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
