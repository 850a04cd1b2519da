"""Gather locality of the level-ordered Gauss-Seidel sweeps under different orders of the rows INSIDE a dependency
level (any such order is exact).  Metric: distinct 64-B sectors / 128-B lines touched by the x gathers of 64
consecutive matrix entries (= one wave of the slot kernels), level-ordered matrix of levels 0-2 of ruge_stuben(poisson(N^3)).

    gcc -O2 -shared -fPIC -o tools/order_probe.so tools/order_probe.c
    python tools/order_probe.py [N]          (CPU only; N = 128 by default)
"""
import os
HERE = os.path.dirname(os.path.abspath(__file__))
import numpy as np, ctypes as C, sys, time
sys.path.insert(0, os.path.dirname(HERE))
import amg_amd as AMG
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ml = AMG.ruge_stuben(AMG.poisson((N, N, N)))
L=C.CDLL(os.path.join(HERE, 'order_probe.so'))
L.sectors.restype=C.c_double
vp=C.c_void_p
L.dep_levels.argtypes=[C.c_int,vp,vp,vp]
L.build_perm.argtypes=[C.c_int,vp,vp,vp,C.c_int,C.c_int,vp,vp]
L.sectors.argtypes=[C.c_int,vp,vp,vp,vp,vp,C.c_int,vp]
for lvl in (0,1,2):
    rp, ci, _ = ml.levels[lvl].A.csr_arrays(); rp = np.ascontiguousarray(rp, dtype=np.int32); ci = np.ascontiguousarray(ci, dtype=np.int32); n = len(rp) - 1
    lev=np.zeros(n,dtype=np.int32)
    nlev=L.dep_levels(n,rp.ctypes.data,ci.ctypes.data,lev.ctypes.data)
    print(f"level {lvl}: n={n} nnz={len(ci)} dep levels={nlev}")
    for mode,name in ((0,'ascending id (shipped)'),(1,'min lower nbr pos'),(2,'mean lower nbr pos'),(3,'mean relative pos')):
        perm=np.zeros(n,dtype=np.int32); inv=np.zeros(n,dtype=np.int32)
        L.build_perm(n,rp.ctypes.data,ci.ctypes.data,lev.ctypes.data,nlev,mode,perm.ctypes.data,inv.ctypes.data)
        out=[]
        for tri in (0,1,2):
            l128=C.c_double(0)
            s=L.sectors(n,rp.ctypes.data,ci.ctypes.data,lev.ctypes.data,perm.ctypes.data,inv.ctypes.data,tri,C.byref(l128))
            out.append((s,l128.value))
        print(f"   {name:26s} sectors/64 entries: all {out[0][0]:.1f} lower {out[1][0]:.1f} upper {out[2][0]:.1f} | 128B lines: all {out[0][1]:.1f} lower {out[1][1]:.1f} upper {out[2][1]:.1f}")
