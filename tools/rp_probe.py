"""Gather locality of restriction and prolongation: natural order vs the level-ordered cycle's numbering (rows of R in the
coarse level's dependency-level order, its columns in the fine level's; P the other way round).  Metric: distinct 64-B
sectors touched by the gathers of 64 consecutive matrix entries.  CPU only.

    gcc -O2 -shared -fPIC -o tools/order_probe.so tools/order_probe.c ; python tools/rp_probe.py [N=128]
"""
import numpy as np, ctypes as C, sys, os
HERE=os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0,os.path.dirname(HERE))
import amg_amd as AMG
L=C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)),'order_probe.so')); vp=C.c_void_p
L.dep_levels.argtypes=[C.c_int,vp,vp,vp]; L.build_perm.argtypes=[C.c_int,vp,vp,vp,C.c_int,C.c_int,vp,vp]
N=int(sys.argv[1]) if len(sys.argv)>1 else 128
ml=AMG.ruge_stuben(AMG.poisson((N,N,N)))
def order(A):
    rp,ci,_=A.csr_arrays(); rp=np.ascontiguousarray(rp,dtype=np.int32); ci=np.ascontiguousarray(ci,dtype=np.int32); n=len(rp)-1
    lev=np.zeros(n,dtype=np.int32); nlev=L.dep_levels(n,rp.ctypes.data,ci.ctypes.data,lev.ctypes.data)
    perm=np.zeros(n,dtype=np.int32); inv=np.zeros(n,dtype=np.int32)
    L.build_perm(n,rp.ctypes.data,ci.ctypes.data,lev.ctypes.data,nlev,0,perm.ctypes.data,inv.ctypes.data)
    return perm,inv,lev
def sectors(rowptr,col,row_perm,col_inv):
    # entries in row order row_perm, columns mapped by col_inv; distinct 64B sectors per 64 consecutive entries
    rowptr=np.asarray(rowptr,dtype=np.int64); col=np.asarray(col)
    lens=(rowptr[1:]-rowptr[:-1])[row_perm]
    starts=rowptr[:-1][row_perm]
    idx=np.repeat(starts-np.concatenate(([0],np.cumsum(lens)[:-1])),lens)+np.arange(lens.sum())
    c=col_inv[col[idx]]
    m=(len(c)//64)*64
    s=(c[:m]>>3).reshape(-1,64); s.sort(axis=1)
    d=1+(np.diff(s,axis=1)!=0).sum(axis=1)
    return d.mean()
for l in (0,1):
    pf,invf,levf=order(ml.levels[l].A)
    Anext=ml.levels[l+1].A if l+1<len(ml.levels) else ml.final_A
    pc,invc,levc=order(Anext)
    lev=ml.levels[l]
    Rr=(lev.P.colptr,lev.P.rowval)   # CSR of R: rows coarse
    Pr=(lev.R.colptr,lev.R.rowval)   # CSR of P: rows fine
    n,nc=lev.A.m,lev.P.n
    idf=np.arange(n,dtype=np.int32); idc=np.arange(nc,dtype=np.int32)
    print(f"level {l}: R natural {sectors(*Rr,idc,idf):.1f}  R level-ordered(coarse lo x fine lo) {sectors(*Rr,pc,invf):.1f}  R (coarse natural x fine lo) {sectors(*Rr,idc,invf):.1f} sectors/64 entries; nnz/row {len(Rr[1])/nc:.1f}")
    print(f"         P natural {sectors(*Pr,idf,idc):.1f}  P level-ordered {sectors(*Pr,pf,invc):.1f}  P (fine lo x coarse natural) {sectors(*Pr,pf,idc):.1f}; nnz/row {len(Pr[1])/n:.1f}")
