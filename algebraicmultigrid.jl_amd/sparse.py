"""SparseMatrixCSC — the host-side sparse matrix, held by libamgsetup.

Mirrors Julia's `SparseMatrixCSC` (colptr / rowval / nzval), 0-based int32/f64.
The arrays are zero-copy numpy views of the C++ object.
"""
import ctypes as C
import threading

import numpy as np

from ._libs import AMGError, setup_lib


# the lazily cached transpose / symmetry flag of a matrix may be asked for by several host threads at once (the ranks
# of a single-process sharded run): create them once — a second wrapper replacing the first would free the arrays
# the first caller still holds views of
_cache_lock = threading.Lock()


def _view(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    ctype = C.c_int32 if dtype == np.int32 else C.c_double
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,))
    return arr


class SparseMatrixCSC:
    """Owns an `amgs_mat*`.  `colptr`, `rowval`, `nzval` are borrowed views."""

    def __init__(self, handle, owner=None):
        if not handle:
            raise AMGError("libamgsetup: " + setup_lib().amgs_last_error().decode())
        self._h = handle
        self._owner = owner  # keeps a hierarchy alive for borrowed matrices
        L = setup_lib()
        self.m = int(L.amgs_mat_rows(handle))
        self.n = int(L.amgs_mat_cols(handle))
        self._nnz = int(L.amgs_mat_nnz(handle))
        self.colptr = _view(L.amgs_mat_colptr(handle), self.n + 1, np.int32)
        self.rowval = _view(L.amgs_mat_rowval(handle), self._nnz, np.int32)
        self.nzval = _view(L.amgs_mat_nzval(handle), self._nnz, np.float64)
        self._sym = None
        self.eltype = np.dtype(np.float64)  # element type the CALLER handed in (storage/arithmetic is always f64)
        self._T = None

    def __del__(self):
        try:
            if self._owner is None and self._h:
                setup_lib().amgs_mat_free(self._h)
        except Exception:
            pass

    # ---- construction -----------------------------------------------------
    @classmethod
    def from_arrays(cls, m, n, colptr, rowval, nzval):
        # indices are int32 on both libraries: refuse what does not fit BEFORE the cast wraps it around
        i32max = np.iinfo(np.int32).max
        cp = np.asarray(colptr)
        if max(int(m), int(n)) >= i32max or (cp.size and int(cp[-1]) > i32max):
            raise AMGError("matrix too large for int32 indices (rows, columns or stored entries >= 2^31)")
        colptr = np.ascontiguousarray(colptr, dtype=np.int32)
        rowval = np.ascontiguousarray(rowval, dtype=np.int32)
        nzval = np.ascontiguousarray(nzval, dtype=np.float64)
        h = setup_lib().amgs_mat_create(m, n, colptr.ctypes.data, rowval.ctypes.data, nzval.ctypes.data)
        return cls(h)

    @classmethod
    def from_scipy(cls, A):
        import scipy.sparse as sp
        if np.iscomplexobj(A.data if hasattr(A, "data") else A):
            raise AMGError("complex matrices are not supported by the HIP path")
        A = sp.csc_matrix(A)
        A.sum_duplicates()
        A.sort_indices()
        out = cls.from_arrays(A.shape[0], A.shape[1], A.indptr, A.indices, A.data)
        if A.dtype == np.float32:
            out.eltype = np.dtype(np.float32)
        return out

    @classmethod
    def from_dense(cls, M):
        import scipy.sparse as sp
        return cls.from_scipy(sp.csc_matrix(np.asarray(M, dtype=np.float64)))

    @classmethod
    def coerce(cls, A):
        if isinstance(A, cls):
            return A
        if isinstance(A, np.ndarray):
            return cls.from_dense(A)
        return cls.from_scipy(A)

    # ---- views ---------------------------------------------------------------
    @property
    def shape(self):
        return (self.m, self.n)

    @property
    def nnz(self):
        return self._nnz

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csc_matrix((self.nzval.copy(), self.rowval.copy(), self.colptr.copy()), shape=self.shape)

    def toarray(self):
        return self.to_scipy().toarray()

    def transpose(self):
        """copy(A')"""
        if self._T is None:
            with _cache_lock:
                if self._T is None:
                    self._T = SparseMatrixCSC(setup_lib().amgs_mat_transpose(self._h))
        return self._T

    @property
    def T(self):
        return self.transpose()

    def is_symmetric(self):
        """True when the CSC arrays equal those of A' (so they are also A's CSR)."""
        if self._sym is None:
            with _cache_lock:
                if self._sym is None:
                    self._sym = bool(setup_lib().amgs_mat_is_symmetric(self._h))
        return self._sym

    def csr_arrays(self):
        """(rowptr, col, val) of this matrix in CSR = CSC arrays of the transpose."""
        T = self if self.is_symmetric() else self.transpose()
        return T.colptr, T.rowval, T.nzval

    def __matmul__(self, x):
        if isinstance(x, SparseMatrixCSC):
            return SparseMatrixCSC(setup_lib().amgs_mat_spgemm(self._h, x._h))
        return self.to_scipy() @ np.asarray(x)

    __mul__ = __matmul__

    def diagonal(self):
        return self.to_scipy().diagonal()

    def __repr__(self):
        return f"{self.m}x{self.n} SparseMatrixCSC{{Float64,Int32}} with {self.nnz} stored entries"
