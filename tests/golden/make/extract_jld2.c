/* extract_jld2.c — dump the sparse matrix / dense arrays held in the
 * reference's JLD2 (HDF5) test fixtures as plain text, so that
 * make_fixtures.py can store them as .npz DATA fixtures.
 *
 *   gcc extract_jld2.c -I/opt/conda/include -L/opt/conda/lib -lhdf5 -o extract_jld2
 *   ./extract_jld2 file.jld2 sparse A     -> m n nnz / colptr / rowval / nzval (1-based as stored)
 *   ./extract_jld2 file.jld2 dense  B     -> ndims dims / values (HDF5 row-major = Julia column-major reversed)
 */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { long long m, n; hobj_ref_t colptr, rowval, nzval; } spref_t;

static void dump_i64(hid_t file, hobj_ref_t* ref) {
  hid_t d = H5Rdereference2(file, H5P_DEFAULT, H5R_OBJECT, ref);
  hid_t s = H5Dget_space(d);
  hssize_t n = H5Sget_simple_extent_npoints(s);
  long long* buf = malloc(sizeof(long long) * (n ? n : 1));
  H5Dread(d, H5T_NATIVE_LLONG, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf);
  printf("%lld\n", (long long)n);
  for (hssize_t i = 0; i < n; ++i) printf("%lld ", buf[i]);
  printf("\n");
  free(buf); H5Sclose(s); H5Dclose(d);
}
static void dump_f64(hid_t d) {
  hid_t s = H5Dget_space(d);
  hssize_t n = H5Sget_simple_extent_npoints(s);
  double* buf = malloc(sizeof(double) * (n ? n : 1));
  H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf);
  printf("%lld\n", (long long)n);
  for (hssize_t i = 0; i < n; ++i) printf("%.17g ", buf[i]);
  printf("\n");
  free(buf); H5Sclose(s);
}

int main(int argc, char** argv) {
  if (argc != 4) return 2;
  hid_t f = H5Fopen(argv[1], H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return 3;
  hid_t d = H5Dopen2(f, argv[3], H5P_DEFAULT);
  if (d < 0) return 4;
  if (!strcmp(argv[2], "sparse")) {
    hid_t t = H5Tcreate(H5T_COMPOUND, sizeof(spref_t));
    H5Tinsert(t, "m", HOFFSET(spref_t, m), H5T_NATIVE_LLONG);
    H5Tinsert(t, "n", HOFFSET(spref_t, n), H5T_NATIVE_LLONG);
    H5Tinsert(t, "colptr", HOFFSET(spref_t, colptr), H5T_STD_REF_OBJ);
    H5Tinsert(t, "rowval", HOFFSET(spref_t, rowval), H5T_STD_REF_OBJ);
    H5Tinsert(t, "nzval", HOFFSET(spref_t, nzval), H5T_STD_REF_OBJ);
    spref_t v;
    if (H5Dread(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, &v) < 0) return 5;
    printf("%lld %lld\n", v.m, v.n);
    dump_i64(f, &v.colptr);
    dump_i64(f, &v.rowval);
    hid_t nz = H5Rdereference2(f, H5P_DEFAULT, H5R_OBJECT, &v.nzval);
    dump_f64(nz);
    H5Dclose(nz);
  } else {
    hid_t s = H5Dget_space(d);
    int nd = H5Sget_simple_extent_ndims(s);
    hsize_t dims[8];
    H5Sget_simple_extent_dims(s, dims, NULL);
    printf("%d", nd);
    for (int i = 0; i < nd; ++i) printf(" %llu", (unsigned long long)dims[i]);
    printf("\n");
    H5Sclose(s);
    dump_f64(d);
  }
  H5Dclose(d); H5Fclose(f);
  return 0;
}
