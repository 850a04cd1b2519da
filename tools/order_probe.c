// order_probe: gather locality of the level-ordered sweep under different orders of the rows INSIDE a dependency level
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
typedef struct { double key; int32_t id; } kv;
static int cmp_kv(const void* a, const void* b) { const kv* x = a; const kv* y = b; if (x->key < y->key) return -1; if (x->key > y->key) return 1; return x->id - y->id; }
// lev: out level per row; returns nlev
int dep_levels(int n, const int32_t* rp, const int32_t* ci, int32_t* lev) {
  int maxl = -1; memset(lev, 0, 4 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    int li = lev[i];
    for (int j = rp[i]; j < rp[i + 1]; ++j) { int c = ci[j]; if (c < i && lev[c] + 1 > li) li = lev[c] + 1; }
    lev[i] = li;
    for (int j = rp[i]; j < rp[i + 1]; ++j) { int c = ci[j]; if (c > i && c < n && lev[c] < li + 1) lev[c] = li + 1; }
    if (li > maxl) maxl = li;
  }
  return maxl + 1;
}
// mode 0: ascending id; 1: by min lower-neighbour position; 2: by mean lower-neighbour position; 3: mean of ALL already placed... 
void build_perm(int n, const int32_t* rp, const int32_t* ci, const int32_t* lev, int nlev, int mode, int32_t* perm, int32_t* inv) {
  int32_t* lp = calloc(nlev + 1, 4);
  for (int i = 0; i < n; ++i) lp[lev[i] + 1]++;
  for (int l = 0; l < nlev; ++l) lp[l + 1] += lp[l];
  int32_t* nx = malloc(4 * (size_t)nlev); memcpy(nx, lp, 4 * (size_t)nlev);
  for (int i = 0; i < n; ++i) perm[nx[lev[i]]++] = i;
  for (int p = 0; p < n; ++p) inv[perm[p]] = p;
  if (mode > 0) {
    kv* tmp = malloc(sizeof(kv) * (size_t)n);
    for (int l = 0; l < nlev; ++l) {
      int a = lp[l], b = lp[l + 1];
      for (int p = a; p < b; ++p) {
        int i = perm[p]; double key = 0; int cnt = 0; double mn = 1e300;
        for (int j = rp[i]; j < rp[i + 1]; ++j) { int c = ci[j]; if (c < n && lev[c] < l) { double q = inv[c]; if (mode == 3) q = (double)(inv[c] - lp[lev[c]]) / (double)(lp[lev[c] + 1] - lp[lev[c]]); key += q; cnt++; if (q < mn) mn = q; } }
        tmp[p - a].id = i;
        tmp[p - a].key = cnt == 0 ? (double)i * 1e-9 : (mode == 1 ? mn : key / cnt);
      }
      qsort(tmp, b - a, sizeof(kv), cmp_kv);
      for (int p = a; p < b; ++p) { perm[p] = tmp[p - a].id; inv[perm[p]] = p; }
    }
    free(tmp);
  }
  free(lp); free(nx);
}
// average distinct 64-B sectors (8 doubles) per 64 consecutive entries of the level-ordered matrix (all entries of the rows)
// tri: 0 all entries, 1 only lower (level < own), 2 only upper
double sectors(int n, const int32_t* rp, const int32_t* ci, const int32_t* lev, const int32_t* perm, const int32_t* inv, int tri, double* lines128) {
  long long tot = 0, tot128 = 0, chunks = 0; int cnt = 0; int32_t buf[64];
  for (int p = 0; p < n; ++p) {
    int i = perm[p];
    for (int j = rp[i]; j < rp[i + 1]; ++j) {
      int c = ci[j]; if (c >= n) continue;
      if (tri == 1 && !(lev[c] < lev[i])) continue;
      if (tri == 2 && !(lev[c] > lev[i])) continue;
      buf[cnt++] = inv[c];
      if (cnt == 64) {
        int d = 0, d2 = 0;
        for (int a = 0; a < 64; ++a) { int s = buf[a] >> 3, s2 = buf[a] >> 4, dup = 0, dup2 = 0; for (int b = 0; b < a; ++b) { if ((buf[b] >> 3) == s) dup = 1; if ((buf[b] >> 4) == s2) dup2 = 1; } d += !dup; d2 += !dup2; }
        tot += d; tot128 += d2; chunks++; cnt = 0;
      }
    }
  }
  *lines128 = (double)tot128 / chunks;
  return (double)tot / chunks;
}
