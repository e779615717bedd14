/* CPU oracle in C for the FSW_embedding forward hot path  --  TEST INFRASTRUCTURE ONLY.
 *
 * Same algorithm as oracle/fsw_oracle.py (which is pinned to golden vectors captured from the reference),
 * written with plain loops so it finishes BASELINE-sized inputs in seconds per slice block.  It is called
 * only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, through ctypes, and only as the
 * checker / the reported CPU baseline ("port").  The product path never links or loads it.
 *
 * Reference lines restated (paths relative to /root/reference/):
 *   fsw_embedding.py:778-829   mass, pad deficit max(tau - m, 0) as one extra element at x = 0, normalisation
 *   fsw_embedding.py:909-913   projection Xp = X . projVecs^T
 *   fsw_embedding.py:917-932, 1016-1025  per-slice ascending sort of each neighbourhood, weights follow
 *   fsw_embedding.py:1031-1032 inclusive cumulative weights (the segmented cumsum)
 *   fsw_embedding.py:1047-1075 Delta_t = 2 w_t sinc(xi w_t) cos(pi xi (2 c_t - w_t)), sinc(z) = sin(pi z)/(pi z)
 *   fsw_embedding.py:1084-1109 out = (1 + xi) * sum_t Delta_t p_(t)
 * Inputs are float32 (what the GPU path receives); all arithmetic is float64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  double key;
  double w;
} pair_t;

static int cmp_pair(const void* a, const void* b) {
  const double x = ((const pair_t*)a)->key, y = ((const pair_t*)b)->key;
  return (x > y) - (x < y);
}

static void sort_pairs(pair_t* p, int64_t n) {
  if (n <= 48) { /* insertion sort: the common short neighbourhoods */
    for (int64_t i = 1; i < n; ++i) {
      pair_t v = p[i];
      int64_t j = i - 1;
      while (j >= 0 && p[j].key > v.key) {
        p[j + 1] = p[j];
        --j;
      }
      p[j + 1] = v;
    }
  } else {
    qsort(p, (size_t)n, sizeof(pair_t), cmp_pair);
  }
}

static double sinc_pi(double z) { return z == 0.0 ? 1.0 : sin(M_PI * z) / (M_PI * z); }

/* out[r, k - s0] for the rows listed in `rows` (or all rows when rows == NULL) and slices s0 <= k < s1.
 * rowptr/col: CSR of adj[recipient, sender] (int64); w == NULL means unit weights.
 * mass_out (nullable): total mass of each evaluated row.  Returns 0, or 1 on allocation failure.        */
int fsw_oracle_embed(const float* X, int64_t n, int d, const int64_t* rowptr, const int64_t* col, const float* w,
                     int64_t nrows, const float* V, const float* freqs, int s0, int s1, double tau, const int64_t* rows,
                     int64_t nsel, double* out, double* mass_out, int nthreads) {
  const int S = s1 - s0;
  const int64_t R = rows ? nsel : nrows;
  if (S <= 0 || R <= 0) return 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
  /* projection of every point onto the requested slices (fsw_embedding.py:909-913) */
  double* Xp = (double*)malloc(sizeof(double) * (size_t)n * (size_t)S);
  if (!Xp) return 1;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float* x = X + i * d;
    for (int k = 0; k < S; ++k) {
      const float* v = V + (int64_t)(s0 + k) * d;
      double acc = 0.0;
      for (int c = 0; c < d; ++c) acc += (double)x[c] * (double)v[c];
      Xp[i * S + k] = acc;
    }
  }
  int failed = 0;
#pragma omp parallel
  {
    int64_t cap = 64;
    pair_t* buf = (pair_t*)malloc(sizeof(pair_t) * (size_t)cap);
#pragma omp for schedule(dynamic, 64)
    for (int64_t q = 0; q < R; ++q) {
      const int64_t r = rows ? rows[q] : q;
      const int64_t a = rowptr[r], b = rowptr[r + 1], D = b - a;
      if (D + 1 > cap) {
        cap = 2 * (D + 1);
        free(buf);
        buf = (pair_t*)malloc(sizeof(pair_t) * (size_t)cap);
      }
      if (!buf) {
        failed = 1;
        continue;
      }
      double m = 0.0;
      for (int64_t e = a; e < b; ++e) m += w ? (double)w[e] : 1.0;
      const double deficit = tau - m > 0.0 ? tau - m : 0.0;
      const double denom = m > tau ? m : tau;
      const int64_t Dt = D + (deficit > 0.0 ? 1 : 0); /* a zero-weight pad element contributes nothing */
      if (mass_out) mass_out[q] = m;
      for (int k = 0; k < S; ++k) {
        for (int64_t e = 0; e < D; ++e) {
          buf[e].key = Xp[col[a + e] * S + k];
          buf[e].w = (w ? (double)w[a + e] : 1.0) / denom;
        }
        if (Dt > D) {
          buf[D].key = 0.0;
          buf[D].w = deficit / denom;
        }
        sort_pairs(buf, Dt);
        const double xi = (double)freqs[s0 + k];
        double c = 0.0, acc = 0.0;
        for (int64_t t = 0; t < Dt; ++t) {
          const double wt = buf[t].w;
          c += wt;
          acc += 2.0 * wt * sinc_pi(xi * wt) * cos(M_PI * xi * (2.0 * c - wt)) * buf[t].key;
        }
        out[q * S + k] = (1.0 + xi) * acc;
      }
    }
    free(buf);
  }
  free(Xp);
  return failed;
}

/* segcumsum_slow (fsw_embedding.py:3016-3027) */
void fsw_oracle_segcumsum_f64(const double* x, const int64_t* ids, int64_t n, double* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = (i > 0 && ids[i] == ids[i - 1]) ? out[i - 1] + x[i] : x[i];
}

int fsw_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
