// Software pipeline over the rows of one degree bin, shared by the unit-weight register-path kernels.  gfx950.
//
// These kernels are bound by memory-level parallelism: a wave that issues the D gathers of a row, waits, sorts and
// only then turns to the next row leaves the memory pipe idle while it sorts.  Here a wave keeps the gathers of
// P consecutive rows in flight: at step r it issues the gathers of row r+P-1, fetches the col indices of row r+P
// (wave-uniform scalar loads, one step ahead of their use) and then finishes row r.  The loop body is branch-free
// (rows past the end are clamped to the last row and their result is dropped), so the wait the compiler inserts
// before row r's sort is s_waitcnt vmcnt((P-1)*D) and not vmcnt(0).
#pragma once
#include "fsw_common.h"
#include "sortnet.h"

namespace fsw {

#ifndef FSW_PIPE_D4
#define FSW_PIPE_D4 0     // degrees <= this: 4 rows in flight
#endif
#ifndef FSW_PIPE_D3
#define FSW_PIPE_D3 0     // degrees <= this: 3 rows in flight
#endif
#ifndef FSW_PIPE_D2
#define FSW_PIPE_D2 22    // degrees <= this: 2 rows in flight; above: 1 (register budget)
#endif

template <int D>
constexpr int pipeline_depth() {
  return D <= FSW_PIPE_D4 ? 4 : D <= FSW_PIPE_D3 ? 3 : D <= FSW_PIPE_D2 ? 2 : 1;
}

template <int D>
struct ColIdx {
  int c[D];
};

// start_of(r): wave-uniform CSR offset of the block's row r (0 <= r < nrows)
// finish(net, r): consume the gathered, still unsorted keys of row r.  It must not branch (store unconditionally;
// lanes past the last slice recompute slice S-1 and may store that same value again): a branch splits the loop body
// and lets the compiler sink the next rows' gathers below it.
// BAR: scheduling barriers around a row's sort (bit 0 before, bit 1 after); measured per kernel, see the call sites
template <int D, int P, int BAR, class StartFn, class FinishFn>
__device__ __forceinline__ void pipelined_rows(int nrows, const int32_t* __restrict__ col, const float* __restrict__ xk, int64_t ldp,
                                               StartFn start_of, FinishFn finish) {
  auto load_cols = [&](int r, ColIdx<D>& cs) {
    const int start = start_of(min(r, nrows - 1));
#pragma unroll
    for (int t = 0; t < D; ++t) cs.c[t] = col[start + t];
  };
  auto gather = [&](const ColIdx<D>& cs, KeyNet<D>& net) {
#pragma unroll
    for (int t = 0; t < D; ++t) net.k[t] = xk[(int64_t)cs.c[t] * ldp];
  };
  if constexpr (P == 1) {
    for (int r = 0; r < nrows; ++r) {
      ColIdx<D> cs;
      KeyNet<D> net;
      load_cols(r, cs);
      gather(cs, net);
      finish(net, r);
    }
  } else {
    KeyNet<D> n[P];
    ColIdx<D> cs;
#pragma unroll
    for (int i = 0; i < P - 1; ++i) {
      load_cols(i, cs);
      gather(cs, n[i]);
    }
    load_cols(P - 1, cs);
    int r = 0;
    for (; r + P <= nrows; r += P) {   // at the top: n[0..P-2] in flight for rows r..r+P-2, cs = col indices of row r+P-1
#pragma unroll
      for (int i = 0; i < P; ++i) {
        gather(cs, n[(i + P - 1) % P]);
        load_cols(r + i + P, cs);
        if constexpr (BAR & 1) __builtin_amdgcn_sched_barrier(0);   // pin the gathers above the sort
        finish(n[i], r + i);
        if constexpr (BAR & 2) __builtin_amdgcn_sched_barrier(0);   // pin the next step's address arithmetic below it
      }
    }
#pragma unroll
    for (int i = 0; i < P - 1; ++i)
      if (r + i < nrows) finish(n[i], r + i);
  }
}

}  // namespace fsw
