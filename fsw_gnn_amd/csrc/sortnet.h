// Compile-time sorting networks held entirely in registers (one neighbourhood per lane).
//
// Batcher's odd-even merge sort for the next power of two, pruned to D wires: every comparator is in
// standard form (min to the lower wire), so wires >= D behave as +inf and comparators touching them can
// be dropped.  All indices are compile-time constants after unrolling, so key[] stays in VGPRs
// (runtime-indexed arrays would go to scratch: cdna_hip_programming.md rule 20).
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FSW_HD __host__ __device__ __forceinline__
#else
#define FSW_HD inline
#endif

namespace fsw {

// keys only: v_min_f32 + v_max_f32 per comparator
template <int D>
struct KeyNet {
  float k[D];
  template <int I, int J>
  FSW_HD void cx() {
    if constexpr (J < D) {
      const float lo = fminf(k[I], k[J]);
      const float hi = fmaxf(k[I], k[J]);
      k[I] = lo;
      k[J] = hi;
    }
  }
};

// (key, weight) pairs: the weight follows its key
template <int D>
struct PairNet {
  float k[D];
  float w[D];
  template <int I, int J>
  FSW_HD void cx() {
    if constexpr (J < D) {
      const bool sw = k[J] < k[I];
      const float ki = k[I], kj = k[J], wi = w[I], wj = w[J];
      k[I] = sw ? kj : ki;
      k[J] = sw ? ki : kj;
      w[I] = sw ? wj : wi;
      w[J] = sw ? wi : wj;
    }
  }
};

// (key, element index) pairs ordered by key, then index: a stable sort by key.  The index travels as float bits in w[].
template <int D>
struct IndexedNet {
  float k[D];
  float w[D];
  template <int I, int J>
  FSW_HD void cx() {
    if constexpr (J < D) {
      int ii, ij;
      __builtin_memcpy(&ii, &w[I], 4);
      __builtin_memcpy(&ij, &w[J], 4);
      const bool sw = k[J] < k[I] || (k[J] == k[I] && ij < ii);
      const float ki = k[I], kj = k[J], wi = w[I], wj = w[J];
      k[I] = sw ? kj : ki;
      k[J] = sw ? ki : kj;
      w[I] = sw ? wj : wi;
      w[J] = sw ? wi : wj;
    }
  }
};

// 64-bit words (orderable key bits << 32 | element index): one unsigned compare orders by key, then index
template <int D>
struct U64Net {
  unsigned long long e[D];
  template <int I, int J>
  FSW_HD void cx() {
    if constexpr (J < D) {
      const unsigned long long a = e[I], b = e[J];
      const bool sw = b < a;
      e[I] = sw ? b : a;
      e[J] = sw ? a : b;
    }
  }
};

// float <-> uint32 with the same order (negative floats reversed, positive above them)
FSW_HD unsigned int orderable_bits(float f) {
  unsigned int u;
  __builtin_memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
FSW_HD float from_orderable_bits(unsigned int o) {
  const unsigned int u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}
FSW_HD unsigned long long pack_key_index(float key, int idx) {
  if (key == 0.f) key = 0.f;   // -0 and +0 are equal keys: both map to +0 so that only the index orders them
  return ((unsigned long long)orderable_bits(key) << 32) | (unsigned int)idx;
}

template <class Net, int LO, int N, int R>
struct OddEvenMerge {
  template <int I, int END, int STEP>
  static FSW_HD void row(Net& n) {
    if constexpr (I < END) {
      n.template cx<I, I + R>();
      row<I + STEP, END, STEP>(n);
    }
  }
  static FSW_HD void run(Net& n) {
    constexpr int M = R * 2;
    if constexpr (M < N) {
      OddEvenMerge<Net, LO, N, M>::run(n);
      OddEvenMerge<Net, LO + R, N, M>::run(n);
      row<LO + R, LO + N - R, M>(n);
    } else {
      n.template cx<LO, LO + R>();
    }
  }
};

template <class Net, int LO, int N>
struct OddEvenSort {
  static FSW_HD void run(Net& n) {
    if constexpr (N > 1) {
      constexpr int M = N / 2;
      OddEvenSort<Net, LO, M>::run(n);
      OddEvenSort<Net, LO + M, M>::run(n);
      OddEvenMerge<Net, LO, N, 1>::run(n);
    }
  }
};

constexpr int net_pow2(int d) {
  int p = 1;
  while (p < d) p <<= 1;
  return p;
}

template <int D, class Net>
FSW_HD void sort_network(Net& n) {
  OddEvenSort<Net, 0, net_pow2(D)>::run(n);
}

}  // namespace fsw
