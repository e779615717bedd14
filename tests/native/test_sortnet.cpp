// Host-side check of the register sorting networks (fsw_gnn_amd/csrc/sortnet.h): zero-one principle
// exhaustively for D <= 16, random keys with ties for every D <= 33, weight-follows-key for PairNet, and the stable
// (key, then element index) order of IndexedNet / U64Net used by the backward kernels, also at the padded sizes 40..129.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../fsw_gnn_amd/csrc/sortnet.h"

using namespace fsw;

template <int D>
int check() {
  int bad = 0;
  if (D <= 16) {
    for (unsigned m = 0; m < (1u << D); ++m) {
      KeyNet<D> n;
      for (int i = 0; i < D; ++i) n.k[i] = (m >> i) & 1 ? 1.f : 0.f;
      sort_network<D>(n);
      for (int i = 1; i < D; ++i) bad += n.k[i - 1] > n.k[i];
    }
  }
  for (int rep = 0; rep < 2000; ++rep) {
    KeyNet<D> n;
    PairNet<D> p;
    std::vector<std::pair<float, float>> ref(D);
    for (int i = 0; i < D; ++i) {
      float k = (float)(rand() % (rep % 2 ? 7 : 100000)) - 50.f;
      n.k[i] = k;
      p.k[i] = k;
      p.w[i] = k * 3.f + 1.f;  // weight is a function of the key: any stable or unstable order must keep pairs intact
      ref[i] = {k, p.w[i]};
    }
    sort_network<D>(n);
    sort_network<D>(p);
    std::sort(ref.begin(), ref.end());
    for (int i = 0; i < D; ++i) bad += (n.k[i] != ref[i].first) + (p.k[i] != ref[i].first) + (p.w[i] != ref[i].second);
  }
  if (bad) printf("D=%d: %d errors\n", D, bad);
  return bad;
}

// (key, index) networks: the result must be THE stable sort by key (ties broken by index), with negative keys, zeros of
// both signs and infinities among the keys
template <int D>
int check_indexed() {
  int bad = 0;
  for (int rep = 0; rep < 300; ++rep) {
    IndexedNet<D> a;
    U64Net<D> b;
    std::vector<std::pair<float, int>> ref(D);
    for (int i = 0; i < D; ++i) {
      float k = (float)(rand() % (rep % 2 ? 5 : 100000)) - 2.f;
      if (rep % 7 == 0 && i % 5 == 0) k = __builtin_inff();
      if (rep % 11 == 0 && i % 3 == 0) k = -k * 0.5f;
      a.k[i] = k;
      __builtin_memcpy(&a.w[i], &i, 4);
      b.e[i] = pack_key_index(k, i);
      ref[i] = {k, i};
    }
    sort_network<D>(a);
    sort_network<D>(b);
    std::stable_sort(ref.begin(), ref.end(), [](const std::pair<float, int>& x, const std::pair<float, int>& y) { return x.first < y.first; });
    for (int i = 0; i < D; ++i) {
      int ia;
      __builtin_memcpy(&ia, &a.w[i], 4);
      const int ea = (a.k[i] != ref[i].first) + (ia != ref[i].second);
      const int eb = (from_orderable_bits((unsigned int)(b.e[i] >> 32)) != ref[i].first) + ((int)(unsigned int)b.e[i] != ref[i].second);
      bad += ea + eb;
    }
  }
  if (bad) printf("indexed D=%d: %d errors\n", D, bad);
  return bad;
}

template <int D>
int check_all() {
  if constexpr (D == 0) return 0;
  else return check<D>() + check_all<D - 1>();
}

int main() {
  int bad = check_all<33>();
  bad += check_indexed<1>() + check_indexed<2>() + check_indexed<7>() + check_indexed<8>() + check_indexed<16>() + check_indexed<32>() +
         check_indexed<33>() + check_indexed<40>() + check_indexed<49>() + check_indexed<64>() + check_indexed<97>() + check_indexed<129>();
  printf(bad ? "FAIL\n" : "OK\n");
  return bad != 0;
}
