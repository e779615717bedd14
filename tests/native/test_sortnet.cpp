// Host-side check of the register sorting networks (fsw_gnn_amd/csrc/sortnet.h): zero-one principle
// exhaustively for D <= 16, random keys with ties for every D <= 33, and weight-follows-key for PairNet.
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

template <int D>
int check_all() {
  if constexpr (D == 0) return 0;
  else return check<D>() + check_all<D - 1>();
}

int main() {
  int bad = check_all<33>();
  printf(bad ? "FAIL\n" : "OK\n");
  return bad != 0;
}
