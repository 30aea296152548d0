// hull.h — exact integer convex hull from per-row extremes (Andrew's monotone chain).
#pragma once
#include "common.h"

struct P2 { int r, c; };

__device__ __forceinline__ long long cross3(P2 o, P2 a, P2 b) {
  return (long long)(a.r - o.r) * (b.c - o.c) - (long long)(a.c - o.c) * (b.r - o.r);
}

// monotone chain over rows [0,nrows): forward pass builds the "lower" chain, backward the "upper";
// points of a row are (row, lo[row]) then (row, hi[row]); rows with lo>hi are empty.
static __device__ int chain_build(const int* lo, const int* hi, int nrows, bool forward, P2* st) {
  int n = 0;
  auto push = [&](P2 p) {
    while (n >= 2 && cross3(st[n - 2], st[n - 1], p) <= 0) --n;
    st[n++] = p;
  };
  if (forward) {
    for (int r = 0; r < nrows; ++r) {
      if (lo[r] > hi[r]) continue;
      push(P2{r, lo[r]});
      if (hi[r] != lo[r]) push(P2{r, hi[r]});
    }
  } else {
    for (int r = nrows - 1; r >= 0; --r) {
      if (lo[r] > hi[r]) continue;
      push(P2{r, hi[r]});
      if (hi[r] != lo[r]) push(P2{r, lo[r]});
    }
  }
  return n;
}

