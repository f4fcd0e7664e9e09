// libstdc++ std::sort restated for (float radius, int index) pairs ordered by `left.r > right.r`.
// Included by k_detect.hip with KVFE_HD = __device__ (the product) and by tests/cpp/stdsort_check.cpp with KVFE_HD
// empty, where it is compared with the host's real std::sort on tie-heavy inputs.
#pragma once
struct BrownRI {
  float r;
  int i;
};
KVFE_HD inline bool brown_comp(const BrownRI& a, const BrownRI& b) { return a.r > b.r; }
KVFE_HD inline void brown_swap(BrownRI* a, int x, int y) {
  const BrownRI t = a[x];
  a[x] = a[y];
  a[y] = t;
}
// std::__adjust_heap + std::__push_heap (bits/stl_heap.h)
KVFE_HD void brown_adjust_heap(BrownRI* first, int holeIndex, int len, BrownRI value) {
  const int topIndex = holeIndex;
  int secondChild = holeIndex;
  while (secondChild < (len - 1) / 2) {
    secondChild = 2 * (secondChild + 1);
    if (brown_comp(first[secondChild], first[secondChild - 1])) secondChild--;
    first[holeIndex] = first[secondChild];
    holeIndex = secondChild;
  }
  if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
    secondChild = 2 * (secondChild + 1);
    first[holeIndex] = first[secondChild - 1];
    holeIndex = secondChild - 1;
  }
  int parent = (holeIndex - 1) / 2;
  while (holeIndex > topIndex && brown_comp(first[parent], value)) {
    first[holeIndex] = first[parent];
    holeIndex = parent;
    parent = (holeIndex - 1) / 2;
  }
  first[holeIndex] = value;
}
// std::__partial_sort(first, last, last) = __heap_select (make_heap; nothing beyond `middle`) + __sort_heap
KVFE_HD void brown_heap_sort(BrownRI* first, int len) {
  if (len >= 2) {
    int parent = (len - 2) / 2;
    for (;;) {
      const BrownRI value = first[parent];
      brown_adjust_heap(first, parent, len, value);
      if (parent == 0) break;
      parent--;
    }
  }
  int last = len;
  while (last > 1) {
    --last;
    const BrownRI value = first[last];   // __pop_heap(first, last, last)
    first[last] = first[0];
    brown_adjust_heap(first, 0, last, value);
  }
}
KVFE_HD inline void brown_unguarded_linear_insert(BrownRI* a, int last) {
  const BrownRI val = a[last];
  int next = last - 1;
  while (brown_comp(val, a[next])) {
    a[last] = a[next];
    last = next;
    --next;
  }
  a[last] = val;
}
KVFE_HD void brown_insertion_sort(BrownRI* a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (brown_comp(a[i], a[first])) {
      const BrownRI val = a[i];
      for (int k = i; k > first; --k) a[k] = a[k - 1];   // std::move_backward(first, i, i + 1)
      a[first] = val;
    } else {
      brown_unguarded_linear_insert(a, i);
    }
  }
}
// std::sort(a, a + n, sort_pred()), one thread; `stack` holds 3 ints per pending range (at most 2 lg n + 2 ranges)
KVFE_HD void brown_std_sort(BrownRI* a, int n, int* stack) {
  if (n <= 0) return;
  int depth0 = 0;
  for (int m = n; m > 1; m >>= 1) depth0++;   // std::__lg(n)
  int sp = 0;
  stack[0] = 0;
  stack[1] = n;
  stack[2] = depth0 * 2;
  sp = 1;
  while (sp > 0) {
    --sp;
    int first = stack[3 * sp], last = stack[3 * sp + 1], depth = stack[3 * sp + 2];
    while (last - first > 16) {
      if (depth == 0) {
#ifdef KVFE_STDSORT_COUNT_HEAP
        KVFE_STDSORT_COUNT_HEAP++;
#endif
        brown_heap_sort(a + first, last - first);
        break;
      }
      --depth;
      // __unguarded_partition_pivot
      const int mid = first + (last - first) / 2;
      {  // __move_median_to_first(first, first + 1, mid, last - 1)
        const int A = first + 1, B = mid, Cc = last - 1;
        if (brown_comp(a[A], a[B])) {
          if (brown_comp(a[B], a[Cc])) brown_swap(a, first, B);
          else if (brown_comp(a[A], a[Cc])) brown_swap(a, first, Cc);
          else brown_swap(a, first, A);
        } else if (brown_comp(a[A], a[Cc])) {
          brown_swap(a, first, A);
        } else if (brown_comp(a[B], a[Cc])) {
          brown_swap(a, first, Cc);
        } else {
          brown_swap(a, first, B);
        }
      }
      int lo = first + 1, hi = last;
      for (;;) {  // __unguarded_partition(first + 1, last, pivot = first)
        while (brown_comp(a[lo], a[first])) ++lo;
        --hi;
        while (brown_comp(a[first], a[hi])) --hi;
        if (!(lo < hi)) break;
        brown_swap(a, lo, hi);
        ++lo;
      }
      const int cut = lo;
      stack[3 * sp] = cut;   // __introsort_loop(cut, last, depth_limit): disjoint range, order irrelevant
      stack[3 * sp + 1] = last;
      stack[3 * sp + 2] = depth;
      ++sp;
      last = cut;
    }
  }
  // __final_insertion_sort
  if (n > 16) {
    brown_insertion_sort(a, 0, 16);
    for (int i = 16; i < n; ++i) brown_unguarded_linear_insert(a, i);
  } else {
    brown_insertion_sort(a, 0, n);
  }
}

