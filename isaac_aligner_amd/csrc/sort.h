// std::sort as libstdc++ implements it (bits/stl_algo.h: __introsort_loop, __unguarded_partition_pivot,
// __move_median_to_first, __final_insertion_sort, heap-sort fallback of bits/stl_heap.h), restated over an index array.
//
// Why this exists: the reference sorts candidate lists with std::sort and comparators under which distinct elements
// compare equal (FragmentBuilder.cpp:284, SimpleIndelAligner.cpp:472) or that are not even strict weak orders
// (TemplateBuilder.cpp:697,708 with the 1e-7 epsilon).  Which of several "equal" elements survives the following
// unique/consolidate step therefore depends on the exact permutation std::sort produces, so the permutation itself is
// part of the behaviour to reproduce.  The algorithm is deterministic in the sequence of comparison results, hence
// running it on indices with the same comparator yields the same permutation.
#pragma once
#include "types.h"

namespace isaac
{

template <typename IdxT, typename Less>
struct ExactSort
{
    IdxT *a; Less less;
    ISAAC_HD ExactSort(IdxT *arr, Less l) : a(arr), less(l) {}
    ISAAC_HD bool lt(int i, int j) const { return less(a[i], a[j]); }
    ISAAC_HD void swp(int i, int j) { IdxT t = a[i]; a[i] = a[j]; a[j] = t; }

    // __unguarded_linear_insert
    ISAAC_HD void unguardedLinearInsert(int last)
    {
        IdxT val = a[last];
        int next = last - 1;
        while (less(val, a[next])) { a[last] = a[next]; last = next; --next; }
        a[last] = val;
    }
    // __insertion_sort
    ISAAC_HD void insertionSort(int first, int last)
    {
        if (first == last) return;
        for (int i = first + 1; i != last; ++i)
        {
            if (less(a[i], a[first]))
            {
                IdxT val = a[i];
                for (int k = i; k > first; --k) a[k] = a[k - 1];
                a[first] = val;
            }
            else unguardedLinearInsert(i);
        }
    }
    // __final_insertion_sort, _S_threshold = 16
    ISAAC_HD void finalInsertionSort(int first, int last)
    {
        if (last - first > 16)
        {
            insertionSort(first, first + 16);
            for (int i = first + 16; i != last; ++i) unguardedLinearInsert(i);
        }
        else insertionSort(first, last);
    }
    // __move_median_to_first(result, a, b, c)
    ISAAC_HD void moveMedianToFirst(int result, int x, int y, int z)
    {
        if (lt(x, y))
        {
            if (lt(y, z)) swp(result, y);
            else if (lt(x, z)) swp(result, z);
            else swp(result, x);
        }
        else if (lt(x, z)) swp(result, x);
        else if (lt(y, z)) swp(result, z);
        else swp(result, y);
    }
    // __unguarded_partition(first, last, pivot)
    ISAAC_HD int unguardedPartition(int first, int last, int pivot)
    {
        while (true)
        {
            while (lt(first, pivot)) ++first;
            --last;
            while (lt(pivot, last)) --last;
            if (!(first < last)) return first;
            swp(first, last);
            ++first;
        }
    }
    // bits/stl_heap.h __push_heap / __adjust_heap / __make_heap / __sort_heap, value-based as in libstdc++
    ISAAC_HD void pushHeap(int first, int holeIndex, int topIndex, IdxT value)
    {
        int parent = (holeIndex - 1) / 2;
        while (holeIndex > topIndex && less(a[first + parent], value))
        {
            a[first + holeIndex] = a[first + parent];
            holeIndex = parent;
            parent = (holeIndex - 1) / 2;
        }
        a[first + holeIndex] = value;
    }
    ISAAC_HD void adjustHeap(int first, int holeIndex, int len, IdxT value)
    {
        const int topIndex = holeIndex;
        int secondChild = holeIndex;
        while (secondChild < (len - 1) / 2)
        {
            secondChild = 2 * (secondChild + 1);
            if (less(a[first + secondChild], a[first + (secondChild - 1)])) secondChild--;
            a[first + holeIndex] = a[first + secondChild];
            holeIndex = secondChild;
        }
        if ((len & 1) == 0 && secondChild == (len - 2) / 2)
        {
            secondChild = 2 * (secondChild + 1);
            a[first + holeIndex] = a[first + (secondChild - 1)];
            holeIndex = secondChild - 1;
        }
        pushHeap(first, holeIndex, topIndex, value);
    }
    ISAAC_HD void heapSort(int first, int last)
    {
        const int len = last - first;
        if (len >= 2)
        {
            int parent = (len - 2) / 2;
            while (true)
            {
                IdxT value = a[first + parent];
                adjustHeap(first, parent, len, value);
                if (parent == 0) break;
                parent--;
            }
        }
        // __heap_select with middle == last selects nothing; __sort_heap:
        int l = last;
        while (l - first > 1)
        {
            --l;
            IdxT value = a[l];
            a[l] = a[first];
            adjustHeap(first, 0, l - first, value);
        }
    }
    // std::sort
    ISAAC_HD void sort(int n)
    {
        if (n <= 0) return;
        int lg = 0; for (int t = n; t > 1; t >>= 1) ++lg;   // std::__lg
        // __introsort_loop with the right-hand recursion replaced by an explicit stack (the two halves are disjoint ranges)
        int stackFirst[40], stackLast[40], stackDepth[40]; int sp = 0;
        stackFirst[0] = 0; stackLast[0] = n; stackDepth[0] = lg * 2; sp = 1;
        while (sp)
        {
            --sp;
            int first = stackFirst[sp], last = stackLast[sp], depth = stackDepth[sp];
            while (last - first > 16)
            {
                if (depth == 0) { heapSort(first, last); break; }
                --depth;
                const int mid = first + (last - first) / 2;
                moveMedianToFirst(first, first + 1, mid, last - 1);
                const int cut = unguardedPartition(first + 1, last, first);
                stackFirst[sp] = cut; stackLast[sp] = last; stackDepth[sp] = depth; ++sp;
                last = cut;
            }
        }
        finalInsertionSort(0, n);
    }
};

template <typename IdxT, typename Less> ISAAC_HD void exactSort(IdxT *a, int n, Less less) { ExactSort<IdxT, Less> s(a, less); s.sort(n); }

} // namespace isaac
