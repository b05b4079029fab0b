// hyas_flood.h -- the open queue of the 2D cost-to-goal flood
// (HybridAStar::BuildAStarCostMap, src/hybrid_a_star.cpp:1020-1074).
//
// The reference keeps the open cells in a std::priority_queue whose comparator
// is `l.second >= r.second` (include/hybrid_a_star.h:45-53).  That is not a
// strict ordering, so which of several cells of equal priority is popped first
// is decided by the exact moves of libstdc++'s binary heap (bits/stl_heap.h:
// __push_heap, __adjust_heap, __pop_heap).  Floating-point cell coordinates and
// costs depend on that order, so the heap is restated here move for move, over
// any storage that offers get(i) / set(i, entry).  Host + device.
#pragma once
#include <stdint.h>

#ifndef HD
#define HD __host__ __device__ __forceinline__
#endif

namespace hyas
{
namespace fl
{

// One open cell.  prio = the cost its parent had when the cell was first reached
// (src/hybrid_a_star.cpp:1068); (x, y) = its world coordinates as computed on that visit.
struct Ent
{
    float prio;
    uint32_t id;
    float x, y;
};

// std::__push_heap(first, hole, top, value, comp) with comp(a, b) = a.prio >= b.prio
// (every entry is fetched once and kept: a fetch is a trip to memory whose answer the next step has to wait for)
template <class Heap>
HD void sift_up(Heap &h, int hole, int top, const Ent &value)
{
    while (hole > top)
    {
        const int parent = (hole - 1) / 2;
        const Ent pe = h.get(parent);
        if (!(pe.prio >= value.prio))
            break;
        h.set(hole, pe);
        hole = parent;
    }
    h.set(hole, value);
}

// priority_queue::emplace: the entry is appended and sifted up.  size = entries before the call.
template <class Heap>
HD void push(Heap &h, int size, const Ent &value)
{
    sift_up(h, size, 0, value);
}

// priority_queue::top + pop: std::__pop_heap moves the last entry out, walks the hole from the root down to a leaf
// along the children the comparator prefers (std::__adjust_heap), then sifts the saved entry up from there.
// size = entries before the call (>= 1); the heap holds size - 1 entries afterwards.
template <class Heap>
HD Ent pop(Heap &h, int size)
{
    const Ent top = h.get(0);
    const Ent value = h.get(size - 1);
    const int len = size - 1;
    int hole = 0, child = 0;
    while (child < (len - 1) / 2)
    {
        child = 2 * (child + 1);
        const Ent right = h.get(child), left = h.get(child - 1); // both children in one trip
        const bool take_left = right.prio >= left.prio;
        if (take_left)
            child--;
        h.set(hole, take_left ? left : right);
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2)
    {
        child = 2 * (child + 1);
        h.set(hole, h.get(child - 1));
        hole = child - 1;
    }
    if (len > 0)
        sift_up(h, hole, 0, value);
    return top;
}

} // namespace fl
} // namespace hyas
