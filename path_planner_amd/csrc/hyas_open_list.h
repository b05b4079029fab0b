// hyas_open_list.h -- the open list of one Hybrid A* search on the host (plain C++, no device code).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "hyas_hugepool.h"

namespace hyas
{

struct HeapItem
{
    float f;
    uint32_t seq;
    int32_t node;
    int32_t idx; // the node's cell (Node3D::setIdx), as the kernel computed it: not recomputed at every look
};
// Open list order: lowest f = g + h first (CompareNodes, include/hybrid_a_star.h:31-43); equal f: earliest
// push first.  (Boost's binomial heap leaves the order of equal keys unspecified.)  See Driver::before().

// The open list.  A search mostly pops what it has just pushed, so the best entries live in a small sorted array (best
// last: popping is a decrement, a new entry near the best moves a few entries) and the rest in a binary heap behind it;
// every entry of the array comes before every entry of the heap.  (f, seq) is a strict total order, so the pop sequence
// is the same whatever the container.
struct OpenList
{
    static constexpr int HOT = 32;
    struct After
    {
        bool operator()(const HeapItem &a, const HeapItem &b) const { return a.f > b.f || (a.f == b.f && a.seq > b.seq); }
    };
    HeapItem hot[HOT]; // hot[nh - 1] is the best entry, hot[0] the worst of the array
    int nh = 0;
    HVec<HeapItem> cold; // (storage from the huge-page pool: every query has one, touched out of order)
    bool empty() const { return nh == 0; } // (the array is refilled whenever it runs dry)
    size_t size() const { return (size_t)nh + cold.size(); }
    const HeapItem &top() const { return hot[nh - 1]; }
    // brings up to `count` entries of the heap into the array, below the ones it holds
    void refill(int count)
    {
        After after;
        count = std::min<int>(std::min<int>(count, HOT - nh), (int)cold.size());
        if (count <= 0)
            return;
        std::memmove(hot + count, hot, (size_t)nh * sizeof(HeapItem));
        for (int i = 0; i < count; ++i) // the heap gives its best first: it goes highest
        {
            std::pop_heap(cold.begin(), cold.end(), after);
            hot[count - 1 - i] = cold.back();
            cold.pop_back();
        }
        nh += count;
    }
    void push(const HeapItem &e)
    {
        After after;
        if (nh == 0 || (after(e, hot[0]) && !(cold.empty() && nh < HOT))) // behind everything in the array
        {
            if (nh == 0 && cold.empty())
            {
                hot[nh++] = e;
                return;
            }
            cold.push_back(e);
            std::push_heap(cold.begin(), cold.end(), after);
            if (nh == 0)
                refill(HOT / 2);
            return;
        }
        if (nh == HOT) // full: the worse half goes to the heap
        {
            for (int i = 0; i < HOT / 2; ++i)
            {
                cold.push_back(hot[i]);
                std::push_heap(cold.begin(), cold.end(), after);
            }
            std::memmove(hot, hot + HOT / 2, (size_t)(HOT / 2) * sizeof(HeapItem));
            nh = HOT / 2;
            if (after(e, hot[0])) // ... and e with it if it now comes behind the array
            {
                cold.push_back(e);
                std::push_heap(cold.begin(), cold.end(), after);
                return;
            }
        }
        int at = nh; // e goes in front of the entries that come after it
        while (at > 0 && after(e, hot[at - 1]))
            --at;
        // (here: hot[at - 1] comes after e or at == 0; entries at .. nh - 1 come before e)
        // the array is sorted worst first: e belongs below the entries that beat it
        std::memmove(hot + at + 1, hot + at, (size_t)(nh - at) * sizeof(HeapItem));
        hot[at] = e;
        ++nh;
    }
    void pop()
    {
        --nh;
        if (nh == 0)
            refill(HOT / 2);
    }
    template <class F>
    void for_each(F f) const
    {
        for (int i = 0; i < nh; ++i)
            f(hot[i]);
        for (const HeapItem &e : cold)
            f(e);
    }
};

} // namespace hyas
