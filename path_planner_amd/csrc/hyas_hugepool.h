// hyas_hugepool.h -- memory for the host state of the searches (plain C++, no device code).
//
// A batch keeps an open list, a cell map, a node arena and a duplicate hash per live query: gigabytes that the worker
// threads touch one cache line at a time, in no order.  On 4 KiB pages every such touch also misses the TLB; on 2 MiB
// pages the page tables of the whole state fit the caches (map_large, chunks of 4096 queries: +8 % queries/s).  The
// pool maps 64 MiB regions, asks for transparent huge pages on them (MADV_HUGEPAGE; harmless where the kernel has them
// switched off) and hands out blocks of a few sizes per octave; freed blocks wait in a list per size for the next
// request of that size (containers grow by doubling, so the sizes repeat from query to query).  Regions go back to the
// system when trim() finds no block in use: hyas_destroy calls it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <new>
#include <vector>

#include <sys/mman.h>

namespace hyas
{

class HugePool
{
  public:
    static HugePool &instance()
    {
        static HugePool *pool = new HugePool; // never destroyed: contexts may be torn down after the static destructors ran
        return *pool;
    }
    void *allocate(size_t bytes)
    {
        if (bytes > BIGGEST)
        {
            void *p = map(bytes, nullptr);
            std::lock_guard<std::mutex> lock(mu_);
            ++in_use_;
            return p;
        }
        size_t rounded;
        const int cls = class_of(bytes, rounded);
        std::lock_guard<std::mutex> lock(mu_);
        ++in_use_;
        if (Free *f = free_[cls])
        {
            free_[cls] = f->next;
            return f;
        }
        if ((size_t)(end_ - cur_) < rounded)
        {
            Region r;
            cur_ = (char *)map(REGION, &r);
            end_ = cur_ + REGION;
            regions_.push_back(r);
        }
        void *p = cur_;
        cur_ += rounded;
        return p;
    }
    void deallocate(void *p, size_t bytes)
    {
        if (!p)
            return;
        if (bytes > BIGGEST)
        {
            // (mapped on its own: whole huge pages from a 2 MiB boundary)
            const size_t len = round_up(bytes, HUGE_PAGE);
            munmap(p, len);
            std::lock_guard<std::mutex> lock(mu_);
            --in_use_;
            return;
        }
        size_t rounded;
        const int cls = class_of(bytes, rounded);
        std::lock_guard<std::mutex> lock(mu_);
        Free *f = (Free *)p;
        f->next = free_[cls];
        free_[cls] = f;
        --in_use_;
    }
    // gives the regions back when nothing of them is in use; returns the bytes released
    size_t trim()
    {
        std::lock_guard<std::mutex> lock(mu_);
        if (in_use_ != 0)
            return 0;
        size_t n = 0;
        for (const Region &r : regions_)
            munmap(r.base, r.len), n += r.len;
        regions_.clear();
        cur_ = end_ = nullptr;
        for (Free *&f : free_)
            f = nullptr;
        return n;
    }
    size_t mapped_bytes()
    {
        std::lock_guard<std::mutex> lock(mu_);
        return regions_.size() * REGION;
    }
    // size classes: multiples of 64 B up to 1 KiB, four per octave above (5/4, 6/4, 7/4, 8/4 of a power of two)
    static int class_of(size_t bytes, size_t &rounded)
    {
        if (bytes <= 1024)
        {
            rounded = bytes ? round_up(bytes, 64) : 64;
            return (int)(rounded / 64) - 1; // 0 .. 15
        }
        const int e = 63 - __builtin_clzll((unsigned long long)(bytes - 1)); // 2^e < bytes <= 2^(e+1), e >= 10
        const size_t step = (size_t)1 << (e - 2);
        const size_t q = (bytes + step - 1) / step; // 5 .. 8
        rounded = q * step;
        return 16 + (e - 10) * 4 + (int)(q - 5);
    }
    static constexpr size_t REGION = (size_t)64 << 20, HUGE_PAGE = (size_t)2 << 20, BIGGEST = (size_t)16 << 20;
    static constexpr int N_CLASSES = 16 + 14 * 4;

  private:
    struct Free
    {
        Free *next;
    };
    struct Region
    {
        void *base;
        size_t len;
    };
    static size_t round_up(size_t v, size_t to) { return (v + to - 1) / to * to; }
    // `bytes` (rounded up to whole huge pages) at a 2 MiB boundary.  With `whole`: the mapping is kept whole and
    // returned through it; without: the bytes around the aligned block are unmapped at once.
    static void *map(size_t bytes, Region *whole)
    {
        const size_t len = round_up(bytes, HUGE_PAGE);
        char *raw = (char *)mmap(nullptr, len + HUGE_PAGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (raw == (char *)MAP_FAILED)
            throw std::bad_alloc();
        char *at = (char *)round_up((size_t)(uintptr_t)raw, HUGE_PAGE);
        if (whole)
            *whole = Region{raw, len + HUGE_PAGE};
        else
        {
            if (at > raw)
                munmap(raw, (size_t)(at - raw));
            if (at + len < raw + len + HUGE_PAGE)
                munmap(at + len, (size_t)(raw + len + HUGE_PAGE - (at + len)));
        }
        (void)madvise(at, len, MADV_HUGEPAGE);
        return at;
    }
    std::mutex mu_;
    std::vector<Region> regions_;
    char *cur_ = nullptr, *end_ = nullptr;
    Free *free_[N_CLASSES] = {};
    size_t in_use_ = 0;
};

template <class T>
struct HugeAlloc
{
    using value_type = T;
    HugeAlloc() = default;
    template <class U>
    HugeAlloc(const HugeAlloc<U> &)
    {
    }
    T *allocate(size_t n) { return (T *)HugePool::instance().allocate(n * sizeof(T)); }
    void deallocate(T *p, size_t n) { HugePool::instance().deallocate(p, n * sizeof(T)); }
    template <class U>
    bool operator==(const HugeAlloc<U> &) const
    {
        return true;
    }
    template <class U>
    bool operator!=(const HugeAlloc<U> &) const
    {
        return false;
    }
};

// a std::vector whose storage comes from the pool
template <class T>
using HVec = std::vector<T, HugeAlloc<T>>;

} // namespace hyas
