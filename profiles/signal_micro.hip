// How fast can the host learn that a short kernel has finished, and see what it wrote into pinned host memory?
// Build: hipcc --offload-arch=gfx950 -O2 -o profiles/signal_micro profiles/signal_micro.hip     Run on the GPU box.
// For a kernel of `blocks` workgroups that each write one 16-byte record {payload x3, tag} into pinned host memory:
//   sync    launch + hipStreamSynchronize
//   query   launch + spinning on hipStreamQuery
//   wrval   launch + hipStreamWriteValue32 into pinned host memory, host spins on the word
//   flagk   launch + a second one-thread kernel that stores the word (system scope), host spins on the word
//   tags    launch, host spins until every record carries this round's tag (no completion signal at all)
//   tagsbar the same with the records in fine-grained device memory, read by the host through the PCIe BAR
// Prints microseconds per round (median of `rounds`), the launch call's own duration, and for `tags` whether any record
// was ever seen with a matching tag but a stale payload.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <emmintrin.h>

#define CHECK(x)                                                                     \
    do                                                                               \
    {                                                                                \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess)                                                        \
        {                                                                            \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// every workgroup: a little work (spin `work` clock ticks), then lane 0 stores one record
__global__ void k_records(u32x4 *out, uint32_t tag, long long work)
{
    const long long t0 = clock64();
    while (clock64() - t0 < work)
        ;
    if (threadIdx.x == 0)
    {
        const u32x4 v = {tag * 3u + 1u, tag * 5u + 2u, blockIdx.x ^ tag, tag};
        __builtin_nontemporal_store(v, out + blockIdx.x);
    }
}
__global__ void k_flag(uint32_t *flag, uint32_t v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 2000, rounds = argc > 2 ? atoi(argv[2]) : 2000;
    const long long work = argc > 3 ? atoll(argv[3]) : 2000; // clock ticks of 10 ns
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    u32x4 *h_out = nullptr, *d_out = nullptr;
    uint32_t *h_flag = nullptr, *d_flag = nullptr;
    CHECK(hipHostMalloc((void **)&h_out, sizeof(u32x4) * blocks, hipHostMallocDefault));
    CHECK(hipHostMalloc((void **)&h_flag, 64, hipHostMallocDefault));
    CHECK(hipHostGetDevicePointer((void **)&d_out, h_out, 0));
    CHECK(hipHostGetDevicePointer((void **)&d_flag, h_flag, 0));
    for (int i = 0; i < blocks; ++i)
        h_out[i] = u32x4{0, 0, 0, 0};
    *h_flag = 0;
    uint32_t tag = 0;
    auto med = [](std::vector<double> &v) {
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
    };
    // records in device memory the host can address (large BAR), where there is such memory
    u32x4 *b_out = nullptr;
    if (hipExtMallocWithFlags((void **)&b_out, sizeof(u32x4) * blocks, hipDeviceMallocFinegrained) != hipSuccess)
        (void)hipGetLastError(), b_out = nullptr;
    if (b_out)
        CHECK(hipMemset(b_out, 0, sizeof(u32x4) * blocks));
    const char *names[6] = {"sync", "query", "wrval", "flagk", "tags", "tagsbar"};
    for (int mode = 0; mode < 6; ++mode)
    {
        if (mode == 5 && !b_out)
        {
            printf("tagsbar no host-addressable device memory\n");
            continue;
        }
        u32x4 *const k_out = mode == 5 ? b_out : d_out;
        const u32x4 *const r_out = mode == 5 ? b_out : h_out;
        const int use_rounds = mode == 5 ? std::min(rounds, 100) : rounds; // reads through the BAR take microseconds each
        std::vector<double> total, call;
        long long torn = 0;
        bool supported = true;
        for (int r = 0; r < use_rounds + 50 && supported; ++r)
        {
            ++tag;
            const double t0 = now_us();
            hipLaunchKernelGGL(k_records, dim3(blocks), dim3(256), 0, st, k_out, tag, work);
            if (mode == 2)
            {
                if (hipStreamWriteValue32(st, d_flag, tag, 0) != hipSuccess)
                {
                    (void)hipGetLastError();
                    supported = false;
                    break;
                }
            }
            if (mode == 3)
                hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, st, d_flag, tag);
            const double t1 = now_us();
            if (mode == 0)
                CHECK(hipStreamSynchronize(st));
            else if (mode == 1)
            {
                while (hipStreamQuery(st) == hipErrorNotReady)
                    ;
            }
            else if (mode == 2 || mode == 3)
            {
                while (*(volatile uint32_t *)h_flag != tag)
                    _mm_pause();
            }
            else // modes 4 and 5
            {
                for (int i = 0; i < blocks; ++i)
                {
                    for (;;)
                    {
                        alignas(16) uint32_t w[4];
                        _mm_store_si128((__m128i *)w, _mm_load_si128((const __m128i *)(r_out + i))); // one 16-byte load
                        if (w[3] == tag)
                        {
                            if (w[0] != tag * 3u + 1u || w[1] != tag * 5u + 2u || w[2] != ((uint32_t)i ^ tag))
                                ++torn;
                            break;
                        }
                        _mm_pause();
                    }
                }
            }
            const double t2 = now_us();
            if (mode >= 4) // the stream must have drained before the records are written again
                CHECK(hipStreamSynchronize(st));
            if (r >= 50)
                total.push_back(t2 - t0), call.push_back(t1 - t0);
            // results the host is about to read must really be this round's
            if (mode < 4)
                for (int i = 0; i < blocks; i += 97)
                    if (h_out[i][3] != tag)
                        ++torn;
        }
        if (!supported)
        {
            printf("%-6s not supported\n", names[mode]);
            continue;
        }
        printf("%-6s blocks %d: launch-to-seen %.1f us (median), launch call(s) %.1f us, stale or torn records %lld\n", names[mode], blocks, med(total), med(call),
               torn);
    }
    return 0;
}
