// Measurement tool (not part of the library): where does the time of the footprint-list kernel go?
// Variants of k_collision_footprint_window<5> on synthetic tables of the benchmark's shape (4096 x 4096 grid, 10 %
// occupied, 7200 lists of 1..4 cells, 10 M random poses), timed with HIP events:
//   0 stream only (pose loads, a trivial sum, verdict store)       3 full, poses through an LDS transpose (coalesced loads)
//   1 + index arithmetic and the LDS bitmap read, no tile gathers  4 full, heading bin by multiplication (not bit-exact)
//   2 full kernel as the library runs it
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o fp_micro profiles/fp_micro.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                      \
    do                                                                             \
    {                                                                              \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess)                                                      \
        {                                                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

struct Map
{
    const uint64_t *tiles;
    uint32_t tiles_x, tiles_y;
    const uint32_t *win;
    uint32_t entries;
    int pr, headings;
    float delta_heading, inv_delta;
};

constexpr int BB = 5;
constexpr uint32_t ROWMASK = (1u << BB) - 1u;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__device__ __forceinline__ uint32_t four_poses(const Map &m, const uint32_t *s_win, f32x4 a, f32x4 b, f32x4 c)
{
    if (V == 0 || V == 8)
    {
        const float s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w;
        return __float_as_uint(s) & 0x01010101u;
    }
    const int pr = m.pr;
    const float px[4] = {a.x, a.w, b.z, c.y}, py[4] = {a.y, b.x, b.w, c.z}, pt[4] = {a.z, b.y, c.x, c.w};
    uint32_t win[4];
    uint64_t tile[4][4];
    int sx[4], sy[4];
    bool in_table[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        const float x = px[k], y = py[k];
        const bool sane = fabsf(x) < 2147483648.f && fabsf(y) < 2147483648.f && pt[k] == pt[k];
        const int X = (int)x, Y = (int)y;
        int iX = (int)((x - (float)X) * pr);
        iX = iX > 0 ? iX : 0;
        int iY = (int)((y - (float)Y) * pr);
        iY = iY > 0 ? iY : 0;
        const int iT = V == 4 ? (int)(pt[k] * m.inv_delta) : (int)(pt[k] / m.delta_heading);
        in_table[k] = sane && !(iT < 0 || iT >= m.headings || iX >= pr || iY >= pr);
        const uint32_t entry = in_table[k] ? (uint32_t)((iY * pr + iX) * m.headings + iT) : 0u;
        win[k] = in_table[k] ? s_win[entry] : 0u;
        const int ox = (int)((win[k] >> 25) & 1u), oy = (int)((win[k] >> 26) & 1u);
        const int wx0 = X - (BB / 2 + ox), wy0 = Y - (BB / 2 + oy);
        const int tx0 = wx0 >> 3, ty0 = wy0 >> 3;
        sx[k] = wx0 & 7, sy[k] = wy0 & 7;
        uint32_t low = 0, high = 0;
#pragma unroll
        for (int i = 0; i < BB; ++i)
        {
            const uint32_t rb = (win[k] >> (i * BB)) & ROWMASK;
            if (sy[k] + i < 8)
                low |= rb;
            else
                high |= rb;
        }
        const uint32_t left_cols = (1u << (8 - sx[k])) - 1u;
        const uint32_t need[4] = {low & left_cols, low & ~left_cols, high & left_cols, high & ~left_cols};
#pragma unroll
        for (int t = 0; t < 4; ++t)
        {
            const int tx = tx0 + (t & 1), ty = ty0 + (t >> 1);
            uint64_t v = 0;
            if (need[t] && V != 1)
            {
                v = ~0ull;
                if (tx >= 0 && (uint32_t)tx < m.tiles_x && ty >= 0 && (uint32_t)ty < m.tiles_y)
                {
                    const uint64_t *p = m.tiles + (size_t)ty * m.tiles_x + tx;
                    if (V == 5)
                        v = __builtin_nontemporal_load(p);
                    else if (V == 6 || V == 10)
                        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else if (V == 7)
                        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    else if (V == 9)
                        v = *(const uint32_t *)p; // 4-byte gather
                    else
                        v = *p;
                }
            }
            tile[k][t] = v;
        }
    }
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        uint32_t hit = 0;
#pragma unroll
        for (int i = 0; i < BB; ++i)
        {
            const uint32_t rb = (win[k] >> (i * BB)) & ROWMASK;
            const int r = sy[k] + i;
            const bool upper = r >= 8;
            const uint64_t L = upper ? tile[k][2] : tile[k][0], R = upper ? tile[k][3] : tile[k][1];
            const int shr = 8 * (r & 7);
            const uint32_t lb = (uint32_t)(L >> shr) & 0xffu, rbt = (uint32_t)(R >> shr) & 0xffu;
            hit |= (((lb | (rbt << 8)) >> sx[k]) & ROWMASK) & rb;
        }
        v |= (in_table[k] && hit == 0) ? 1u << (8 * k) : 0u;
    }
    return v;
}

template <int V, int THREADS>
__global__ void __launch_bounds__(THREADS) k_variant(Map m, const float *__restrict__ xyt, size_t n, uint8_t *__restrict__ out)
{
    extern __shared__ uint32_t s_all[];
    uint32_t *s_win = s_all;
    for (uint32_t i = threadIdx.x; i < m.entries; i += blockDim.x)
        s_win[i] = m.win[i];
    __syncthreads();
    const size_t groups = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(xyt);
    if (V == 3 || V == 8 || V == 10)
    {
        // coalesced: a wave reads 3 KB (256 poses) as three fully contiguous 1 KB loads, writes them to its own LDS
        // strip and every lane reads back its four poses
        f32x4 *strip = reinterpret_cast<f32x4 *>(s_all + ((m.entries + 3) & ~3u)) + (threadIdx.x >> 6) * 192;
        const int lane = threadIdx.x & 63;
        for (size_t w0 = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) & ~(size_t)63; w0 < groups; w0 += stride)
        {
            const size_t base = 3 * w0; // first float4 of this wave's 64 groups
            f32x4 l[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                l[j] = (w0 + 64 <= groups) ? __builtin_nontemporal_load(src + base + j * 64 + lane) : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 3; ++j)
                strip[j * 64 + lane] = l[j];
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the strip is this wave's own: no barrier, only order
            const f32x4 a = strip[3 * lane], b = strip[3 * lane + 1], c = strip[3 * lane + 2];
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint32_t v = four_poses<(V == 3 ? 2 : V)>(m, s_win, a, b, c);
            if (w0 + lane < groups)
                __builtin_nontemporal_store(v, reinterpret_cast<uint32_t *>(out + 4 * (w0 + lane)));
        }
        return;
    }
    if (V == 12)
    {
        // coalesced loads through the LDS strip, the next iteration's loads requested before this one's gathers
        f32x4 *strip = reinterpret_cast<f32x4 *>(s_all + ((m.entries + 3) & ~3u)) + (threadIdx.x >> 6) * 192;
        const int lane = threadIdx.x & 63;
        size_t w0 = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) & ~(size_t)63;
        f32x4 l[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        if (w0 + 64 <= groups)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                l[j] = __builtin_nontemporal_load(src + 3 * w0 + j * 64 + lane);
        while (w0 < groups)
        {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                strip[j * 64 + lane] = l[j];
            const size_t wn = w0 + stride;
            if (wn + 64 <= groups)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    l[j] = __builtin_nontemporal_load(src + 3 * wn + j * 64 + lane);
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const f32x4 a = strip[3 * lane], b = strip[3 * lane + 1], c = strip[3 * lane + 2];
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint32_t v = four_poses<2>(m, s_win, a, b, c);
            if (w0 + lane < groups)
                __builtin_nontemporal_store(v, reinterpret_cast<uint32_t *>(out + 4 * (w0 + lane)));
            w0 = wn;
        }
        return;
    }
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    f32x4 a = {0, 0, 0, 0}, b = a, c = a;
    if (g < groups)
        a = __builtin_nontemporal_load(src + 3 * g), b = __builtin_nontemporal_load(src + 3 * g + 1), c = __builtin_nontemporal_load(src + 3 * g + 2);
    while (g < groups)
    {
        const size_t gn = g + stride;
        f32x4 na = {0, 0, 0, 0}, nb = na, nc = na;
        if (gn < groups)
            na = __builtin_nontemporal_load(src + 3 * gn), nb = __builtin_nontemporal_load(src + 3 * gn + 1),
            nc = __builtin_nontemporal_load(src + 3 * gn + 2);
        const uint32_t v = four_poses<V>(m, s_win, a, b, c);
        __builtin_nontemporal_store(v, reinterpret_cast<uint32_t *>(out + 4 * g));
        a = na, b = nb, c = nc;
        g = gn;
    }
}

// gathers only: poses made up in registers (a hash of the group index), one tile gather per pose, verdict store
template <int PER, int THREADS>
__global__ void __launch_bounds__(THREADS) k_gather_only(Map m, size_t n, uint8_t *__restrict__ out)
{
    const size_t groups = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint32_t n_tiles = m.tiles_x * m.tiles_y;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < groups; g += stride)
    {
        uint64_t t[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k)
        {
            uint32_t h = (uint32_t)(g * PER + k) * 2654435761u;
            h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
            t[k] = m.tiles[h % n_tiles];
        }
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k)
            v |= (uint32_t)((t[k] >> (k * 9)) & 1u) << (8 * (k & 3));
        __builtin_nontemporal_store(v, reinterpret_cast<uint32_t *>(out + 4 * g));
    }
}

template <int PER, int THREADS>
static float run_gather(const Map &m, size_t n, uint8_t *out, int blocks, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((k_gather_only<PER, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, m, n, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((k_gather_only<PER, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, m, n, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

template <int V, int THREADS>
static float run(const Map &m, const float *xyt, size_t n, uint8_t *out, int blocks, int reps, uint64_t *sum)
{
    size_t lds = ((m.entries + 3) & ~3u) * 4 + ((V == 3 || V == 8 || V == 10 || V == 12) ? (THREADS / 64) * 192 * 16 : 0);
    CK(hipFuncSetAttribute((const void *)k_variant<V, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((k_variant<V, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, m, xyt, n, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((k_variant<V, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, m, xyt, n, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint8_t> h(n);
    CK(hipMemcpy(h.data(), out, n, hipMemcpyDeviceToHost));
    uint64_t s = 0;
    for (uint8_t b : h)
        s += b;
    *sum = s;
    return ms * 1000.f / reps;
}

int main()
{
    const uint32_t W = 4096, H = 4096, tx = W / 8, ty = H / 8;
    const size_t n = 10000000;
    std::mt19937_64 rng(1);
    std::vector<uint64_t> tiles((size_t)tx * ty);
    for (auto &t : tiles)
    {
        uint64_t v = 0;
        for (int b = 0; b < 64; ++b)
            if (rng() % 10 == 0)
                v |= 1ull << b;
        t = v;
    }
    const int pr = 10, headings = 72;
    std::vector<uint32_t> win((size_t)pr * pr * headings);
    for (auto &w : win)
    {
        // 1..3 cells around the box centre, as the in-tree footprint gives (mean 1.6)
        uint32_t m = 1u << (2 * BB + 2);
        if (rng() % 2)
            m |= 1u << (2 * BB + 3);
        if (rng() % 10 == 0)
            m |= 1u << (3 * BB + 2);
        w = m | ((rng() & 1) << 25) | ((rng() & 1) << 26);
    }
    std::vector<float> xyt(3 * n);
    std::uniform_real_distribution<float> ux(8, W - 8), ut(0, 6.2831853f);
    for (size_t i = 0; i < n; ++i)
        xyt[3 * i] = ux(rng), xyt[3 * i + 1] = ux(rng), xyt[3 * i + 2] = ut(rng);
    uint64_t *d_tiles;
    uint32_t *d_win;
    float *d_xyt;
    uint8_t *d_out;
    CK(hipMalloc(&d_tiles, tiles.size() * 8));
    CK(hipMalloc(&d_win, win.size() * 4));
    CK(hipMalloc(&d_xyt, xyt.size() * 4));
    CK(hipMalloc(&d_out, n));
    CK(hipMemcpy(d_tiles, tiles.data(), tiles.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_win, win.data(), win.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_xyt, xyt.data(), xyt.size() * 4, hipMemcpyHostToDevice));
    Map m{d_tiles, tx, ty, d_win, (uint32_t)win.size(), pr, headings, (float)(2 * 3.14159265358979323846 / 72.0), 0.f};
    m.inv_delta = 1.f / m.delta_heading;
    uint64_t s;
    const int reps = 20;
#define RUN(V, T, B)                                                                           \
    {                                                                                          \
        float us = run<V, T>(m, d_xyt, n, d_out, B, reps, &s);                                 \
        printf("variant %d threads %4d blocks %4d : %7.1f us  (free %llu)\n", V, T, B, us, (unsigned long long)s); \
        fflush(stdout);                                                                        \
    }
    RUN(0, 512, 512)
    printf("gather only, 4 per thread (10 M gathers): 512x512 %.1f us, 256x4096 %.1f us, 1024x512 %.1f us\n", run_gather<4, 512>(m, n, d_out, 512, reps),
           run_gather<4, 256>(m, n, d_out, 4096, reps), run_gather<4, 1024>(m, n, d_out, 512, reps));
    printf("gather only, 8 per thread (20 M gathers): 512x512 %.1f us, 256x4096 %.1f us\n", run_gather<8, 512>(m, n, d_out, 512, reps),
           run_gather<8, 256>(m, n, d_out, 4096, reps));
    printf("gather only, 1 per thread (2.5 M gathers): 256x4096 %.1f us\n", run_gather<1, 256>(m, n, d_out, 4096, reps));
    fflush(stdout);
    RUN(1, 512, 512)
    RUN(2, 512, 512)
    RUN(3, 512, 512)
    RUN(12, 512, 512)
    RUN(12, 256, 1024)
    RUN(12, 1024, 256)
    RUN(12, 512, 1024)
    RUN(4, 512, 512)
    RUN(5, 512, 512)
    RUN(6, 512, 512)
    RUN(7, 512, 512)
    RUN(9, 512, 512)
    RUN(8, 512, 512)
    RUN(8, 1024, 256)
    RUN(10, 512, 512)
    RUN(10, 1024, 256)
    return 0;
}
