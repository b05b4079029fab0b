// hyas_kernels.hip -- gfx950 kernels of the Hybrid A* expansion path and their
// launch wrappers.  Build: hipcc --offload-arch=gfx950 -ffp-contract=off.
//
// None of this is a dense contraction, so there is no MFMA here.  The map
// (char grid, obstacle quads, per-key lists, heuristic table) is read-only and
// small, so it lives in L2/L1 after first touch; the kernels are organised so
// that lanes of a wave walk the same per-key obstacle list (neighbouring
// samples of one curve, successors of one node) and leave together through a
// ballot as soon as any lane reports a collision.
#include <hip/hip_runtime.h>

#include "hyas_dev.h"
#include <algorithm>
#include <cstdlib>

#include "hyas_kernels.h"
#include "hyas_adaptive.h"
#include "hyas_bezier.h"
#include "hyas_dubins.h"
#include "hyas_flood.h"
#include "hyas_rs.h"

namespace hyas
{

// LDS pointers carry their address space: through generic pointers every access would be a flat load that waits for
// both memory counters.
#define HYAS_LDS __attribute__((address_space(3)))

// ------------------------------------------------------------------ collision
__global__ void __launch_bounds__(256) k_collision_polygon(DevMap m, const float *__restrict__ xyt,
                                                           const float *__restrict__ pred_xy, size_t n,
                                                           uint8_t *__restrict__ out)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        float x = xyt[3 * i], y = xyt[3 * i + 1], t = xyt[3 * i + 2];
        const GlobalQuads gq{m.quads};
        bool ok = pred_xy ? traversable(m, gq, x, y, t, true, pred_xy[2 * i], pred_xy[2 * i + 1]) : traversable(m, gq, x, y, t, false, 0, 0);
        out[i] = ok ? 1 : 0;
    }
}

// Footprint-list mode: 4 poses per thread so the verdict store is one dword.
// (aligned == false: xyt is not 16-byte or out not 4-byte aligned; every access is then a scalar one)
__global__ void __launch_bounds__(256) k_collision_footprint(DevMap m, const float *__restrict__ xyt, size_t n,
                                                             uint8_t *__restrict__ out, bool aligned)
{
    const size_t groups = (n + 3) / 4;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x)
    {
        const size_t i0 = g * 4;
        if (aligned && i0 + 4 <= n)
        {
            // 48 contiguous bytes per thread: three 16-byte loads
            const float4 *p = reinterpret_cast<const float4 *>(xyt + 3 * i0);
            float4 a = p[0], b = p[1], c = p[2];
            uint32_t v = 0;
            v |= footprint_free(m, a.x, a.y, a.z) ? 1u : 0u;
            v |= footprint_free(m, a.w, b.x, b.y) ? 1u << 8 : 0u;
            v |= footprint_free(m, b.z, b.w, c.x) ? 1u << 16 : 0u;
            v |= footprint_free(m, c.y, c.z, c.w) ? 1u << 24 : 0u;
            *reinterpret_cast<uint32_t *>(out + i0) = v;
        }
        else
        {
            for (size_t i = i0; i < n; ++i)
                out[i] = footprint_free(m, xyt[3 * i], xyt[3 * i + 1], xyt[3 * i + 2]) ? 1 : 0;
        }
    }
}

// Footprint-list mode over the bit-tile grid with the packed footprint table in
// LDS (include/lookup.h:30-284 builds the lists; the test itself is the
// upstream configurationTest: every listed cell must be on the grid and free).
// Four poses per thread; the up to 16 tile words of a thread are requested
// before the first is used, so the gathers overlap.
template <int MAXC>
__global__ void __launch_bounds__(1024) k_collision_footprint_tiles(DevMap m, const float *__restrict__ xyt, size_t n,
                                                                    uint8_t *__restrict__ out)
{
    extern __shared__ uint4 s_fp[];
    for (uint32_t i = threadIdx.x; i < m.fp_entries; i += blockDim.x)
        s_fp[i] = reinterpret_cast<const uint4 *>(m.fp_pack)[i];
    __syncthreads();
    const int pr = m.pr;
    const size_t groups = n / 4;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x)
    {
        const float4 *p = reinterpret_cast<const float4 *>(xyt + 12 * g);
        const float4 a = p[0], b = p[1], c = p[2];
        const float px[4] = {a.x, a.w, b.z, c.y}, py[4] = {a.y, b.x, b.w, c.z}, pt[4] = {a.z, b.y, c.x, c.w};
        uint64_t word[4][MAXC];
        uint32_t bit[4][MAXC]; // bit index, 64: cell not listed, 65: off the grid or pose outside the table
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            const float x = px[k], y = py[k];
            // |x|, |y| < 2^31 (false for NaN): there (long)x == (int)x, so the fraction needs no 64-bit conversion.
            // Anything else is off the grid for the reference as well (its int conversion yields INT_MIN).
            const bool sane = hy_fabsf(x) < 2147483648.f && hy_fabsf(y) < 2147483648.f && pt[k] == pt[k];
            const int X = (int)x, Y = (int)y;
            int iX = (int)((x - (float)X) * pr);
            iX = iX > 0 ? iX : 0;
            int iY = (int)((y - (float)Y) * pr);
            iY = iY > 0 ? iY : 0;
            const int iT = (int)(pt[k] / m.delta_heading);
            const bool in_table = sane && !(iT < 0 || iT >= m.headings || iX >= pr || iY >= pr);
            uint4 e = make_uint4(0, 0, 0, 0);
            if (in_table)
                e = s_fp[(iY * pr + iX) * m.headings + iT];
            const uint32_t ew[4] = {e.x, e.y, e.z, e.w};
            const int len = in_table ? (int)(e.x & 0xff) : 0;
#pragma unroll
            for (int i = 0; i < MAXC; ++i)
            {
                const int o = 1 + 2 * i; // byte offset of (dx, dy)
                const int dx = (int)(int8_t)((ew[o >> 2] >> ((o & 3) * 8)) & 0xff);
                const int dy = (int)(int8_t)((ew[(o + 1) >> 2] >> (((o + 1) & 3) * 8)) & 0xff);
                const int cX = X + dx, cY = Y + dy;
                const bool listed = i < len;
                const bool on = cX >= 0 && (uint32_t)cX < m.w && cY >= 0 && (uint32_t)cY < m.h;
                word[k][i] = 0;
                bit[k][i] = !in_table ? 65u : !listed ? 64u : !on ? 65u : (uint32_t)(((cY & 7) << 3) | (cX & 7));
                if (listed && on)
                    word[k][i] = m.occ_tiles[(size_t)(cY >> 3) * m.tiles_x + (cX >> 3)];
            }
        }
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            bool free_ = bit[k][0] != 65u || MAXC == 0;
#pragma unroll
            for (int i = 0; i < MAXC; ++i)
            {
                if (bit[k][i] == 65u)
                    free_ = false;
                else if (bit[k][i] < 64u && ((word[k][i] >> bit[k][i]) & 1ull))
                    free_ = false;
            }
            v |= free_ ? 1u << (8 * k) : 0u;
        }
        *reinterpret_cast<uint32_t *>(out + 4 * g) = v;
    }
    // the last n % 4 poses
    if (blockIdx.x == 0 && threadIdx.x < (n & 3))
    {
        const size_t i = (n & ~(size_t)3) + threadIdx.x;
        out[i] = footprint_free(m, xyt[3 * i], xyt[3 * i + 1], xyt[3 * i + 2]) ? 1 : 0;
    }
}

// Footprint-list mode over the bit-tile grid with every list held in LDS as a bitmap over the box it was rasterised
// in (DevMap::fp_win).  A pose's box covers at most 2 x 2 bit tiles: the tiles its bitmap touches are fetched (one
// 8-byte gather each, usually a single one: the lists are short and an 8 x 8 tile is wide), the box's rows are cut
// out of them and ANDed with the bitmap.  A listed cell off the grid reads as occupied (tile bits beyond the grid
// are 1, tiles beyond the grid all ones), which is the verdict the per-cell test gives.
// Four poses per thread and iteration, one 4-byte verdict store.
template <int BB>
__global__ void __launch_bounds__(BB <= 5 ? 512 : 1024) k_collision_footprint_window(DevMap m, const float *__restrict__ xyt, size_t n,
                                                                                     uint8_t *__restrict__ out)
{
    constexpr int WORDS = (BB * BB + 2 + 31) / 32;
    constexpr uint32_t ROWMASK = (1u << BB) - 1u;
    extern __shared__ uint32_t s_win[];
    {
        const uint32_t total = m.fp_entries * WORDS;
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x)
            s_win[i] = m.fp_win[i];
    }
    __syncthreads();
    const int pr = m.pr;
    const size_t groups = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 *src = reinterpret_cast<const f32x4 *>(xyt);
    // A wave reads the 3 KB of its 256 poses as three fully contiguous 1 KB loads, parks them in its own strip of
    // LDS and every lane takes its four poses (48 bytes) back out of it: lane-strided 16-byte loads straight from
    // memory touch every cache line three times (measured 31 vs 24 us for the pose stream alone).
    HYAS_LDS f32x4 *strip = (HYAS_LDS f32x4 *)(s_win + ((m.fp_entries * WORDS + 3u) & ~3u)) + (threadIdx.x >> 6) * 192;
    const int lane = threadIdx.x & 63;
    for (size_t w0 = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) & ~(size_t)63; w0 < groups; w0 += stride)
    {
        const size_t g = w0 + lane;
        f32x4 a = {0, 0, 0, 0}, b = a, c = a;
        if (w0 + 64 <= groups) // uniform across the wave
        {
            f32x4 l[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                l[j] = __builtin_nontemporal_load(src + 3 * w0 + j * 64 + lane);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                strip[j * 64 + lane] = l[j];
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the strip is the wave's own: order, no barrier
            a = strip[3 * lane], b = strip[3 * lane + 1], c = strip[3 * lane + 2];
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        else if (g < groups)
            a = src[3 * g], b = src[3 * g + 1], c = src[3 * g + 2];
        const float px[4] = {a.x, a.w, b.z, c.y}, py[4] = {a.y, b.x, b.w, c.z}, pt[4] = {a.z, b.y, c.x, c.w};
        uint32_t win[4][WORDS];
        uint64_t tile[4][4];
        int sx[4], sy[4];
        bool in_table[4];
        // pass 1: bitmaps from LDS, tile gathers requested
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            const float x = px[k], y = py[k];
            // |x|, |y| < 2^31 (false for NaN): there (long)x == (int)x, so the fraction needs no 64-bit conversion.
            // Anything else is off the grid for the reference as well (its int conversion yields INT_MIN).
            const bool sane = hy_fabsf(x) < 2147483648.f && hy_fabsf(y) < 2147483648.f && pt[k] == pt[k];
            const int X = (int)x, Y = (int)y;
            int iX = (int)((x - (float)X) * pr);
            iX = iX > 0 ? iX : 0;
            int iY = (int)((y - (float)Y) * pr);
            iY = iY > 0 ? iY : 0;
            const int iT = (int)(pt[k] / m.delta_heading);
            in_table[k] = sane && !(iT < 0 || iT >= m.headings || iX >= pr || iY >= pr);
            const uint32_t entry = in_table[k] ? (uint32_t)((iY * pr + iX) * m.headings + iT) : 0u;
#pragma unroll
            for (int wv = 0; wv < WORDS; ++wv)
                win[k][wv] = in_table[k] ? s_win[entry * WORDS + wv] : 0u;
            const int ox = (int)((win[k][(BB * BB) >> 5] >> ((BB * BB) & 31)) & 1u);
            const int oy = (int)((win[k][(BB * BB + 1) >> 5] >> ((BB * BB + 1) & 31)) & 1u);
            const int wx0 = X - (BB / 2 + ox), wy0 = Y - (BB / 2 + oy);
            const int tx0 = wx0 >> 3, ty0 = wy0 >> 3;
            sx[k] = wx0 & 7, sy[k] = wy0 & 7;
            // which of the four tiles does the bitmap touch?
            uint32_t low = 0, high = 0;
#pragma unroll
            for (int i = 0; i < BB; ++i)
            {
                constexpr int dummy = 0;
                (void)dummy;
                const int bit0 = i * BB, wv = bit0 >> 5, sh = bit0 & 31;
                uint32_t rb = win[k][wv] >> sh;
                if (sh + BB > 32 && wv + 1 < WORDS)
                    rb |= win[k][wv + 1] << (32 - sh);
                rb &= ROWMASK;
                if (sy[k] + i < 8)
                    low |= rb;
                else
                    high |= rb;
            }
            const uint32_t left_cols = (1u << (8 - sx[k])) - 1u; // box columns that fall into the left tile
            const uint32_t need[4] = {low & left_cols, low & ~left_cols, high & left_cols, high & ~left_cols};
#pragma unroll
            for (int t = 0; t < 4; ++t)
            {
                const int tx = tx0 + (t & 1), ty = ty0 + (t >> 1);
                uint64_t v = 0;
                if (need[t])
                {
                    v = ~0ull;
                    if (tx >= 0 && (uint32_t)tx < m.tiles_x && ty >= 0 && (uint32_t)ty < m.tiles_y)
                        v = m.occ_tiles[(size_t)ty * m.tiles_x + tx];
                }
                tile[k][t] = v;
            }
        }
        // pass 2: rows of the box cut out of the tiles and tested against the bitmap
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            uint32_t hit = 0;
#pragma unroll
            for (int i = 0; i < BB; ++i)
            {
                const int bit0 = i * BB, wv = bit0 >> 5, sh = bit0 & 31;
                uint32_t rb = win[k][wv] >> sh;
                if (sh + BB > 32 && wv + 1 < WORDS)
                    rb |= win[k][wv + 1] << (32 - sh);
                rb &= ROWMASK;
                const int r = sy[k] + i;
                const bool upper = r >= 8;
                const uint64_t L = upper ? tile[k][2] : tile[k][0], R = upper ? tile[k][3] : tile[k][1];
                const int shr = 8 * (r & 7);
                const uint32_t lb = (uint32_t)(L >> shr) & 0xffu, rbt = (uint32_t)(R >> shr) & 0xffu;
                const uint32_t cells = ((lb | (rbt << 8)) >> sx[k]) & ROWMASK;
                hit |= cells & rb;
            }
            v |= (in_table[k] && hit == 0) ? 1u << (8 * k) : 0u;
        }
        if (g < groups)
            __builtin_nontemporal_store(v, reinterpret_cast<uint32_t *>(out + 4 * g));
    }
    // the last n % 4 poses
    if (blockIdx.x == 0 && threadIdx.x < (n & 3))
    {
        const size_t i = (n & ~(size_t)3) + threadIdx.x;
        out[i] = footprint_free(m, xyt[3 * i], xyt[3 * i + 1], xyt[3 * i + 2]) ? 1 : 0;
    }
}

// Footprint-list mode for boxes of up to 7 cells, the lists as 8 x 8 bitmaps (DevMap::fp_m8): instead of cutting the
// box out of the occupancy tiles, the bitmap is moved into tile coordinates.  The box's corner lies at (sx, sy) inside
// tile (tx0, ty0): the columns that stay in the left tile are shifted up by sx inside every row byte, the others down
// by 8 - sx into the right tile, and a shift by sy bytes splits both into the part over the upper and the lower tile.
// A tile is fetched iff its part of the bitmap is not empty, and the pose is free iff no fetched tile has a bit in
// common with its part.  Tiles off the grid come from the all-ones border of occ_pad (coordinates clamped to it).
// Four poses per thread and iteration, one 4-byte verdict store; poses arrive through a per-wave strip of LDS.
__global__ void __launch_bounds__(1024) k_collision_footprint_m8(DevMap m, int half_box, const float *__restrict__ xyt, size_t n, uint8_t *__restrict__ out)
{
    extern __shared__ uint32_t s_win[];
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    {
        const uint32_t total = m.fp_entries * 2;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(m.fp_m8);
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x)
            s_win[i] = src[i];
    }
    __syncthreads();
    const HYAS_LDS u32x2 *s_m8 = (const HYAS_LDS u32x2 *)s_win;
    const int pr = m.pr;
    const size_t groups = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 *src = reinterpret_cast<const f32x4 *>(xyt);
    HYAS_LDS f32x4 *strip = (HYAS_LDS f32x4 *)(s_win + ((m.fp_entries * 2 + 3u) & ~3u)) + (threadIdx.x >> 6) * 192;
    const int lane = threadIdx.x & 63;
    const int pitch = (int)m.tiles_x + 3;
    const int tx_max = (int)m.tiles_x, ty_max = (int)m.tiles_y;
    for (size_t w0 = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) & ~(size_t)63; w0 < groups; w0 += stride)
    {
        const size_t g = w0 + lane;
        f32x4 a = {0, 0, 0, 0}, b = a, c = a;
        if (w0 + 64 <= groups) // uniform across the wave
        {
            f32x4 l[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                l[j] = __builtin_nontemporal_load(src + 3 * w0 + j * 64 + lane);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                strip[j * 64 + lane] = l[j];
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the strip is the wave's own: order, no barrier
            a = strip[3 * lane], b = strip[3 * lane + 1], c = strip[3 * lane + 2];
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        else if (g < groups)
            a = src[3 * g], b = src[3 * g + 1], c = src[3 * g + 2];
        const float px[4] = {a.x, a.w, b.z, c.y}, py[4] = {a.y, b.x, b.w, c.z}, pt[4] = {a.z, b.y, c.x, c.w};
        uint64_t part[4][4], tile[4][4];
        bool in_table[4];
        // pass 1: the bitmap's four parts, tile gathers requested
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            const float x = px[k], y = py[k];
            // |x|, |y| < 2^31 (false for NaN): there (long)x == (int)x, so the fraction needs no 64-bit conversion.
            // Anything else is off the grid for the reference as well (its int conversion yields INT_MIN).
            const bool sane = hy_fabsf(x) < 2147483648.f && hy_fabsf(y) < 2147483648.f && pt[k] == pt[k];
            const int X = (int)x, Y = (int)y;
            int iX = (int)((x - (float)X) * pr);
            iX = iX > 0 ? iX : 0;
            int iY = (int)((y - (float)Y) * pr);
            iY = iY > 0 ? iY : 0;
            const int iT = (int)(pt[k] / m.delta_heading);
            in_table[k] = sane && !(iT < 0 || iT >= m.headings || iX >= pr || iY >= pr);
            const uint32_t entry = in_table[k] ? (uint32_t)((iY * pr + iX) * m.headings + iT) : 0u;
            const u32x2 raw = s_m8[entry];
            const uint32_t mlo = in_table[k] ? raw.x : 0u, mhi = in_table[k] ? (raw.y & 0x00FFFFFFu) : 0u;
            const int wx0 = X - (half_box + (int)((raw.y >> 30) & 1u)), wy0 = Y - (half_box + (int)(raw.y >> 31));
            const uint32_t sx = (uint32_t)wx0 & 7u, sh = ((uint32_t)wy0 & 7u) * 8u;
            const uint32_t keep = 0x01010101u * (0xFFu >> sx); // box columns that stay in the left tile, in every row byte
            const uint64_t left = ((uint64_t)((mhi & keep) << sx) << 32) | ((mlo & keep) << sx);
            const uint64_t right = ((uint64_t)((mhi & ~keep) >> (8u - sx)) << 32) | ((mlo & ~keep) >> (8u - sx));
            part[k][0] = left << sh, part[k][1] = right << sh;
            part[k][2] = (left >> 1) >> (63u - sh), part[k][3] = (right >> 1) >> (63u - sh); // sh == 0: nothing below
            // tile (tx0, ty0) and its three neighbours, clamped onto the all-ones border
            const int tx0 = min(max(wx0 >> 3, -1), tx_max), ty0 = min(max(wy0 >> 3, -1), ty_max);
            const uint64_t *t0 = m.occ_pad + ((ty0 + 1) * pitch + tx0 + 1);
#pragma unroll
            for (int t = 0; t < 4; ++t)
                tile[k][t] = part[k][t] ? t0[(t >> 1) * pitch + (t & 1)] : 0ull;
        }
        // pass 2
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            const uint64_t hit = (tile[k][0] & part[k][0]) | (tile[k][1] & part[k][1]) | (tile[k][2] & part[k][2]) | (tile[k][3] & part[k][3]);
            v |= (in_table[k] && hit == 0) ? 1u << (8 * k) : 0u;
        }
        if (g < groups)
            __builtin_nontemporal_store(v, reinterpret_cast<uint32_t *>(out + 4 * g));
    }
    // the last n % 4 poses
    if (blockIdx.x == 0 && threadIdx.x < (n & 3))
    {
        const size_t i = (n & ~(size_t)3) + threadIdx.x;
        out[i] = footprint_free(m, xyt[3 * i], xyt[3 * i + 1], xyt[3 * i + 2]) ? 1 : 0;
    }
}

// --------------------------------------------------------------- map ingest
// nav_msgs::OccupancyGrid value of every costmap cell (src/utility.cpp:185-204): 0 free, 99 inscribed, 100 lethal,
// -1 unknown, 1..98 in between.
__global__ void __launch_bounds__(256) k_occupancy(const uint8_t *__restrict__ cost, size_t n, int8_t *__restrict__ occ)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        const int c = cost[i];
        occ[i] = (int8_t)(c == 0 ? 0 : c == 253 ? 99 : c == 254 ? 100 : c == 255 ? -1 : 1 + (97 * (c - 1)) / 251);
    }
}

// Occupancy bits of the footprint-list test: one 64-bit word per 8 x 8 cells (bit 8 * (y & 7) + (x & 7); cost >= 250 or
// beyond the grid: set), once as [tiles_y][tiles_x] and once with a border of all-ones tiles (one tile left and above,
// two right and below), pitch tiles_x + 3.  One thread per word of the padded form.
__global__ void __launch_bounds__(256) k_occ_tiles(const uint8_t *__restrict__ cost, uint32_t w, uint32_t h, uint32_t tx, uint32_t ty,
                                                   uint64_t *__restrict__ tiles, uint64_t *__restrict__ padded)
{
    const uint32_t pitch = tx + 3;
    const size_t total = (size_t)pitch * (ty + 3);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    {
        const uint32_t py = (uint32_t)(i / pitch), px = (uint32_t)(i % pitch);
        uint64_t bits = ~0ull;
        if (px >= 1 && px <= tx && py >= 1 && py <= ty)
        {
            const uint32_t x0 = (px - 1) * 8, y0 = (py - 1) * 8;
            bits = 0;
            for (uint32_t r = 0; r < 8; ++r)
                for (uint32_t q = 0; q < 8; ++q)
                {
                    const uint32_t x = x0 + q, y = y0 + r;
                    if (x >= w || y >= h || cost[(size_t)y * w + x] >= 250)
                        bits |= 1ull << (8 * r + q);
                }
            tiles[(size_t)(py - 1) * tx + (px - 1)] = bits;
        }
        if (padded)
            padded[i] = bits;
    }
}

// in_range_obstacle_map_.emplace(key, list) keeps the list of the FIRST loop position (x outer, y inner) that maps to
// a key (src/collisiondetection.cpp:327-348): owner[key] = min position i * nys + j; 0xFFFFFFFF where no corner maps.
__global__ void __launch_bounds__(256) k_cell_owner(DevMap m, const float *__restrict__ xs, uint32_t nxs, const float *__restrict__ ys, uint32_t nys,
                                                    uint32_t *__restrict__ owner)
{
    const size_t total = (size_t)nxs * nys;
    const uint32_t cells = m.w * m.h;
    for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < total; c += (size_t)gridDim.x * blockDim.x)
    {
        const uint32_t key = grid_key(m, xs[c / nys], ys[c % nys]);
        if (key < cells)
            atomicMin(&owner[key], (uint32_t)c);
    }
}

// Bins for the in-range lists: the grid keys in tiles of HYAS_BIN_TILE x HYAS_BIN_TILE, and per tile the obstacle
// polygons whose bounding box, grown by `range`, reaches the box around the owner corners of the tile's keys, in
// obstacle order.  A polygon nearer than `range` to a corner has its bounding box nearer still, so a key's list is a
// sub-sequence of its tile's bin; the growth is a little more than `range`, to cover the rounding of the distance.
// One wave per tile, 64 polygons per pass, positions inside a pass by ballot.  FILL = false counts
// (off[tile] = bin length), FILL = true writes the ids behind off[tile].
template <bool FILL>
__global__ void __launch_bounds__(256) k_range_bins(const Quad *__restrict__ quads, uint32_t n_quads, const float *__restrict__ xs,
                                                    const float *__restrict__ ys, uint32_t nys, const uint32_t *__restrict__ owner, uint32_t w, uint32_t h,
                                                    float range, RangeBins bins)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tiles = bins.tiles_x * bins.tiles_y;
    const size_t waves = (size_t)gridDim.x * (blockDim.x >> 6);
    for (size_t t = blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6); t < tiles; t += waves)
    {
        const uint32_t x0 = (uint32_t)(t % bins.tiles_x) * HYAS_BIN_TILE, y0 = (uint32_t)(t / bins.tiles_x) * HYAS_BIN_TILE;
        float xlo = 3.0e38f, xhi = -3.0e38f, ylo = 3.0e38f, yhi = -3.0e38f;
        for (uint32_t k = lane; k < HYAS_BIN_TILE * HYAS_BIN_TILE; k += 64)
        {
            const uint32_t kx = x0 + k % HYAS_BIN_TILE, ky = y0 + k / HYAS_BIN_TILE;
            const uint32_t pos = kx < w && ky < h ? owner[(size_t)ky * w + kx] : 0xFFFFFFFFu;
            if (pos != 0xFFFFFFFFu)
            {
                const float px = xs[pos / nys], py = ys[pos % nys];
                xlo = fminf(xlo, px), xhi = fmaxf(xhi, px), ylo = fminf(ylo, py), yhi = fmaxf(yhi, py);
            }
        }
        for (int d = 1; d < 64; d <<= 1)
        {
            xlo = fminf(xlo, __shfl_xor(xlo, d, 64)), xhi = fmaxf(xhi, __shfl_xor(xhi, d, 64));
            ylo = fminf(ylo, __shfl_xor(ylo, d, 64)), yhi = fmaxf(yhi, __shfl_xor(yhi, d, 64));
        }
        uint32_t n = 0;
        if (xlo <= xhi)
        {
            const float grow = range + 1e-3f * range + 1e-5f * (fabsf(xlo) + fabsf(xhi) + fabsf(ylo) + fabsf(yhi) + 1.f) + bins.extra;
            const uint32_t base = FILL ? bins.off[t] : 0u;
            for (uint32_t q0 = 0; q0 < n_quads; q0 += 64)
            {
                const uint32_t q = q0 + lane;
                bool hit = false;
                if (q < n_quads)
                {
                    const float4 box = *reinterpret_cast<const float4 *>(&quads[q].xmin); // xmin, ymin, xmax, ymax
                    hit = box.x - grow <= xhi && box.z + grow >= xlo && box.y - grow <= yhi && box.w + grow >= ylo;
                }
                const unsigned long long mask = __ballot(hit);
                if (FILL && hit)
                    bins.idx[base + n + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)q;
                n += (uint32_t)__popcll(mask);
            }
        }
        if (!FILL && lane == 0)
            bins.off[t] = n;
    }
}

// The in-range list of every grid key (src/collisiondetection.cpp:335-345): the obstacle polygons nearer than `range`
// to the key's owner corner, in obstacle order, picked from the bin of the key's tile.  One wave per key, 64 polygons
// per pass, positions inside a pass by ballot.  FILL = false counts (off[key] = list length), FILL = true writes the
// ids behind off[key].
template <bool FILL>
__global__ void __launch_bounds__(256) k_in_range_lists(const Quad *__restrict__ quads, const float *__restrict__ xs, const float *__restrict__ ys,
                                                        uint32_t nys, const uint32_t *__restrict__ owner, uint32_t w, uint32_t cells, float range,
                                                        RangeBins bins, uint32_t *__restrict__ off, uint16_t *__restrict__ idx)
{
    const uint32_t lane = threadIdx.x & 63u;
    const size_t waves = (size_t)gridDim.x * (blockDim.x >> 6);
    for (size_t key = blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6); key < cells; key += waves)
    {
        const uint32_t pos = owner[key];
        uint32_t n = 0;
        if (pos != 0xFFFFFFFFu)
        {
            const F2 p = f2(xs[pos / nys], ys[pos % nys]);
            const uint32_t base = FILL ? off[key] : 0u;
            const uint32_t t = (uint32_t)(key / w) / HYAS_BIN_TILE * bins.tiles_x + (uint32_t)(key % w) / HYAS_BIN_TILE;
            const uint32_t b1 = bins.off[t + 1];
            for (uint32_t b0 = bins.off[t]; b0 < b1; b0 += 64)
            {
                const bool have = b0 + lane < b1;
                const uint32_t q = have ? bins.idx[b0 + lane] : 0u;
                const bool hit = have && dist_quad_point(quads[q], p) < range;
                const unsigned long long mask = __ballot(hit);
                if (FILL && hit)
                    idx[base + n + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)q;
                n += (uint32_t)__popcll(mask);
            }
        }
        if (!FILL && lane == 0)
            off[key] = n;
    }
}

// Exact pruning of the per-key lists (what hyas_prune.h does on the host, same arithmetic in double): per key the
// entries whose quad can still take part in a pose test (pose list) or in a swept-segment test (segment list; entries
// that only the ray-casting inside test can reach carry bit 15).  One thread per key; FILL = false counts
// (pose_off[key], seg_off[key] = lengths), FILL = true writes the ids behind the offsets.
template <bool FILL>
__global__ void __launch_bounds__(256) k_prune_lists(const Quad *__restrict__ quads, const uint32_t *__restrict__ cell_off,
                                                     const uint16_t *__restrict__ cell_idx, PruneArgs a, uint32_t *__restrict__ pose_off,
                                                     uint16_t *__restrict__ pose_idx, uint32_t *__restrict__ seg_off, uint16_t *__restrict__ seg_idx)
{
    for (size_t key = blockIdx.x * (size_t)blockDim.x + threadIdx.x; key < a.cells; key += (size_t)gridDim.x * blockDim.x)
    {
        const double x0 = a.ox + a.res * (double)(key % a.w), y0 = a.oy + a.res * (double)(key / a.w);
        const double x1 = x0 + a.res, y1 = y0 + a.res;
        uint32_t np = FILL ? pose_off[key] : 0u, ns = FILL ? seg_off[key] : 0u;
        for (uint32_t k = cell_off[key]; k < cell_off[key + 1]; ++k)
        {
            const uint16_t qi = cell_idx[k];
            const float4 box = *reinterpret_cast<const float4 *>(&quads[qi].xmin); // xmin, ymin, xmax, ymax
            const double xmin = box.x, ymin = box.y, xmax = box.z, ymax = box.w;
            const bool pose = !(xmax < x0 - a.diag || xmin > x1 + a.diag || ymax < y0 - a.diag || ymin > y1 + a.diag);
            const bool seg = !(xmax < x0 - a.reach || xmin > x1 + a.reach || ymax < y0 - a.reach || ymin > y1 + a.reach);
            const bool ray = !seg && !(ymax < y0 - a.lmax - a.slack || ymin > y1 + a.lmax + a.slack) && xmax >= x0 - a.lmax - a.slack;
            if (pose)
            {
                if (FILL)
                    pose_idx[np] = qi;
                ++np;
            }
            if (seg || ray)
            {
                if (FILL)
                    seg_idx[ns] = seg ? qi : (uint16_t)(qi | 0x8000);
                ++ns;
            }
        }
        if (!FILL)
            pose_off[key] = np, seg_off[key] = ns;
    }
}

// In-place exclusive prefix sum of v[0..n) with the total in v[n] and, as 64 bits, in *total.  One workgroup of 1024
// lanes, four entries per lane and pass.
__global__ void __launch_bounds__(1024) k_exclusive_scan_u32(uint32_t *__restrict__ v, size_t n, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long wave_sum[16];
    __shared__ unsigned long long carry_s;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        carry_s = 0;
    __syncthreads();
    for (size_t base = 0; base < n; base += 4096)
    {
        const size_t i0 = base + 4 * (size_t)threadIdx.x;
        uint32_t a[4];
        unsigned long long mine = 0;
        for (int k = 0; k < 4; ++k)
        {
            a[k] = i0 + k < n ? v[i0 + k] : 0u;
            mine += a[k];
        }
        unsigned long long incl = mine;
        for (int d = 1; d < 64; d <<= 1)
        {
            const unsigned long long up = __shfl_up(incl, d, 64);
            if ((int)lane >= d)
                incl += up;
        }
        if (lane == 63)
            wave_sum[wave] = incl;
        __syncthreads();
        unsigned long long before = carry_s;
        for (uint32_t k = 0; k < wave; ++k)
            before += wave_sum[k];
        unsigned long long run = before + incl - mine;
        for (int k = 0; k < 4; ++k)
        {
            if (i0 + k < n)
                v[i0 + k] = (uint32_t)run;
            run += a[k];
        }
        __syncthreads();
        if (threadIdx.x == 1023)
            carry_s = run;
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        v[n] = (uint32_t)carry_s;
        *total = carry_s;
    }
}

// Adaptive step size / steering, map ingest: the adaptive maps of every grid key in one kernel.  For the key's owner
// corner: the minimum distance to its in-range obstacles, 1000 without any (CollisionDetection::SetMinDistanceMap,
// src/collisiondetection.cpp:1403-1447); the sweep of HYAS_SWEEP_MAX rays at most, each the nearest hit among the
// in-range obstacles or `radius` when none is hit (FindNoCollisionDistance, :973-1011; six rays per lane, into LDS);
// then the run-length pass over the sweep and the cut to the circle (ad::ranges_of_sweep_each,
// ad::cut_range_to_circle: SetDistanceAngleRangeMap and FindFreeAngleRangeAndObstacleAngleRange) on the wave's first
// lane.  Nothing of size corners x rays leaves the chip.  One wave per key; FILL = false counts
// (off[key] = number of ranges, min_distance[key]), FILL = true writes {label, start, width} behind off[key].
// kept (optional): room for `keep` ranges per key.  The counting pass leaves a key's first `keep` ranges there, and the
// filling pass copies them for every key that has no more than that instead of sweeping again.
template <bool FILL>
__global__ void __launch_bounds__(256) k_adaptive_ranges(DevMap m, const float *__restrict__ xs, const float *__restrict__ ys, uint32_t nys,
                                                         const uint32_t *__restrict__ owner, const float *__restrict__ angles, uint32_t n_ang,
                                                         float radius, float *__restrict__ min_distance, uint32_t *__restrict__ off,
                                                         float *__restrict__ rng3, float *__restrict__ kept, uint32_t keep)
{
    __shared__ float s_dist[4][HYAS_SWEEP_MAX];
    __shared__ float s_ang[HYAS_SWEEP_MAX];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t cells = m.w * m.h;
    for (uint32_t a = threadIdx.x; a < n_ang; a += blockDim.x)
        s_ang[a] = angles[a];
    // (every wave of a workgroup makes the same number of trips: the barriers inside are the workgroup's)
    for (size_t base = (size_t)blockIdx.x * 4; base < cells; base += (size_t)gridDim.x * 4)
    {
        const size_t key = base + wave;
        const uint32_t pos = key < cells ? owner[key] : 0xFFFFFFFFu;
        const bool have = pos != 0xFFFFFFFFu;
        float md = 1000.f;
        F2 p = f2(0.f, 0.f);
        uint32_t b = 0, e = 0;
        const uint32_t n_known = FILL && have && kept ? off[key + 1] - off[key] : 0xFFFFFFFFu;
        const bool copy = n_known <= keep; // (uniform across the wave)
        if (copy && lane < 3 * n_known)
            rng3[3 * (size_t)off[key] + lane] = kept[(size_t)key * keep * 3 + lane];
        if (have && !copy)
        {
            p = f2(xs[pos / nys], ys[pos % nys]);
            b = m.cell_off[key], e = m.cell_off[key + 1];
            for (uint32_t k = b + lane; k < e; k += 64)
            {
                const float d = dist_quad_point(m.quads[m.cell_idx[k]], p);
                if (md > d)
                    md = d;
            }
            for (int d = 1; d < 64; d <<= 1)
            {
                const float o = __shfl_xor(md, d, 64);
                if (md > o)
                    md = o;
            }
            for (uint32_t a = lane; a < n_ang; a += 64)
            {
                float best = radius;
                for (uint32_t k = b; k < e; ++k)
                {
                    const float t = ad::ray_distance(m.quads[m.cell_idx[k]], p, s_ang[a]);
                    if (t < 0)
                        continue;
                    if (best > t)
                        best = t;
                }
                s_dist[wave][a] = best;
            }
        }
        __syncthreads();
        if (have && !copy && lane == 0)
        {
            uint32_t n = 0;
            const uint32_t at = FILL ? off[key] : 0u;
            ad::ranges_of_sweep_each([&](uint32_t i) { return s_ang[i]; }, [&](uint32_t i) { return s_dist[wave][i]; }, n_ang, md, radius,
                                     [&](float label, float start, float width) {
                                         ad::Range r{label, start, width};
                                         if (!ad::cut_range_to_circle(r))
                                             return;
                                         if (FILL || (kept && n < keep))
                                         {
                                             float *o = FILL ? rng3 + 3 * (size_t)(at + n) : kept + 3 * ((size_t)key * keep + n);
                                             o[0] = r.dist, o[1] = r.start, o[2] = r.range;
                                         }
                                         ++n;
                                     });
            if (!FILL)
                off[key] = n;
        }
        if (!FILL && lane == 0 && key < cells)
        {
            min_distance[key] = md;
            if (!have)
                off[key] = 0;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------ heuristic table
// One table entry per thread (src/lookup_table.cpp:209-218): cost of the
// Reeds-Shepp path from the origin to (xs[i], ys[j], ts[k]).  The kernel is bound by the issue of its f64
// arithmetic (no memory traffic to speak of), so it runs the length-only form of the solver (rs::shortest_length:
// shared polar forms, candidates whose first arc is negative dropped before their arc tangents) with the arc
// tangent's five table values in LDS; one entry per thread and many small workgroups, so that the tail of the launch
// is one workgroup's time.
template <typename Index> // uint32_t while the table has fewer than 2^32 entries: the index arithmetic is per entry
__global__ void __launch_bounds__(256) k_table_rs(const float *__restrict__ xs, uint32_t nx,
                                                  const float *__restrict__ ys, uint32_t ny,
                                                  const float *__restrict__ ts, uint32_t k0, uint32_t k1, double rho,
                                                  float *__restrict__ out)
{
    __shared__ double s_quarter_atans[8];
    {
        const double qa[8] = HY_QUARTER_ATANS;
        if (threadIdx.x < 8)
            s_quarter_atans[threadIdx.x] = qa[threadIdx.x];
        __syncthreads();
    }
    const Index total = (Index)(k1 - k0) * ny * nx;
    for (Index e = blockIdx.x * (Index)blockDim.x + threadIdx.x; e < total; e += (Index)gridDim.x * blockDim.x)
    {
        uint32_t i = (uint32_t)(e % nx);
        uint32_t j = (uint32_t)((e / nx) % ny);
        uint32_t k = k0 + (uint32_t)(e / ((Index)nx * ny));
        const double q0[3] = {0, 0, 0};
        const double q1[3] = {(double)xs[i], (double)ys[j], (double)ts[k]};
        double x, y, phi;
        rs::normalise(q0, q1, rho, x, y, phi);
        out[e] = (float)(rho * rs::shortest_length(x, y, phi, s_quarter_atans));
        if (sizeof(Index) == 4) // launched with one thread per entry
            break;
    }
}

// The same for curve_type 0 (src/lookup_table.cpp:147-180): OMPL's DubinsStateSpace::distance in double
// (db::ompl::distance, hyas_dubins.h), stored as float.
__global__ void __launch_bounds__(256) k_table_dubins(const float *__restrict__ xs, uint32_t nx,
                                                      const float *__restrict__ ys, uint32_t ny,
                                                      const float *__restrict__ ts, uint32_t k0, uint32_t k1, float rho,
                                                      float *__restrict__ out)
{
    const size_t total = (size_t)(k1 - k0) * ny * nx;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
    {
        uint32_t i = (uint32_t)(e % nx);
        uint32_t j = (uint32_t)((e / nx) % ny);
        uint32_t k = k0 + (uint32_t)(e / ((size_t)nx * ny));
        out[e] = (float)db::ompl::distance(0., 0., 0., (double)xs[i], (double)ys[j], (double)ts[k], (double)rho);
    }
}

// The same for curve_type 2 (src/lookup_table.cpp:226-277): 100-interval polyline length of the cubic Bezier curve
// from the origin, heading 0, to (xs[i], ys[j], ts[k]) with both inner control points a third of the chord away.
// The 101 parameter values and their Bernstein weights are the same for every entry: the workgroup forms them once
// in LDS, and every lane reads the same four weights per vertex (one broadcast read).
__global__ void __launch_bounds__(256) k_table_bezier(const float *__restrict__ xs, uint32_t nx, const float *__restrict__ ys, uint32_t ny,
                                                      const float *__restrict__ ts, uint32_t k0, uint32_t k1, float *__restrict__ out)
{
    __shared__ float4 s_w[HYAS_BZ_INTERVALS + 1];
    if (threadIdx.x <= HYAS_BZ_INTERVALS)
    {
        float w[4];
        bz::lib_weights(bz::lib_param(threadIdx.x), w);
        s_w[threadIdx.x] = make_float4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    const size_t total = (size_t)(k1 - k0) * ny * nx;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
    {
        uint32_t i = (uint32_t)(e % nx);
        uint32_t j = (uint32_t)((e / nx) % ny);
        uint32_t k = k0 + (uint32_t)(e / ((size_t)nx * ny));
        float px[4], py[4];
        bz::lib_control(xs[i], ys[j], ts[k], px, py);
        float4 wv = s_w[0];
        float w[4] = {wv.x, wv.y, wv.z, wv.w};
        float ax = bz::lib_axis(w, px), ay = bz::lib_axis(w, py), length = 0.0f;
        for (int v = 1; v <= HYAS_BZ_INTERVALS; ++v)
        {
            wv = s_w[v];
            w[0] = wv.x, w[1] = wv.y, w[2] = wv.z, w[3] = wv.w;
            const float bx = bz::lib_axis(w, px), by = bz::lib_axis(w, py);
            const float dx = bx - ax, dy = by - ay;
            length += hy_sqrtf(dx * dx + dy * dy);
            ax = bx, ay = by;
        }
        out[e] = length;
    }
}

// First-write-wins emplace (src/lookup_table.cpp:215) in two passes: pass 0
// records for every key the earliest loop position (x outer, y, theta inner)
// that maps to it, pass 1 lets exactly that position store its cost.
__global__ void __launch_bounds__(256) k_table_install(DevMap m, const float *__restrict__ raw,
                                                       const float *__restrict__ xs, uint32_t nx,
                                                       const float *__restrict__ ys, uint32_t ny,
                                                       const float *__restrict__ ts, uint32_t ns,
                                                       uint32_t *__restrict__ rank, float *__restrict__ table,
                                                       uint64_t table_size, int pass)
{
    const size_t total = (size_t)ns * ny * nx;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
    {
        uint32_t i = (uint32_t)(e % nx);
        uint32_t j = (uint32_t)((e / nx) % ny);
        uint32_t k = (uint32_t)(e / ((size_t)nx * ny));
        if (xs[i] == 0 && ys[j] == 0)
            continue; // :204-208
        int key = table_key(m, xs[i], ys[j], ts[k]);
        if (key < 0 || (uint64_t)key >= table_size)
            continue;
        uint32_t order = (i * ny + j) * ns + k;
        if (pass == 0)
            atomicMin(&rank[key], order);
        else if (rank[key] == order)
            table[key] = raw[e];
    }
}

__global__ void __launch_bounds__(256) k_table_lookup(DevMap m, const float *__restrict__ xyt, size_t n,
                                                      float *__restrict__ cost)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        cost[i] = table_cost(m, xyt[3 * i], xyt[3 * i + 1], xyt[3 * i + 2]);
}

__global__ void __launch_bounds__(256) k_fill_u32(uint32_t *p, size_t n, uint32_t v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = v;
}

// ------------------------------------------------------------- expansion step
__device__ __forceinline__ double wave_min(double v)
{
    for (int off = 32; off > 0; off >>= 1)
    {
        double o = __shfl_xor(v, off, 64);
        v = (o < v) ? o : v;
    }
    return v;
}

// v of lane `src` (wave-uniform), as a wave-uniform value
__device__ __forceinline__ double lane_value(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// Lanes 0..43 of one wave evaluate one Reeds-Shepp candidate each; the five
// families are then reduced in reference order.  Every lane returns the path.
__device__ __forceinline__ rs::Path solve_wave(double x, double y, double phi, int lane)
{
    rs::Path mine;
    double L = 0;
    bool valid = false;
    if (lane < RS_NUM_CANDIDATES)
        valid = rs::candidate(lane, x, y, phi, mine, L);
    const double inf = __builtin_huge_val();
    rs::Path best = rs::empty_path();
    for (int fam = 0; fam < RS_NUM_FAMILIES; ++fam)
    {
        const bool in_fam = lane >= rs::family_first(fam) && lane < rs::family_first(fam + 1);
        const double key = (valid && in_fam && L == L) ? L : inf;
        const double mn = wave_min(key);
        const double bound = rs::family_bound(fam, best.total);
        if (bound > mn) // uniform across the wave
        {
            unsigned long long who = __ballot(key == mn);
            int src = __ffsll((long long)who) - 1;
            best.type = __shfl(mine.type, src, 64);
            for (int i = 0; i < 5; ++i)
                best.len[i] = __shfl(mine.len[i], src, 64);
            best.total = __shfl(mine.total, src, 64);
        }
    }
    return best;
}

// Number of samples the loop `for (seg = 0; seg <= length; seg += step)`
// produces (src/ReedsShepp.cpp:566): seg_tab[k] holds seg after k additions.
__device__ __forceinline__ uint32_t count_samples(const double *__restrict__ seg_tab, uint32_t cap, double length)
{
    uint32_t lo = 0, hi = cap; // first k with seg_tab[k] > length
    while (lo < hi)
    {
        uint32_t mid = (lo + hi) >> 1;
        if (seg_tab[mid] <= length)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// Number of samples of the Dubins shot loop `x = 0; while (x < length) x += 0.2 * res`
// (src/hybrid_a_star.cpp:396-424): seg_tab[k] holds x after k additions.
__device__ __forceinline__ uint32_t count_samples_db(const float *__restrict__ seg_tab, uint32_t cap, float length)
{
    uint32_t lo = 0, hi = cap; // first k with !(seg_tab[k] < length)
    while (lo < hi)
    {
        uint32_t mid = (lo + hi) >> 1;
        if (seg_tab[mid] < length)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

#define STEP_LDS_QUADS 192 // obstacle quads kept in LDS (12 KB)

// Quads from the workgroup's LDS copy when there is one, else from global memory.  The LDS pointers carry their
// address space: through generic pointers every access would be a flat load that waits for both memory counters.
struct StepQuads
{
    const HYAS_LDS Quad *lds; // valid when local
    const Quad *glb;
    volatile HYAS_LDS int *flag; // workgroup flag "a sample of this shot collides"
    bool local, flagged;         // flagged == false: every verdict is wanted
    __device__ __forceinline__ Quad operator()(uint32_t i) const
    {
        if (!local)
            return glb[i];
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const HYAS_LDS u32x4 *p = (const HYAS_LDS u32x4 *)(lds + i);
        const u32x4 a = p[0], b = p[1], c = p[2], d = p[3];
        Quad q;
        q.vx[0] = __uint_as_float(a.x), q.vx[1] = __uint_as_float(a.y), q.vx[2] = __uint_as_float(a.z), q.vx[3] = __uint_as_float(a.w);
        q.vy[0] = __uint_as_float(b.x), q.vy[1] = __uint_as_float(b.y), q.vy[2] = __uint_as_float(b.z), q.vy[3] = __uint_as_float(b.w);
        q.xmin = __uint_as_float(c.x), q.ymin = __uint_as_float(c.y), q.xmax = __uint_as_float(c.z), q.ymax = __uint_as_float(c.w);
        q.axis_rect = d.x, q.pad[0] = q.pad[1] = q.pad[2] = 0;
        return q;
    }
    __device__ __forceinline__ bool stop() const { return flagged && *flag != 0; }
    __device__ __forceinline__ void hit() const
    {
        if (flagged)
            *flag = 1;
    }
    // Eight list entries per 16-byte load, the following eight requested one block ahead, so that a walk waits for
    // memory once per list and not once per entry.  Reads up to 32 bytes past the last entry: the index arrays are
    // allocated with that much slack and start on a 16-byte boundary.
    struct Cursor
    {
        const uint4 *base;
        uint4 cur, nxt;
        uint32_t blk;
        __device__ __forceinline__ Cursor(const uint16_t *idx, uint32_t begin) : base(reinterpret_cast<const uint4 *>(idx)), blk(begin >> 3)
        {
            cur = base[blk];
            nxt = base[blk + 1];
        }
        __device__ __forceinline__ uint16_t at(uint32_t k)
        {
            const uint32_t b = k >> 3;
            if (b != blk)
            {
                cur = nxt;
                nxt = base[b + 1];
                blk = b;
            }
            const uint32_t j = k & 7u;
            const uint32_t lo = (j & 2u) ? cur.y : cur.x, hi = (j & 2u) ? cur.w : cur.z;
            const uint32_t word = (j & 4u) ? hi : lo;
            return (uint16_t)((j & 1u) ? (word >> 16) : (word & 0xffffu));
        }
    };
};
#ifndef STEP_MIN_WAVES
#define STEP_MIN_WAVES 4 // waves per SIMD the register allocation must allow: three workgroups per CU
#endif


// rs::build_prefix with the sines and cosines of the six headings given (sn[i], cs[i] in front of segment i).
__device__ __forceinline__ void prefix_from_sincos(const rs::Path &path, double phi0, const double *sn, const double *cs, rs::Prefix &P)
{
    double phi = phi0;
    P.x[0] = 0.0, P.y[0] = 0.0;
    for (int i = 0; i < 5; ++i)
    {
        const int kind = rs::seg_kind(path.type, i);
        const double v = path.len[i];
        double x = P.x[i], y = P.y[i];
        P.phi[i] = phi, P.s[i] = sn[i], P.c[i] = cs[i];
        if (kind == 1)
        {
            x += (sn[i + 1] - sn[i]);
            y += (-cs[i + 1] + cs[i]);
            phi = phi + v;
        }
        else if (kind == 3)
        {
            x += (-sn[i + 1] + sn[i]);
            y += (cs[i + 1] - cs[i]);
            phi = phi - v;
        }
        else if (kind == 2)
        {
            x += (v * cs[i]);
            y += (v * sn[i]);
        }
        P.x[i + 1] = x, P.y[i + 1] = y;
    }
    P.phi[5] = phi;
}

// Obstacle-list entries the reference walks for one (pose, predecessor) pair:
// the list of the cell under polygon[0] plus the lists of both segment ends.
__device__ __forceinline__ uint32_t lists_walked(const DevMap &m, float x, float y, float t, float px, float py)
{
    const uint32_t cells = m.w * m.h;
    uint32_t n = 0;
    F2 c[4];
    vehicle_corners(m, x, y, t, c);
    uint32_t k0 = grid_key(m, c[0].x, c[0].y), ks = grid_key(m, x, y), ke = grid_key(m, px, py);
    if (on_grid(m, c[0].x, c[0].y) && k0 < cells)
        n += m.cell_off[k0 + 1] - m.cell_off[k0];
    if (on_grid(m, x, y) && ks < cells)
        n += m.cell_off[ks + 1] - m.cell_off[ks];
    if (on_grid(m, px, py) && ke < cells)
        n += m.cell_off[ke + 1] - m.cell_off[ke];
    return n;
}

// Motion-primitive successors of n popped nodes with verdict, g, h and index
// (src/hybrid_a_star.cpp:540-811, :316-366).  Sixteen nodes per workgroup; for
// each node eight lanes of waves 0-1 run the pose tests of its primitives and
// eight lanes of waves 2-3 the swept-segment tests and, for survivors, cost
// and heuristic.
// MODE 0: plain; 1: work counters (stats[3..6]).
// 256 threads: with several nodes per query in a launch the chip is busy, and four-wave workgroups pack better than
// eight-wave ones (512 threads measured 3 % slower on 1024 maze queries, 15 % slower on 8192 large-map queries)
#ifndef ROUND_THREADS
#define ROUND_THREADS 256
#endif
#define EXPAND_THREADS ROUND_THREADS
#define SHOT_THREADS ROUND_THREADS
// samples of a curve interpolated per pass and then tested in rounds of SHOT_THREADS / 2 (half of the workgroup runs
// the pose tests, half the swept-segment tests).  Passes of SHOT_THREADS / 2 samples would waste less work behind the
// first colliding sample but measured slower (61 vs 54 us per launch on the 1024-query maze batch): the launch waits
// for its longest curve, and that one pays a barrier pair per pass.
#ifndef SHOT_BATCH
#define SHOT_BATCH SHOT_THREADS
#endif

// workgroup-local copy of the obstacle quads when they fit (four 16-byte words each); the caller synchronises.
// Threads [first, ROUND_THREADS) copy: the shot role keeps the wave that solves the curve out of it.  All loads of a
// thread are requested before the first store, so the copy costs one trip to memory, not one per word.
__device__ __forceinline__ StepQuads stage_quads(const DevMap &m, Quad *s_quads, int tid, int first = 0, int count = ROUND_THREADS)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const bool quads_local = m.n_quads <= STEP_LDS_QUADS;
    const int t = tid - first;
    if (quads_local && t >= 0 && t < count)
    {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(m.quads);
        HYAS_LDS u32x4 *dst = (HYAS_LDS u32x4 *)s_quads;
        const uint32_t words = m.n_quads * 4u;
        constexpr int PER = (STEP_LDS_QUADS * 4 + ROUND_THREADS - 64 - 1) / (ROUND_THREADS - 64); // enough for three waves
        u32x4 w[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j)
        {
            const uint32_t i = (uint32_t)t + (uint32_t)j * (uint32_t)count;
            if (i < words)
                w[j] = src[i];
        }
#pragma unroll
        for (int j = 0; j < PER; ++j)
        {
            const uint32_t i = (uint32_t)t + (uint32_t)j * (uint32_t)count;
            if (i < words)
                dst[i] = w[j];
        }
    }
    return StepQuads{(const HYAS_LDS Quad *)s_quads, m.quads, nullptr, quads_local, false};
}

// ADAPT: 0 fixed motion set, 1 adaptive steering angle and step size, 2 adaptive step size (hyas_adaptive.h).
// With ADAPT == 1 a node has up to HYAS_AD_MAX_PAIRS + 1 candidates: sixteen candidate lanes per node and test, two nodes
// per workgroup; the first lane of a node forms the node's {step, steering} list in workgroup memory, then the lanes take
// one pair each, sixteen pairs per pass.  count = 0xFF marks a node whose list did not fit HYAS_AD_MAX_PAIRS.
// Four lanes (eight with adaptive step size) share each candidate's test: a wave lasts as long as the longest obstacle
// list among its lanes, and the lists are walked entry by entry, so each of the lanes takes every fourth (eighth)
// entry (the verdict is an AND over the list).  Without list walks the successor kernel takes 19 us, with them
// 29-37 us undivided (fixed motion set; eight lanes gain nothing over four there); the adaptive modes' steps of
// several metres read the unpruned lists of two cells (round kernel 70 -> 53 us and 59 -> 39 us).  A node's lanes of
// one kind stay within a wave.
template <int ADAPT>
struct ExpandShape
{
    static constexpr int CAND = ADAPT == 1 ? 16 : 8;                // candidate slots per node
    static constexpr int SUB = ADAPT == 2 ? 8 : 4;                  // lanes that share one candidate's list walk
    static constexpr int LANES = CAND * SUB;                        // lanes per node and kind of test
    static constexpr int NODES = EXPAND_THREADS / 2 / LANES;        // nodes per workgroup
};

template <int MODE, int ADAPT>
__device__ __forceinline__ void expand_role(const DevMap &m, const StepIn *__restrict__ in, uint32_t n, const hyas_goal *__restrict__ goals,
                                            const float *__restrict__ twod, uint8_t *__restrict__ count, hyas_succ *__restrict__ succ,
                                            unsigned long long *stats, Quad *s_quads, uint32_t block, StepIn *__restrict__ child = nullptr,
                                            uint32_t flags = 0, const int32_t *__restrict__ bound = nullptr, uint32_t bound_cells = 0,
                                            RoundOut ro = RoundOut{nullptr, 0, 0})
{
    constexpr int LANES = ExpandShape<ADAPT>::LANES, NODES = ExpandShape<ADAPT>::NODES, CAND = ExpandShape<ADAPT>::CAND, SUB = ExpandShape<ADAPT>::SUB;
    const int tid = threadIdx.x;
    const StepQuads sq = stage_quads(m, s_quads, tid);
    __shared__ uint8_t s_pose[EXPAND_THREADS / 2];
    __shared__ float s_step[ADAPT == 1 ? NODES : 1][HYAS_AD_MAX_PAIRS], s_steer[ADAPT == 1 ? NODES : 1][HYAS_AD_MAX_PAIRS];
    __shared__ int s_pairs[ADAPT == 1 ? NODES : 1]; // pairs of the node, -1: too many
    // LANES lanes per node and test; waves 0-1 run the pose tests, waves 2-3 the swept-segment tests
    // (a wave that mixed the two would run them one after the other)
    // a node's lanes: candidate sl, part `part` of its list walk (part 0 lanes of a node lie side by side)
    const int lane = tid & 63, sl = tid & (CAND - 1), part = (tid & (LANES - 1)) / CAND, local = (tid & (EXPAND_THREADS / 2 - 1)) / LANES;
    const int nl = tid & (LANES - 1); // lane within the node's lanes of this kind
    const uint32_t slot = block * NODES + local;
    const bool in_launch = slot < n;
    const StepIn node = in[in_launch ? slot : 0];
    const bool live = in_launch && (node.shoot & STEP_DEAD) == 0;
    const hyas_goal goal = goals[node.query];
    const float *my_twod = twod ? twod + (size_t)node.query * m.twod_size : nullptr;
    const bool pose_lane = tid < EXPAND_THREADS / 2, seg_lane = !pose_lane;
    const float dist = normf(sub(f2(node.x, node.y), f2(goal.x, goal.y)));
    // Adaptive modes: the angle ranges of the node's cell are walked once per candidate, every step depending on the
    // one before: from global memory that is a chain of some eighty cache round trips.  The node's lanes fetch the
    // ranges side by side into workgroup memory first (cells with more than AD_LDS_RANGES keep to global memory).
    constexpr int AD_LDS_RANGES = 16;
    __shared__ ad::Range s_rng[ADAPT != 0 ? NODES : 1][AD_LDS_RANGES];
    ad::Cell cell;
    cell.rng = nullptr, cell.n = 0, cell.density = 0;
    if (ADAPT != 0 && live)
    {
        cell = ad::cell_of(m, node.x, node.y);
        if (pose_lane && cell.n <= (uint32_t)AD_LDS_RANGES)
            for (uint32_t i = (uint32_t)nl; i < cell.n; i += LANES)
                s_rng[local][i] = cell.rng[i];
    }
    if (ADAPT != 0)
    {
        __syncthreads();
        if (live && cell.n <= (uint32_t)AD_LDS_RANGES)
            cell.rng = s_rng[local];
    }
    if constexpr (ADAPT == 1)
    {
        // The node's {step, steering} list.  Formed by one lane alone it is the longest chain of the launch (every
        // candidate walks all ranges, normalising their ends on the way), so the node's lanes share it out: lane i
        // takes range i -- its ends in normal form, then its candidates -- and the first lane only applies the
        // duplicate rule to what the others found, in range order.  (Not for the configurations and cells named
        // at ad::emit_range: the first lane then forms the list by itself.)
        static_assert(CAND >= AD_LDS_RANGES, "one lane per staged range");
        __shared__ ad::RangePre s_pre[NODES][AD_LDS_RANGES];
        __shared__ float s_estep[NODES][AD_LDS_RANGES * HYAS_AD_EMIT_MAX], s_esteer[NODES][AD_LDS_RANGES * HYAS_AD_EMIT_MAX];
        __shared__ int s_ecount[NODES][AD_LDS_RANGES];
        const bool shared_out = live && ad::staged_selection_applies(m) && cell.n <= (uint32_t)AD_LDS_RANGES &&
                                !(cell.density > m.ad_constant_density);
        const bool my_range = shared_out && pose_lane && part == 0 && (uint32_t)sl < cell.n;
        if (my_range)
            s_pre[local][sl] = ad::range_pre(cell.rng[sl]);
        __syncthreads();
        if (my_range)
        {
            const float toward = ad::toward_goal_steering(m, node.x, node.y, node.t, goal.x, goal.y, goal.t, __uint_as_float(node.pad));
            s_ecount[local][sl] = ad::emit_range(m, cell, s_pre[local], (uint32_t)sl, node.t, dist, toward, m.res_f,
                                                 &s_estep[local][sl * HYAS_AD_EMIT_MAX], &s_esteer[local][sl * HYAS_AD_EMIT_MAX]);
        }
        __syncthreads();
        if (live && pose_lane && sl == 0 && part == 0)
        {
            ad::Pairs p;
            p.step = s_step[local], p.steer = s_steer[local];
            if (shared_out)
                ad::merge_emitted(cell, s_ecount[local], s_estep[local], s_esteer[local], p);
            else
                ad::step_steer_pairs(m, cell, node.x, node.y, node.t, goal.x, goal.y, goal.t, __uint_as_float(node.pad), p);
            s_pairs[local] = p.overflow ? -1 : p.n;
        }
    }
    // With adaptive steering a node may have more candidates than it has lanes of a kind: they are taken CAND at a time
    // (one pass is the rule: the reference's default switches give at most 7, the 5-degree fans of
    // fixed_number_of_steering_angle_in_free_angle_range = 0 up to about 20).  Every workgroup makes as many passes as
    // its longest list needs.
    __shared__ int s_passes;
    if (tid == 0)
        s_passes = 1;
    __syncthreads(); // quads staged, pair lists formed
    int np = 0;      // the node's pairs (adaptive steering), the reverse pair not counted
    bool overflow = false;
    const bool reverse_pair = ADAPT == 1 && m.succ_reverse[m.n_succ - 1] != 0;
    if (ADAPT == 1)
    {
        np = s_pairs[local];
        overflow = np < 0;
        const int total = (!live || overflow) ? 0 : np + (reverse_pair ? 1 : 0);
        if (nl == 0 && pose_lane && total > CAND)
            atomicMax(&s_passes, (total + CAND - 1) / CAND);
        __syncthreads();
    }
    const int n_passes = ADAPT == 1 ? s_passes : 1;
    const unsigned grp = (lane / LANES) * LANES; // the node's part 0 lanes: grp .. grp + CAND - 1
    constexpr unsigned long long GROUP = (1ull << CAND) - 1ull;
    const bool writer = in_launch && seg_lane && sl == 0 && part == 0;
    int emitted = 0, verdicts = 0; // of the node, over the passes so far
    uint32_t walked_all = 0;
    for (int pass = 0; pass < n_passes; ++pass)
    {
        const int ci = pass * CAND + sl; // this lane's candidate
        bool part_ok = false, evaluated = false;
        float sx = 0, sy = 0, st = 0;
        int prim = 0;
        float s_g = 0, beat = INFINITY; // the successor's cost, and what it has to beat to be of use to the search
        int s_idx = 0;
        if (live)
        {
            // this lane's candidate: {step, steering angle}
            bool have = false;
            float first = 0, steer = 0;
            if (ADAPT == 1)
            {
                if (!overflow && ci < np)
                    have = true, first = s_step[local][ci], steer = s_steer[local][ci];
                else if (!overflow && ci == np && reverse_pair) // the reverse pair follows the node's list (src/hybrid_a_star.cpp:629-654)
                    have = true, first = motion_step(true, dist), steer = m.succ_steer[m.n_succ - 1];
            }
            else if (sl < m.n_succ)
            {
                have = true, steer = m.succ_steer[sl];
                const bool rev = m.succ_reverse[sl] != 0;
                if (ADAPT == 2 && !rev) // src/hybrid_a_star.cpp:585-599
                    first = ad::find_step_size(m, cell, node.t, steer, dist, m.res_f);
                else
                    first = motion_step(rev, dist);
            }
            if (have && successor_from_pair(node.x, node.y, node.t, first, steer, sx, sy, st, prim))
            {
                evaluated = true;
                if (seg_lane && part == 0)
                {
                    // cost and cell of the lane that writes the record; the cell's bound is requested here and used
                    // after the list walks
                    s_g = cost_so_far(m, node.g, normf(sub(f2(sx, sy), f2(node.x, node.y))), prim, node.prim);
                    s_idx = idx_3d(m, sx, sy, st);
                    if (bound && s_idx >= 0 && (uint32_t)s_idx < bound_cells)
                        beat = __int_as_float(__builtin_nontemporal_load(bound + (size_t)node.query * bound_cells + (uint32_t)s_idx));
                }
                if (pose_lane)
                    part_ok = on_grid(m, sx, sy) && pose_free(m, sq, sx, sy, st, (uint32_t)part, (uint32_t)SUB);
                else
                    part_ok = segment_free(m, sq, sx, sy, node.x, node.y, (uint32_t)part, (uint32_t)SUB);
            }
        }
        if (SUB > 1) // a candidate is free iff every part of its list says so
        {
#pragma unroll
            for (int o = CAND; o < LANES; o <<= 1)
                part_ok = (__shfl_xor(part_ok ? 1 : 0, o, 64) != 0) && part_ok;
        }
        if (pass)
            __syncthreads(); // the pass before has read its pose verdicts
        if (pose_lane)
            s_pose[tid] = part_ok ? 1 : 0;
        __syncthreads();
        // (s_g >= beat is false for a NaN cost: such a successor is left to the host)
        const bool ok = seg_lane && part == 0 && evaluated && part_ok && s_pose[tid & (EXPAND_THREADS / 2 - 1)] != 0 && !(s_g >= beat);
        hyas_succ s;
        if (ok)
        {
            s.x = sx, s.y = sy, s.t = st;
            s.g = s_g;
            s.h = heuristic(m, sx, sy, st, goal.x, goal.y, goal.t, my_twod);
            s.idx = s_idx;
            s.prim = (int8_t)prim;
            s.free = 1;
            // search rounds: which entry of the motion set this successor comes from (1-based, high byte; fixed motion set)
            s.pad = (flags & ROUND_TAG_CANDIDATE) ? (uint16_t)(((ci + 1) & 31) << 8) : (uint16_t)0;
        }
        // compact in candidate order, as the reference's output vector
        const unsigned mask = (unsigned)((__ballot(ok) >> grp) & GROUP);
        const unsigned emask = (unsigned)((__ballot(seg_lane && evaluated) >> grp) & GROUP);
        if (child && pass == 0)
        {
            // The successor with the lowest f = g + h (the earliest of equals: the one the open list would pop first; of
            // the first pass, where there are several) becomes node `slot` of the next, deeper launch; its record carries
            // bit 0 of pad so that the host knows which one was taken without redoing the comparison.
            float bf = ok ? s.g + s.h : INFINITY;
            if (!(bf == bf))
                bf = INFINITY;
            int bi = ok ? sl : 255;
            for (int off = CAND / 2; off > 0; off >>= 1)
            {
                const float of = __shfl_xor(bf, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (of < bf || (of == bf && oi < bi))
                    bf = of, bi = oi;
            }
            if (ok && bi == sl)
            {
                s.pad |= 1;
                child[slot] = StepIn{s.x, s.y, s.t, s.g, prim, node.query, flags & ROUND_CHILD_SHOOT, node.pad};
            }
            if (in_launch && seg_lane && sl == 0 && part == 0 && bi == 255) // nothing to expand one level down
                child[slot] = StepIn{node.x, node.y, node.t, node.g, node.prim, node.query, STEP_DEAD, node.pad};
        }
        if (ok)
        {
            const int pos = emitted + __popc(mask & ((1u << sl) - 1u));
            if (ro.base)
            {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                u32x4 *unit = reinterpret_cast<u32x4 *>(ro.base) + (size_t)slot * ro.units + 2 + 2 * pos;
                const u32x4 a = {__float_as_uint(s.x), __float_as_uint(s.y), __float_as_uint(s.t),
                                 ro.tag << 9 | (uint32_t)(s.pad & 1u) << 8 | (uint32_t)(s.pad >> 8) << 3 | (uint32_t)prim};
                const u32x4 b = {__float_as_uint(s.g), __float_as_uint(s.h), (uint32_t)s.idx, ro.tag};
                __builtin_nontemporal_store(a, unit);
                __builtin_nontemporal_store(b, unit + 1);
            }
            else
                succ[(size_t)slot * m.n_slots + pos] = s;
        }
        emitted += __popc(mask), verdicts += __popc(emask);
        if (MODE == 1)
        {
            uint32_t walked = (seg_lane && evaluated && part == 0) ? lists_walked(m, sx, sy, st, node.x, node.y) : 0;
            for (int off = CAND / 2; off > 0; off >>= 1)
                walked += __shfl_xor(walked, off, 64);
            walked_all += walked;
        }
    }
    if (writer) // successors emitted, candidates that got a verdict; 0xFF: more pairs than a node's list holds
    {
        if (ro.base)
        {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 e = {overflow ? 0xFFu : (uint32_t)emitted | (uint32_t)verdicts << 8, 0u, 0u, ro.tag};
            __builtin_nontemporal_store(e, reinterpret_cast<u32x4 *>(ro.base) + (size_t)slot * ro.units + 1);
        }
        else
            count[slot] = overflow ? (uint8_t)0xFF : (uint8_t)emitted;
    }
    if (MODE == 1 && writer && live)
    {
        atomicAdd(&stats[3], (unsigned long long)verdicts);
        atomicAdd(&stats[4], (unsigned long long)walked_all);
        atomicAdd(&stats[5], (unsigned long long)emitted);
        atomicAdd(&stats[6], 1ull);
    }
}

// Analytic Reeds-Shepp shot of one popped node (src/hybrid_a_star.cpp:429-468),
// one workgroup of four waves per node.
//  phase 1: the 44 closed-form candidates, one per lane of wave 0;
//  phase 2: the same wave picks the winner in reference order and builds the
//           per-segment prefix of the winning curve;
//  phase 3: all lanes interpolate 256 samples of the curve at a time; the
//           samples are then tested 128 at a time, pose tests on waves 0-1 and
//           swept-segment tests on waves 2-3, until one collides.
// stats[] (MODE 1): 0 samples tested, 1 samples up to and including the first
// colliding one (what a serial scan tests), 2 obstacle-list entries behind (1),
// 7 shots; (MODE 2) shader-clock sums: 8 phases 1+2, 9 interpolation,
// 10 collision tests, 11..14 time from the start to: frame normalised, candidates evaluated, winner known, prefix
// stored; 15 arrival of node and goal, 16 whole workgroup, 17 / 18 tests of wave 0 (pose tests in even rounds) / wave 2,
// 6 workgroups, 19 sample passes.
template <int MODE, int CURVE>
__device__ __forceinline__ void shot_role(const DevMap &m, const StepIn *__restrict__ in, const hyas_goal *__restrict__ goals,
                                          const double *__restrict__ seg_tab, uint32_t seg_cap, uint8_t *__restrict__ shot_ok,
                                          float *__restrict__ shot_len, unsigned long long *stats, Quad *s_quads, uint32_t slot,
                                          RoundOut ro = RoundOut{nullptr, 0, 0})
{
    // the verdict leaves in one store: into the arrays, or into unit 0 of the node's block
    auto answer = [&](uint8_t ok, float len) {
        if (ro.base)
        {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = {__float_as_uint(len), (uint32_t)ok, 0u, ro.tag};
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(ro.base) + (size_t)slot * ro.units);
        }
        else
            shot_ok[slot] = ok, shot_len[slot] = len;
    };
    __shared__ rs::Path s_path;
    __shared__ rs::Prefix s_prefix;
    __shared__ db::Path s_dubins;
    __shared__ float s_xyt[SHOT_BATCH + 1][3]; // [0] = the sample in front of this batch
    __shared__ volatile int s_hit;
    __shared__ unsigned int s_first, s_lists;
    __shared__ bz::Curve s_bezier;          // cubic Bezier only
    __shared__ float s_bezier_len;          //
    __shared__ uint8_t s_sharp[SHOT_BATCH]; // sample's curvature is above 1 / min_turning_radius
    constexpr bool RS = CURVE == HYAS_CURVE_RS, BZ = CURVE == HYAS_CURVE_BEZIER;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const StepIn node = in[slot];
    if ((node.shoot & 1u) == 0) // uniform across the workgroup
    {
        if (tid == 0)
            answer(0, 0.f);
        return;
    }
    constexpr bool STATS = MODE == 1, CLOCKS = MODE == 2;
    // clock sums are kept in registers and added to stats[] once, at the end: atomics inside the phases would be
    // waited for at the next barrier and lengthen what they measure
    long long c0 = 0, c1 = 0, c2 = 0, t_wave = 0, t_solve = 0, t_interp = 0, t_collide = 0, t_load = 0, t_test = 0, mk0 = 0, mk1 = 0, mk2 = 0, mk3 = 0;
    unsigned n_batches = 0;
    if (CLOCKS)
        c0 = clock64();
    const hyas_goal goal = goals[node.query];
    const double q0[3] = {(double)node.x, (double)node.y, (double)node.t};
    if (CLOCKS)
    {
        // the node comes from host memory, the goal from device memory: time until both have arrived
        if (__float_as_uint(goal.x) + __float_as_uint(node.x) != 0x12345u)
            t_load = clock64() - c0;
    }
    // the quads are copied by the waves that do not solve the curve; visible after the barrier that ends phase 2
    constexpr int N_WAVES = SHOT_THREADS / 64;
    const int solver = RS ? (int)(slot % N_WAVES) : 0;
    StepQuads sq = stage_quads(m, s_quads, (tid + (N_WAVES - 1 - solver) * 64) % SHOT_THREADS, 0, SHOT_THREADS - 64);
    if (!STATS) // one colliding sample decides the shot: the other lanes stop walking their lists
        sq.flag = (volatile HYAS_LDS int *)&s_hit, sq.flagged = true;

    // ---- phase 1 ------------------------------------------------------------
    if (BZ)
    {
        // Cubic Bezier (src/hybrid_a_star.cpp:470-476): the curve, then CubicBezier::GetLength: 101 polyline vertices
        // and 100 segment norms side by side (staged in the sample buffer), their float sum in order on one lane
        static_assert(SHOT_BATCH >= HYAS_BZ_INTERVALS + 1 && SHOT_THREADS >= HYAS_BZ_INTERVALS + 1, "polyline staged in the sample buffer");
        const float a[3] = {node.x, node.y, node.t}, b[3] = {goal.x, goal.y, goal.t};
        const bz::Curve c = bz::make(a, b); // every lane: a few dozen float operations, no broadcast needed
        if (tid <= HYAS_BZ_INTERVALS)
            bz::value(c, bz::powers(bz::length_param((unsigned)tid)), s_xyt[tid + 1][0], s_xyt[tid + 1][1]);
        __syncthreads();
        if (tid < HYAS_BZ_INTERVALS)
        {
            const float dx = s_xyt[tid + 2][0] - s_xyt[tid + 1][0], dy = s_xyt[tid + 2][1] - s_xyt[tid + 1][1];
            s_xyt[tid + 1][2] = hy_sqrtf(dx * dx + dy * dy);
        }
        __syncthreads();
        if (tid == 0)
        {
            float len = 0;
            for (int i = 0; i < HYAS_BZ_INTERVALS; ++i)
                len += s_xyt[i + 1][2];
            s_bezier = c, s_bezier_len = len;
            s_hit = 0;
            s_xyt[0][0] = node.x, s_xyt[0][1] = node.y, s_xyt[0][2] = node.t;
        }
    }
    else if (!RS)
    {
        // Dubins: six float words, cheap next to the sample tests; one lane solves (src/hybrid_a_star.cpp:378-395)
        if (tid == 0)
        {
            const float a[3] = {node.x, node.y, node.t}, b[3] = {goal.x, goal.y, goal.t};
            db::Path p;
            p.qi[0] = a[0], p.qi[1] = a[1], p.qi[2] = a[2];
            p.param[0] = p.param[1] = p.param[2] = 0, p.type = 0, p.rho = m.shot_rho;
            (void)db::init(a, b, m.shot_rho, p);
            s_dubins = p;
            s_hit = 0;
            s_xyt[0][0] = node.x, s_xyt[0][1] = node.y, s_xyt[0][2] = node.t;
        }
    }
    else if (wave == solver)
    {
        // (which wave: the waves of a workgroup sit on different SIMDs, and workgroups that start together on one
        // compute unit should not all keep the same SIMD busy while the other three idle)
        // Phases 1 and 2 on one wave, nothing leaves its registers in between: lane c evaluates candidate c in the
        // side-by-side form (the time of the longest closed form, not the sum of all), the wave walks the 44 results
        // in reference order on uniform values, six lanes take sine and cosine of the heading in front of each
        // segment, and lane 0 stores the curve and its per-segment prefix states.
        const double q1[3] = {(double)goal.x, (double)goal.y, (double)goal.t};
        double x, y, phi;
        rs::normalise(q0, q1, (double)m.shot_rho, x, y, phi);
        rs::Path mine = rs::empty_path();
        double my_L = __builtin_huge_val();
        if (CLOCKS)
            mk0 = clock64() - c0; // goal loaded, frame normalised
        if (lane < RS_NUM_CANDIDATES)
        {
            double L = 0;
            if (rs::candidate_staged(lane, x, y, phi, mine, L) && L == L)
                my_L = L; // an invalid or NaN length never wins: lmin > L is false for +inf
        }
        if (CLOCKS)
            t_wave = mk1 = clock64() - c0; // candidates evaluated
        // The reference scans a family with a running minimum that starts at the family's bound and is replaced by
        // strictly shorter candidates only: the family's entry is the first candidate that attains the family's
        // minimum, and it is taken iff that minimum is below the bound.  The five minima do not depend on one another
        // (five short chains instead of one of 44 steps); the bounds then chain through the families in order.
        constexpr int first[RS_NUM_FAMILIES + 1] = {0, 8, 16, 24, 40, 44}; // rs::family_first
        double fam_min[RS_NUM_FAMILIES];
        int fam_arg[RS_NUM_FAMILIES];
#pragma unroll
        for (int fam = 0; fam < RS_NUM_FAMILIES; ++fam)
        {
            double mn = lane_value(my_L, first[fam]);
            int arg = first[fam];
#pragma unroll
            for (int c = first[fam] + 1; c < first[fam + 1]; ++c)
            {
                const double L = lane_value(my_L, c);
                if (mn > L)
                    mn = L, arg = c;
            }
            fam_min[fam] = mn, fam_arg[fam] = arg;
        }
        int win = -1;
        double best_total = rs::empty_path().total;
#pragma unroll
        for (int fam = 0; fam < RS_NUM_FAMILIES; ++fam)
            if (rs::family_bound(fam, best_total) > fam_min[fam])
                win = fam_arg[fam], best_total = lane_value(mine.total, win);
        rs::Path best = rs::empty_path();
        if (win >= 0) // uniform
        {
            best.type = __builtin_amdgcn_readlane(mine.type, win);
            for (int i = 0; i < 5; ++i)
                best.len[i] = lane_value(mine.len[i], win);
            best.total = lane_value(mine.total, win);
        }
        if (CLOCKS)
            mk2 = clock64() - c0; // winner known
        double my_phi = q0[2], ph = q0[2];
        for (int i = 0; i < 5; ++i)
        {
            const int kind = rs::seg_kind(best.type, i);
            ph = (kind == 1) ? ph + best.len[i] : (kind == 3) ? ph - best.len[i] : ph;
            my_phi = (lane == i + 1) ? ph : my_phi;
        }
        double my_s, my_c;
        hy_sincos(my_phi, &my_s, &my_c);
        double sn[6], cs[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
            sn[i] = lane_value(my_s, i), cs[i] = lane_value(my_c, i);
        if (lane == 0)
        {
            rs::Prefix pf;
            prefix_from_sincos(best, q0[2], sn, cs, pf);
            s_path = best;
            s_prefix = pf;
            s_xyt[0][0] = node.x, s_xyt[0][1] = node.y, s_xyt[0][2] = node.t; // predecessor of sample 0: the node
            s_hit = 0;
        }
        if (CLOCKS)
            mk3 = clock64() - c0; // prefix stored
    }
    __syncthreads();
    if (CLOCKS)
    {
        c1 = clock64();
        t_solve = c1 - c0;
    }

    // ---- phase 3 ------------------------------------------------------------
    const rs::Path &path = s_path; // read in place: a private copy indexed by segment would live in scratch memory
    db::Path dpath;
    const double rho = (double)m.shot_rho;
    double length;
    uint32_t nsamp;
    if (RS)
    {
        length = rho * path.total; // src/ReedsShepp.cpp:564
        nsamp = count_samples(seg_tab, seg_cap, length);
    }
    else if (BZ)
    {
        length = (double)s_bezier_len;
        nsamp = count_samples_db(m.seg_bz, m.seg_bz_cap, s_bezier_len); // x = 0; while (x < length) x += 0.1 * resolution
    }
    else
    {
        dpath = s_dubins;
        const float lf = db::length(dpath);
        length = (double)lf;
        nsamp = count_samples_db(m.seg_db, m.seg_db_cap, lf);
    }
    const float sharpest = 1 / m.shot_rho; // src/hybrid_a_star.cpp:496
    bool collided = false;
    unsigned long long tested = 0;
    // sample k of the curve, as the reference's loop produces it
    auto sample_at = [&](uint32_t k, float &x, float &y, float &t, bool &sharp) {
        if (RS)
        {
            double s[3];
            rs::interpolate_prefix(q0, path, s_prefix, seg_tab[k] / rho, rho, s);
            x = (float)s[0], y = (float)s[1], t = rad_0_2pi((float)s[2]);
        }
        else if (BZ)
        {
            // src/hybrid_a_star.cpp:481-483: value, curvature and heading at x / length
            const bz::Powers pw = bz::powers(m.seg_bz[k] / s_bezier_len);
            float angle, curvature;
            bz::value(s_bezier, pw, x, y);
            bz::angle_curvature(s_bezier, pw, angle, curvature);
            t = rad_0_2pi(angle);
            sharp = !(curvature <= sharpest);
        }
        else
        {
            float q[3] = {0, 0, 0};
            (void)db::sample(dpath, m.seg_db[k], q);
            x = q[0], y = q[1], t = rad_0_2pi(q[2]);
        }
    };
    // The shot fails when ANY of its samples collides (src/hybrid_a_star.cpp:441-447, :409-415, :500-510), so the
    // order in which they are looked at is free, and a sample whose centre cell is occupied collides: every batch first
    // reads one cost byte per sample.  Most shots through an obstacle end there, before a single polygon is tested.
    // (The counting kernel keeps to the serial order its figures are defined by.)
    // Curves longer than one batch: batch b takes samples b, b + nb, b + 2 nb, ... (nb batches), so that every batch
    // covers the whole curve.  A collision is a stretch of the curve, not a single sample, so the first batch nearly
    // always meets it, mostly with its cost bytes alone; taken in order, a curve that is free for its first 25 m
    // would run the exact tests of a whole batch for nothing.  A sample's predecessor is the lane's own sample of the
    // batch before; the first batch forms its predecessors itself.
    const bool strided = !STATS && nsamp > SHOT_BATCH; // uniform
    if (strided && !collided)
    {
        __shared__ float s_pxy[SHOT_BATCH][2];
        const uint32_t nb = (nsamp + SHOT_BATCH - 1) / SHOT_BATCH;
        const bool sampler = tid < SHOT_BATCH;
        float prev_x = 0, prev_y = 0;
        for (uint32_t b = 0; b < nb && !collided; ++b)
        {
            const uint32_t k = (uint32_t)tid * nb + b;
            float x = 0, y = 0, t = 0;
            bool sharp = false, blocked = false;
            if (sampler && k < nsamp)
            {
                sample_at(k, x, y, t, sharp);
                blocked = centre_blocked(m, x, y) || (BZ && sharp);
            }
            if (sampler)
            {
                s_xyt[tid + 1][0] = x, s_xyt[tid + 1][1] = y, s_xyt[tid + 1][2] = t;
                s_pxy[tid][0] = prev_x, s_pxy[tid][1] = prev_y;
                if (BZ)
                    s_sharp[tid] = sharp ? 1 : 0;
            }
            if (__syncthreads_or(blocked ? 1 : 0))
            {
                collided = true;
                break;
            }
            if (CLOCKS)
                ++n_batches;
            if (b == 0)
            {
                float qx = node.x, qy = node.y, qt = 0;
                bool qs = false;
                if (sampler && k >= 1 && k < nsamp)
                    sample_at(k - 1, qx, qy, qt, qs);
                if (sampler)
                    s_pxy[tid][0] = qx, s_pxy[tid][1] = qy;
                __syncthreads();
            }
            for (int half = 0; half < SHOT_BATCH / (SHOT_THREADS / 2); ++half)
            {
                const uint32_t j = half * (SHOT_THREADS / 2) + (tid & (SHOT_THREADS / 2 - 1)); // sample inside the batch
                const uint32_t kk = j * nb + b;
                const bool pose_lane = (tid < SHOT_THREADS / 2) != ((half & 1) != 0);
                if ((uint32_t)(half * (SHOT_THREADS / 2)) * nb + b >= nsamp)
                    break;
                bool hit = false;
                const float jx = s_xyt[j + 1][0], jy = s_xyt[j + 1][1], jt = s_xyt[j + 1][2];
                const float px = s_pxy[j][0], py = s_pxy[j][1];
                if (kk < nsamp && s_hit == 0)
                {
                    if (pose_lane)
                        hit = !(on_grid(m, jx, jy) && pose_free(m, sq, jx, jy, jt)) || (BZ && s_sharp[j] != 0);
                    else
                        hit = !segment_free(m, sq, jx, jy, px, py);
                }
                if (hit)
                    s_hit = 1;
            }
            __syncthreads();
            collided = s_hit != 0;
            prev_x = x, prev_y = y;
        }
    }
    for (uint32_t base = 0; base < nsamp && !collided && !strided; base += SHOT_BATCH)
    {
        if (base)
        {
            __syncthreads(); // everyone is done with the previous batch
            if (tid == 0)
                s_xyt[0][0] = s_xyt[SHOT_BATCH][0], s_xyt[0][1] = s_xyt[SHOT_BATCH][1];
        }
        bool blocked = false;
        {
            const uint32_t k = base + tid;
            float x = 0, y = 0, t = 0;
            bool sharp = false;
            if (tid < SHOT_BATCH && k < nsamp)
            {
                sample_at(k, x, y, t, sharp);
                blocked = !STATS && (centre_blocked(m, x, y) || (BZ && sharp));
            }
            if (base)
                __syncthreads(); // slot 0 has been refreshed before slot SHOT_BATCH is overwritten
            if (tid < SHOT_BATCH)
                s_xyt[tid + 1][0] = x, s_xyt[tid + 1][1] = y, s_xyt[tid + 1][2] = t;
            if (BZ && tid < SHOT_BATCH)
                s_sharp[tid] = sharp ? 1 : 0;
        }
        // the batch's own cost bytes first (the barrier that publishes the samples carries the answer)
        if (__syncthreads_or(blocked ? 1 : 0))
        {
            collided = true;
            if (CLOCKS)
                ++n_batches;
            break;
        }
        if (CLOCKS)
        {
            c2 = clock64();
            t_interp += c2 - c1, ++n_batches;
        }
        // two test rounds: one half of the workgroup runs the pose tests, the other half the swept-segment tests, and
        // the halves change roles for the second round (a swept-segment test takes about 1.4 times a pose test: both
        // halves then finish together); the workgroup meets once, after both (in between, the flag alone tells a
        // lane that the shot is decided)
        for (int half = 0; half < SHOT_BATCH / (SHOT_THREADS / 2); ++half)
        {
            const uint32_t j = half * (SHOT_THREADS / 2) + (tid & (SHOT_THREADS / 2 - 1)); // sample inside the batch
            const uint32_t k = base + j;
            const bool pose_lane = (tid < SHOT_THREADS / 2) != ((half & 1) != 0);
            if (base + half * (SHOT_THREADS / 2) >= nsamp)
                break;
            if (STATS && tid == 0)
                s_first = 0xFFFFFFFFu, s_lists = 0;
            if (STATS)
                __syncthreads();
            bool hit = false;
            const float x = s_xyt[j + 1][0], y = s_xyt[j + 1][1], t = s_xyt[j + 1][2];
            const float px = s_xyt[j][0], py = s_xyt[j][1];
            long long tt = 0;
            if (CLOCKS)
                tt = clock64();
            if (k < nsamp && (STATS || s_hit == 0))
            {
                if (pose_lane) // a Bezier sample also fails on its curvature (src/hybrid_a_star.cpp:496-510)
                    hit = !(on_grid(m, x, y) && pose_free(m, sq, x, y, t)) || (BZ && s_sharp[j] != 0);
                else
                    hit = !segment_free(m, sq, x, y, px, py);
                tested += pose_lane;
            }
            if (CLOCKS)
                t_test += clock64() - tt; // this wave's tests without the wait for the other waves
            if (hit)
                s_hit = 1;
            if (STATS)
            {
                if (hit)
                    atomicMin(&s_first, k);
                __syncthreads();
                const uint32_t first = s_first;
                if (k < nsamp && k <= first && pose_lane)
                    atomicAdd(&s_lists, lists_walked(m, x, y, t, px, py));
                __syncthreads();
                if (tid == 0)
                {
                    const uint32_t lo = base + half * (SHOT_THREADS / 2);
                    const uint32_t in_round = (nsamp - lo < SHOT_THREADS / 2) ? nsamp - lo : SHOT_THREADS / 2;
                    const uint32_t last = (first == 0xFFFFFFFFu) ? in_round : first - lo + 1;
                    atomicAdd(&stats[1], (unsigned long long)last);
                    atomicAdd(&stats[2], (unsigned long long)s_lists);
                }
                if (s_hit)
                {
                    collided = true;
                    break;
                }
            }
        }
        if (!STATS)
        {
            __syncthreads();
            collided = s_hit != 0;
        }
        if (CLOCKS)
        {
            c1 = clock64();
            t_collide += c1 - c2;
        }
    }
    if (tid == 0)
    {
        // an empty sample list is "no shot" (path_vec.size() == 0, src/hybrid_a_star.cpp:524)
        answer((!collided && nsamp > 0) ? 1 : 0, (float)length);
    }
    if (STATS)
    {
        if (tested)
            atomicAdd(&stats[0], tested);
        if (tid == 0)
            atomicAdd(&stats[7], 1ull);
    }
    if (CLOCKS)
    {
        if (lane == 0 && wave == solver)
        {
            atomicAdd(&stats[11], (unsigned long long)mk0), atomicAdd(&stats[12], (unsigned long long)mk1);
            atomicAdd(&stats[13], (unsigned long long)mk2), atomicAdd(&stats[14], (unsigned long long)mk3);
        }
        if (lane == 0 && (wave == 0 || wave == 2)) // tests of wave 0 (pose tests first) and of wave 2 (swept-segment tests first)
            atomicAdd(&stats[17 + wave / 2], (unsigned long long)t_test);
        if (tid == 0)
        {
            atomicAdd(&stats[6], 1ull);
            atomicAdd(&stats[8], (unsigned long long)t_solve);
            atomicAdd(&stats[9], (unsigned long long)t_interp);
            atomicAdd(&stats[10], (unsigned long long)t_collide);
            atomicAdd(&stats[19], (unsigned long long)n_batches);
            atomicAdd(&stats[15], (unsigned long long)t_load);
            atomicAdd(&stats[16], (unsigned long long)(clock64() - c0)); // whole workgroup
        }
    }
}

template <int MODE, int ADAPT>
__global__ void __launch_bounds__(ROUND_THREADS) k_expand(DevMap m, const StepIn *__restrict__ in, uint32_t n,
                                                          const hyas_goal *__restrict__ goals,
                                                          const float *__restrict__ twod, // [query][twod_size] or null
                                                          uint8_t *__restrict__ count, hyas_succ *__restrict__ succ, unsigned long long *stats)
{
    __shared__ Quad s_quads[STEP_LDS_QUADS];
    expand_role<MODE, ADAPT>(m, in, n, goals, twod, count, succ, stats, s_quads, blockIdx.x);
}

template <int MODE, int CURVE>
__global__ void __launch_bounds__(ROUND_THREADS, STEP_MIN_WAVES) k_shot(DevMap m, const StepIn *__restrict__ in, uint32_t n,
                                                                       const hyas_goal *__restrict__ goals,
                                                                       const double *__restrict__ seg_tab, uint32_t seg_cap,
                                                                       uint8_t *__restrict__ shot_ok, float *__restrict__ shot_len,
                                                                       unsigned long long *stats)
{
    __shared__ Quad s_quads[STEP_LDS_QUADS];
    if (blockIdx.x < n)
        shot_role<MODE, CURVE>(m, in, goals, seg_tab, seg_cap, shot_ok, shot_len, stats, s_quads, blockIdx.x);
}

// One search round in one launch: the first ceil(n_upd / ROUND_THREADS) workgroups bring the cell bounds of the
// successor filter up to date (CellFilter), workgroups [0, n) behind them evaluate the analytic shot
// of node i, the following ceil(n / NODES) workgroups the successors of NODES nodes
// each.  The parts are independent (a bound that arrives late only lets a successor through that the host then drops);
// the shot workgroups come before the successor workgroups because they run longest.
template <int MODE, int CURVE, int ADAPT>
__global__ void __launch_bounds__(ROUND_THREADS, STEP_MIN_WAVES) k_round(DevMap m, const StepIn *__restrict__ in, uint32_t n,
                                                                        const hyas_goal *__restrict__ goals, const float *__restrict__ twod,
                                                                        const double *__restrict__ seg_tab, uint32_t seg_cap,
                                                                        uint8_t *__restrict__ shot_ok, float *__restrict__ shot_len,
                                                                        uint8_t *__restrict__ count, hyas_succ *__restrict__ succ,
                                                                        unsigned long long *stats, StepIn *__restrict__ child, uint32_t flags,
                                                                        CellFilter filter, RoundOut ro)
{
    __shared__ Quad s_quads[STEP_LDS_QUADS];
    const uint32_t n_upd_blocks = (filter.n_upd + ROUND_THREADS - 1) / ROUND_THREADS;
    if (blockIdx.x < n_upd_blocks) // uniform across the workgroup
    {
        const uint32_t i = blockIdx.x * ROUND_THREADS + threadIdx.x;
        if (i < filter.n_upd)
        {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 u = reinterpret_cast<const u32x4 *>(filter.upd)[i];
            if (u.y < filter.cells) // signed order of the bits = order of the floats for costs >= 0, and -inf lies below them all
                atomicMin(filter.bound + (size_t)u.x * filter.cells + u.y, (int32_t)u.z);
        }
        return;
    }
    const uint32_t block = blockIdx.x - n_upd_blocks;
    if (block < n)
        shot_role<MODE, CURVE>(m, in, goals, seg_tab, seg_cap, shot_ok, shot_len, stats, s_quads, block, ro);
    else
        expand_role<MODE == 2 ? 0 : MODE, ADAPT>(m, in, n, goals, twod, count, succ, stats, s_quads, block - n, child, flags, filter.bound, filter.cells, ro);
}

// The accepted curves of n queries, one workgroup each: samples {x, y, t} of
// the shot from nodes[i] to goals[nodes[i].query] (the caller appends the goal,
// src/hybrid_a_star.cpp:526).  With offsets == nullptr only the sample counts
// are written; otherwise query i's samples start at out + 3 * offsets[i].
template <int CURVE>
__global__ void __launch_bounds__(256) k_shot_path(DevMap m, const StepIn *__restrict__ nodes, const hyas_goal *__restrict__ goals,
                                                   const double *__restrict__ seg_tab, uint32_t seg_cap,
                                                   const uint32_t *__restrict__ offsets, float *__restrict__ out,
                                                   uint32_t *__restrict__ n_out)
{
    __shared__ rs::Path s_path;
    __shared__ db::Path s_dubins;
    __shared__ float s_poly[HYAS_BZ_INTERVALS + 1][2], s_bezier_len; // cubic Bezier only
    bz::Curve bezier = {};
    const int tid = threadIdx.x;
    const StepIn node = nodes[blockIdx.x];
    const hyas_goal goal = goals[node.query];
    const double q0[3] = {(double)node.x, (double)node.y, (double)node.t};
    const double q1[3] = {(double)goal.x, (double)goal.y, (double)goal.t};
    const double rho = (double)m.shot_rho;
    if (CURVE == HYAS_CURVE_RS)
    {
        if (tid < 64)
        {
            double x, y, phi;
            rs::normalise(q0, q1, rho, x, y, phi);
            rs::Path p = solve_wave(x, y, phi, tid);
            if (tid == 0)
                s_path = p;
        }
    }
    else if (CURVE == HYAS_CURVE_BEZIER)
    {
        const float a[3] = {node.x, node.y, node.t}, b[3] = {goal.x, goal.y, goal.t};
        bezier = bz::make(a, b);
        if (tid <= HYAS_BZ_INTERVALS)
            bz::value(bezier, bz::powers(bz::length_param((unsigned)tid)), s_poly[tid][0], s_poly[tid][1]);
        __syncthreads();
        if (tid == 0) // CubicBezier::GetLength, src/cubic_bezier.cpp:166-178
        {
            float len = 0;
            for (int i = 0; i < HYAS_BZ_INTERVALS; ++i)
            {
                const float dx = s_poly[i + 1][0] - s_poly[i][0], dy = s_poly[i + 1][1] - s_poly[i][1];
                len += hy_sqrtf(dx * dx + dy * dy);
            }
            s_bezier_len = len;
        }
    }
    else if (tid == 0)
    {
        const float a[3] = {node.x, node.y, node.t}, b[3] = {goal.x, goal.y, goal.t};
        db::Path p;
        p.qi[0] = a[0], p.qi[1] = a[1], p.qi[2] = a[2];
        p.param[0] = p.param[1] = p.param[2] = 0, p.type = 0, p.rho = m.shot_rho;
        (void)db::init(a, b, m.shot_rho, p);
        s_dubins = p;
    }
    __syncthreads();
    const rs::Path path = s_path;
    const db::Path dpath = s_dubins;
    uint32_t nsamp;
    if (CURVE == HYAS_CURVE_RS)
        nsamp = count_samples(seg_tab, seg_cap, rho * path.total);
    else if (CURVE == HYAS_CURVE_BEZIER)
        nsamp = count_samples_db(m.seg_bz, m.seg_bz_cap, s_bezier_len);
    else
        nsamp = count_samples_db(m.seg_db, m.seg_db_cap, db::length(dpath));
    if (!offsets)
    {
        if (tid == 0)
            n_out[blockIdx.x] = nsamp;
        return;
    }
    float *dst = out + 3 * (size_t)offsets[blockIdx.x];
    for (uint32_t k = tid; k < nsamp; k += blockDim.x)
    {
        if (CURVE == HYAS_CURVE_RS)
        {
            double s[3];
            rs::interpolate(q0, path, seg_tab[k] / rho, rho, s);
            dst[3 * k] = (float)s[0], dst[3 * k + 1] = (float)s[1], dst[3 * k + 2] = rad_0_2pi((float)s[2]);
        }
        else if (CURVE == HYAS_CURVE_BEZIER)
        {
            const bz::Powers pw = bz::powers(m.seg_bz[k] / s_bezier_len);
            float x, y, angle, curvature;
            bz::value(bezier, pw, x, y);
            bz::angle_curvature(bezier, pw, angle, curvature);
            dst[3 * k] = x, dst[3 * k + 1] = y, dst[3 * k + 2] = rad_0_2pi(angle);
        }
        else
        {
            float q[3] = {0, 0, 0};
            (void)db::sample(dpath, m.seg_db[k], q);
            dst[3 * k] = q[0], dst[3 * k + 1] = q[1], dst[3 * k + 2] = rad_0_2pi(q[2]);
        }
    }
}

// Piecewise cubic Bezier interpolation of traced paths (PiecewiseCubicBezier::ConvertPiecewiseCubicBezierToVector3f,
// src/piecewise_cubic_bezier.cpp:336-357, over CubicBezier::ConvertCubicBezierToVector3f, src/cubic_bezier.cpp:179-194):
// curve c runs from anchor first[c] to the anchor behind it and is sampled at i / n[c], i = 0 .. n[c]; its n[c] + 1
// poses {x, y, heading} start at out + 3 * off[c].  One workgroup per curve.
__global__ void __launch_bounds__(64) k_bezier_points(const float *__restrict__ anchors, const uint32_t *__restrict__ first, const uint32_t *__restrict__ off,
                                                      const uint32_t *__restrict__ n, float *__restrict__ out)
{
    const uint32_t c = blockIdx.x, np = n[c];
    const float *a = anchors + 3 * (size_t)first[c];
    const float s[3] = {a[0], a[1], a[2]}, g[3] = {a[3], a[4], a[5]};
    const bz::Curve curve = bz::make(s, g);
    float *dst = out + 3 * (size_t)off[c];
    for (uint32_t i = threadIdx.x; i < np + 1; i += blockDim.x)
    {
        const bz::Powers pw = bz::powers((float)i / (float)np);
        float x, y, angle, curvature;
        bz::value(curve, pw, x, y);
        bz::angle_curvature(curve, pw, angle, curvature);
        // the heading is normalised by the curve and once more by the conversion to a node (src/utility.cpp:128-133)
        dst[3 * i] = x, dst[3 * i + 1] = y, dst[3 * i + 2] = rad_0_2pi(rad_0_2pi(angle));
    }
}

// -------------------------------------------------------------- path shortcut
// HybridAStar::ShortCut(path, true) (src/hybrid_a_star.cpp:1076-1179), one path per workgroup.  From the path's last
// node the reference walks towards its first for as long as CollisionDetection::IsTraversable(current, node)
// (src/collisiondetection.cpp:129-145: both poses free and the segment between them) holds, keeps the last node that
// passed, and starts again two nodes further down.  The walk of one anchor is a search for the first failing node:
// the workgroup tests 256 candidates at a time and stops at the first chunk that holds a failure.  A node fails either
// because it is not traversable from the anchor (:1135-1150: the node kept is the last one that passed) or because the
// steering angle from it towards the anchor is not below max_steering_angle (:1097-1125: the node itself is kept).  (The
// reference compares radians with a limit whose default is 30: every finite angle passes unless the limit is lowered.)
// A path of fewer than two nodes is returned unchanged (the reference reads input[size - 2] first: undefined there).
// paths: all poses {x, y, t}, path q = poses [offs[q], offs[q + 1]); out_idx + offs[q] + q: up to len + 1 input indices in
// the order the reference pushes them (last node first; the caller reverses), out_n[q] their number.
__global__ void __launch_bounds__(256) k_shortcut(DevMap m, const float *__restrict__ paths, const uint32_t *__restrict__ offs, uint32_t *__restrict__ out_idx,
                                                  uint32_t *__restrict__ out_n)
{
    __shared__ int s_fail, s_cur_free;
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x, base = offs[q];
    const int n = (int)(offs[q + 1] - base);
    const float *P = paths + 3 * (size_t)base;
    uint32_t *out = out_idx + base + q;
    if (n < 2)
    {
        if (tid == 0)
        {
            for (int i = 0; i < n; ++i)
                out[i] = (uint32_t)(n - 1 - i);
            out_n[q] = (uint32_t)n;
        }
        return;
    }
    const GlobalQuads gq{m.quads};
    int index1 = n - 1, prev = n - 2, count = 0;
    for (;;)
    {
        const float cx = P[3 * index1], cy = P[3 * index1 + 1], ct = P[3 * index1 + 2];
        if (tid == 0)
        {
            out[count] = (uint32_t)index1;
            s_fail = -1;
            s_cur_free = (on_grid(m, cx, cy) && pose_free(m, gq, cx, cy, ct)) ? 1 : 0;
        }
        ++count;
        __syncthreads();
        int fail = -1; // highest node below index1 that fails, as node << 1 | (1: on its steering angle)
        if (!s_cur_free)
            fail = (index1 - 1) << 1;
        else
            for (int j0 = index1 - 1; j0 >= 0; j0 -= 256)
            {
                const int j = j0 - tid;
                if (j >= 0)
                {
                    const float px = P[3 * j], py = P[3 * j + 1], pt = P[3 * j + 2];
                    const bool ok = on_grid(m, px, py) && segment_free(m, gq, cx, cy, px, py) && pose_free(m, gq, px, py, pt);
                    if (!ok)
                        atomicMax(&s_fail, j << 1);
                    else if (!(hy_fabsf(find_steering_angle(px, py, pt, cx, cy)) < m.max_steer))
                        atomicMax(&s_fail, j << 1 | 1);
                }
                __syncthreads();
                fail = s_fail;
                __syncthreads();
                if (fail >= 0)
                    break;
            }
        if (fail < 0) // every node down to the first passed (:1100-1106)
        {
            if (tid == 0)
                out[count] = 0u;
            ++count;
            break;
        }
        const bool on_steering = (fail & 1) != 0;
        fail >>= 1;
        if (on_steering)
            prev = fail; // the node itself (:1094, :1113)
        else if (fail + 1 <= index1 - 1)
            prev = fail + 1; // the last node that passed; none: the one kept from before (:1138)
        if (tid == 0)
            out[count] = (uint32_t)prev;
        ++count;
        index1 = fail == 0 ? 0 : fail - 1;
        if (index1 == 0)
            break;
        __syncthreads();
    }
    if (tid == 0)
        out_n[q] = (uint32_t)count;
}

// ---------------------------------------------------- 2D cost-to-goal flood
// HybridAStar::BuildAStarCostMap (src/hybrid_a_star.cpp:1020-1074) with AStar::CreateSuccessor(pred, 8)
// (src/a_star.cpp:214-277, step = resolution) and AStar::UpdateCostSoFar (:281-288): a lazy Dijkstra flood from the
// goal over the whole grid (obstacles do not matter, only IsOnGrid).  Cell coordinates are floats accumulated along
// the way the flood first reached the cell and a cell's cost is added up from float edge lengths, so the cost map
// depends on the exact pop order of the reference's open queue; fl::push / fl::pop restate that queue.
//
// One query per wave (the algorithm is serial; the work is spread across queries).  Per cell the wave keeps
// {cost, parent, state} in LDS; the open queue lives in LDS up to heap_cap entries and spills into `ws_heap` beyond;
// cell coordinates go to `ws_xy` (read once per pop, for the parent, while the queue is being rearranged).  Every lane
// runs the queue operations on the same values; lanes 0..7 look at one neighbour each.
// Workgroup b handles queries b, b + gridDim.x, ... and owns slot b of the workspaces (n_cells entries each).
typedef uint32_t flood_u32x4 __attribute__((ext_vector_type(4)));
struct FloodHeap
{
    HYAS_LDS flood_u32x4 *lds;
    flood_u32x4 *spill;
    int cap;
    __device__ __forceinline__ fl::Ent get(int i) const
    {
        flood_u32x4 v;
        if (i < cap)
            v = lds[i];
        else
            v = spill[i - cap];
        fl::Ent e;
        e.prio = __uint_as_float(v.x), e.id = v.y, e.x = __uint_as_float(v.z), e.y = __uint_as_float(v.w);
        return e;
    }
    __device__ __forceinline__ void set(int i, const fl::Ent &e) const
    {
        const flood_u32x4 v = {__float_as_uint(e.prio), e.id, __float_as_uint(e.x), __float_as_uint(e.y)};
        if (i < cap)
            lds[i] = v;
        else
            spill[i - cap] = v;
    }
};

// PS: {parent cell << 2 | state}; state 0 unseen, 1 open without parent (the goal), 2 open, 3 closed.
template <typename PS, bool CELLS_IN_LDS>
__global__ void __launch_bounds__(64) k_flood(DevMap m, const float *__restrict__ poses6, const uint32_t *__restrict__ qlist, uint32_t nq,
                                              float *__restrict__ twod, float *__restrict__ start_cost, float2 *ws_xy, uint4 *ws_heap,
                                              float *ws_g, PS *ws_ps, uint32_t heap_cap, uint32_t cells_pad)
{
    extern __shared__ uint4 s_flood[];
    const int lane = threadIdx.x;
    const uint32_t N = m.twod_size;
    HYAS_LDS flood_u32x4 *s_heap = (HYAS_LDS flood_u32x4 *)s_flood;
    HYAS_LDS float *s_g = (HYAS_LDS float *)(s_heap + heap_cap);
    HYAS_LDS PS *s_ps = (HYAS_LDS PS *)(s_g + cells_pad);
    float2 *xy = ws_xy + (size_t)blockIdx.x * N;
    float *g_g = ws_g ? ws_g + (size_t)blockIdx.x * N : nullptr;
    PS *g_ps = ws_ps ? ws_ps + (size_t)blockIdx.x * N : nullptr;
    const FloodHeap heap{s_heap, (flood_u32x4 *)ws_heap + (size_t)blockIdx.x * N, (int)heap_cap};
    auto ld_g = [&](uint32_t c) -> float { return CELLS_IN_LDS ? s_g[c] : g_g[c]; };
    auto st_g = [&](uint32_t c, float v) {
        if (CELLS_IN_LDS)
            s_g[c] = v;
        else
            g_g[c] = v;
    };
    auto ld_ps = [&](uint32_t c) -> PS { return CELLS_IN_LDS ? s_ps[c] : g_ps[c]; };
    auto st_ps = [&](uint32_t c, PS v) {
        if (CELLS_IN_LDS)
            s_ps[c] = v;
        else
            g_ps[c] = v;
    };
    for (uint32_t k = blockIdx.x; k < nq; k += gridDim.x)
    {
        const uint32_t q = qlist ? qlist[k] : k;
        for (uint32_t c = lane; c < N; c += 64)
            st_ps(c, (PS)0);
        __syncthreads();
        const float gx = poses6[6 * (size_t)q + 3], gy = poses6[6 * (size_t)q + 4];
        const int gid = idx_2d(m, gx, gy);
        int size = 0;
        if (gid >= 0 && (uint32_t)gid < N) // src/hybrid_a_star.cpp:1026-1031: the goal opens the queue with cost 0
        {
            st_ps((uint32_t)gid, (PS)1);
            st_g((uint32_t)gid, 0.f);
            xy[gid] = make_float2(gx, gy);
            fl::Ent e;
            e.prio = 0.f, e.id = (uint32_t)gid, e.x = gx, e.y = gy;
            heap.set(0, e);
            size = 1;
        }
        const float step = m.res_f;
        while (size > 0)
        {
            // ---- pop (every lane, same values) ----
            const fl::Ent top = heap.get(0);
            const uint32_t id = top.id;
            const float cx = top.x, cy = top.y;
            const PS p = ld_ps(id);
            float cg = ld_g(id);
            const bool has_parent = (p & 3) == 2;
            float2 pxy = make_float2(0.f, 0.f);
            if (has_parent)
                pxy = xy[(uint32_t)(p >> 2)]; // in flight while the queue is rearranged
            (void)fl::pop(heap, size);
            --size;
            if (has_parent) // AStar::UpdateCostSoFar: the edge from the parent is added when the cell is popped (:1041)
                cg = cg + normf(sub(f2(cx, cy), f2(pxy.x, pxy.y)));
            st_g(id, cg);
            st_ps(id, (PS)(p | 3));
            __syncthreads();
            // ---- the eight neighbours, x offset outer, y offset inner (src/a_star.cpp:226-262) ----
            bool act = false, want_push = false;
            uint32_t nid = 0xffffffffu;
            float xs = 0.f, ys = 0.f;
            if (lane < 8)
            {
                const int o = lane + (lane >= 4 ? 1 : 0); // position in the 3 x 3 loop, centre left out
                const int i = o / 3, j = o - 3 * i;
                xs = cx + (i == 0 ? -step : i == 1 ? 0.f : step);
                ys = cy + (j == 0 ? -step : j == 1 ? 0.f : step);
                if (on_grid(m, xs, ys))
                {
                    const int n_ = idx_2d(m, xs, ys);
                    if (n_ >= 0 && (uint32_t)n_ < N)
                        nid = (uint32_t)n_, act = true;
                }
            }
            // two neighbours in one cell (cannot happen on a regular grid, but nothing in the arithmetic forbids it):
            // then the visits run one after the other, as the reference's loop does
            bool clash = false;
            for (int d = 1; d < 8; ++d)
            {
                const uint32_t other = (uint32_t)__shfl((int)nid, (lane + d) & 7, 64);
                clash = clash || (act && other == nid);
            }
            const bool serial = __any(clash) != 0;
            auto visit = [&]() {
                const PS s = ld_ps(nid);
                const int st = (int)(s & 3);
                if (st == 3)
                    return; // already popped: cost final (:1053)
                if (st != 0)
                {
                    if (ld_g(nid) > cg) // :1056-1061: a cheaper parent; the queue keeps the old priority
                        st_g(nid, cg), st_ps(nid, (PS)((id << 2) | 2u));
                    return;
                }
                xy[nid] = make_float2(xs, ys);
                st_g(nid, cg);
                st_ps(nid, (PS)((id << 2) | 2u));
                want_push = true;
            };
            if (!serial)
            {
                if (act)
                    visit();
            }
            else
                for (int l = 0; l < 8; ++l)
                {
                    if (lane == l && act)
                        visit();
                    __syncthreads();
                }
            __syncthreads();
            // ---- pushes in loop order (:1066-1069) ----
            unsigned long long mask = __ballot(want_push);
            while (mask)
            {
                const int l = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                fl::Ent e;
                e.prio = cg;
                e.id = (uint32_t)__shfl((int)nid, l, 64);
                e.x = __shfl(xs, l, 64), e.y = __shfl(ys, l, 64);
                fl::push(heap, size, e);
                ++size;
            }
        }
        __syncthreads();
        // the cost map as BuildAStarCostMap leaves it: closed cells carry their cost, the rest are absent (-1)
        float *out = twod + (size_t)q * N;
        for (uint32_t c = lane; c < N; c += 64)
            out[c] = ((ld_ps(c) & 3) == 3) ? ld_g(c) : -1.f;
        if (start_cost && lane == 0)
        {
            const int sid = idx_2d(m, poses6[6 * (size_t)q], poses6[6 * (size_t)q + 1]);
            start_cost[q] = (sid >= 0 && (uint32_t)sid < N && (ld_ps((uint32_t)sid) & 3) == 3) ? ld_g((uint32_t)sid) : -1.f;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------ launch wrappers
static inline int grid_for(size_t n, int block = 256, int cap = 256 * 8)
{
    size_t g = (n + block - 1) / block;
    if (g < 1)
        g = 1;
    return (int)(g > (size_t)cap ? cap : g);
}

void launch_collision(const DevMap &m, int mode, const float *xyt, const float *pred_xy, size_t n, uint8_t *out, hipStream_t s)
{
    if (n == 0)
        return;
    static const bool force_packed_lists = std::getenv("HYAS_FP_PACKED_LISTS") != nullptr; // the per-cell kernel, for comparison
    // the batched footprint kernels load poses 16 bytes and store verdicts 4 bytes at a time
    const bool aligned = ((uintptr_t)xyt & 15u) == 0 && ((uintptr_t)out & 3u) == 0;
    if (mode == HYAS_MODE_FOOTPRINT && !aligned)
        hipLaunchKernelGGL(k_collision_footprint, dim3(grid_for((n + 3) / 4)), dim3(256), 0, s, m, xyt, n, out, false);
    else if (mode == HYAS_MODE_FOOTPRINT && m.fp_m8 && m.occ_pad && !force_packed_lists)
    {
        // one workgroup of sixteen waves per compute unit: bitmaps (8 bytes per list) and the waves' pose strips in LDS
        const int threads = 1024;
        const size_t lds = (((size_t)m.fp_entries * 2 + 3) & ~(size_t)3) * sizeof(uint32_t) + (size_t)(threads / 64) * 192 * 16;
        const int blocks = (int)std::max<size_t>(1, std::min<size_t>(256, (n / 4 + threads - 1) / threads));
        static size_t lds_m8 = 0;
        if (lds_m8 < lds && hipFuncSetAttribute((const void *)k_collision_footprint_m8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess)
            lds_m8 = lds;
        hipLaunchKernelGGL(k_collision_footprint_m8, dim3(blocks), dim3(threads), lds, s, m, m.fp_bb / 2, xyt, n, out);
    }
    else if (mode == HYAS_MODE_FOOTPRINT && m.fp_win && m.occ_tiles && (m.fp_bb == 5 || m.fp_bb == 7 || m.fp_bb == 9) && !force_packed_lists)
    {
        // persistent grid: as many workgroups as fit on the chip, grid-stride over groups of four poses
        const int threads = m.fp_bb <= 5 ? 512 : 1024;
        const size_t lds = (((size_t)m.fp_entries * m.fp_win_words + 3) & ~(size_t)3) * sizeof(uint32_t) + (size_t)(threads / 64) * 192 * 16;
        // workgroups resident per compute unit: the registers allow four waves per SIMD
        const int per_cu = std::max(1, std::min<int>(1024 / threads, (int)((160 * 1024 - 512) / std::max<size_t>(lds, 1))));
        const int blocks = (int)std::max<size_t>(1, std::min<size_t>((size_t)256 * per_cu, (n / 4 + threads - 1) / threads));
        static size_t lds_set[3] = {0, 0, 0}; // dynamic LDS limit each variant was last given
        auto go = [&](auto kernel, int slot) {
            if (lds_set[slot] < lds && hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess)
                lds_set[slot] = lds;
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, s, m, xyt, n, out);
        };
        if (m.fp_bb == 5)
            go(k_collision_footprint_window<5>, 0);
        else if (m.fp_bb == 7)
            go(k_collision_footprint_window<7>, 1);
        else
            go(k_collision_footprint_window<9>, 2);
    }
    else if (mode == HYAS_MODE_FOOTPRINT && m.fp_pack && m.occ_tiles)
    {
        // one workgroup per CU (the table takes most of its LDS), grid-stride over the poses
        const size_t lds = (size_t)m.fp_entries * sizeof(uint4);
        const int blocks = (int)std::min<size_t>(256, (n / 4 + 1023) / 1024 + 1);
        if (m.fp_max_len <= 4)
        {
            (void)hipFuncSetAttribute((const void *)k_collision_footprint_tiles<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_collision_footprint_tiles<4>, dim3(blocks), dim3(1024), lds, s, m, xyt, n, out);
        }
        else
        {
            (void)hipFuncSetAttribute((const void *)k_collision_footprint_tiles<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_collision_footprint_tiles<7>, dim3(blocks), dim3(1024), lds, s, m, xyt, n, out);
        }
    }
    else if (mode == HYAS_MODE_FOOTPRINT)
        hipLaunchKernelGGL(k_collision_footprint, dim3(grid_for((n + 3) / 4)), dim3(256), 0, s, m, xyt, n, out, true);
    else
        hipLaunchKernelGGL(k_collision_polygon, dim3(grid_for(n)), dim3(256), 0, s, m, xyt, pred_xy, n, out);
}

void launch_occupancy(const uint8_t *cost, size_t n, int8_t *occ, hipStream_t s)
{
    if (n)
        hipLaunchKernelGGL(k_occupancy, dim3(grid_for(n)), dim3(256), 0, s, cost, n, occ);
}

void launch_occ_tiles(const uint8_t *cost, uint32_t w, uint32_t h, uint32_t tx, uint32_t ty, uint64_t *tiles, uint64_t *padded, hipStream_t s)
{
    const size_t total = (size_t)(tx + 3) * (ty + 3);
    hipLaunchKernelGGL(k_occ_tiles, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, cost, w, h, tx, ty, tiles, padded);
}

void launch_range_bins(const DevMap &m, const Quad *quads, uint32_t n_quads, const float *xs, uint32_t nxs, const float *ys, uint32_t nys, float range,
                       uint32_t *owner, RangeBins bins, unsigned long long *total, int pass, hipStream_t s)
{
    const uint32_t cells = m.w * m.h, tiles = bins.tiles_x * bins.tiles_y;
    if (cells == 0)
        return;
    const int grid = grid_for(tiles, 4, 256 * 32); // four waves (tiles) per workgroup
    if (pass == 0)
    {
        hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(cells)), dim3(256), 0, s, owner, (size_t)cells, 0xFFFFFFFFu);
        if ((size_t)nxs * nys)
            hipLaunchKernelGGL(k_cell_owner, dim3(grid_for((size_t)nxs * nys)), dim3(256), 0, s, m, xs, nxs, ys, nys, owner);
        hipLaunchKernelGGL(k_range_bins<false>, dim3(grid), dim3(256), 0, s, quads, n_quads, xs, ys, nys, owner, m.w, m.h, range, bins);
        hipLaunchKernelGGL(k_exclusive_scan_u32, dim3(1), dim3(1024), 0, s, bins.off, (size_t)tiles, total);
    }
    else
        hipLaunchKernelGGL(k_range_bins<true>, dim3(grid), dim3(256), 0, s, quads, n_quads, xs, ys, nys, owner, m.w, m.h, range, bins);
}

void launch_in_range_lists(const DevMap &m, const Quad *quads, const float *xs, const float *ys, uint32_t nys, float range, const uint32_t *owner,
                           RangeBins bins, uint32_t *off, uint16_t *idx, unsigned long long *total, int pass, hipStream_t s)
{
    const uint32_t cells = m.w * m.h;
    if (cells == 0)
        return;
    const int grid = grid_for(cells, 4, 256 * 32); // four waves (keys) per workgroup
    if (pass == 0)
    {
        hipLaunchKernelGGL(k_in_range_lists<false>, dim3(grid), dim3(256), 0, s, quads, xs, ys, nys, owner, m.w, cells, range, bins, off, idx);
        hipLaunchKernelGGL(k_exclusive_scan_u32, dim3(1), dim3(1024), 0, s, off, (size_t)cells, total);
    }
    else
        hipLaunchKernelGGL(k_in_range_lists<true>, dim3(grid), dim3(256), 0, s, quads, xs, ys, nys, owner, m.w, cells, range, bins, off, idx);
}

void launch_prune_lists(const Quad *quads, const uint32_t *cell_off, const uint16_t *cell_idx, const PruneArgs &a, uint32_t *pose_off, uint16_t *pose_idx,
                        uint32_t *seg_off, uint16_t *seg_idx, unsigned long long *totals2, int pass, hipStream_t s)
{
    if (a.cells == 0)
        return;
    const int grid = grid_for(a.cells, 256, 256 * 32);
    if (pass == 0)
    {
        hipLaunchKernelGGL(k_prune_lists<false>, dim3(grid), dim3(256), 0, s, quads, cell_off, cell_idx, a, pose_off, pose_idx, seg_off, seg_idx);
        hipLaunchKernelGGL(k_exclusive_scan_u32, dim3(1), dim3(1024), 0, s, pose_off, (size_t)a.cells, totals2);
        hipLaunchKernelGGL(k_exclusive_scan_u32, dim3(1), dim3(1024), 0, s, seg_off, (size_t)a.cells, totals2 + 1);
    }
    else
        hipLaunchKernelGGL(k_prune_lists<true>, dim3(grid), dim3(256), 0, s, quads, cell_off, cell_idx, a, pose_off, pose_idx, seg_off, seg_idx);
}

void launch_adaptive_ranges(const DevMap &m, const float *xs, const float *ys, uint32_t nys, const uint32_t *owner, const float *angles, uint32_t n_ang,
                            float radius, float *min_distance, uint32_t *off, float *rng3, float *kept, uint32_t keep, unsigned long long *total, int pass,
                            hipStream_t s)
{
    if (!kept || keep > 21) // (a wave copies a key's kept ranges with one lane per float)
        kept = nullptr, keep = 0;
    const uint32_t cells = m.w * m.h;
    if (cells == 0)
        return;
    const int grid = grid_for(cells, 4, 256 * 32); // four waves (keys) per workgroup
    if (pass == 0)
    {
        hipLaunchKernelGGL(k_adaptive_ranges<false>, dim3(grid), dim3(256), 0, s, m, xs, ys, nys, owner, angles, n_ang, radius, min_distance, off, rng3, kept, keep);
        hipLaunchKernelGGL(k_exclusive_scan_u32, dim3(1), dim3(1024), 0, s, off, (size_t)cells, total);
    }
    else
        hipLaunchKernelGGL(k_adaptive_ranges<true>, dim3(grid), dim3(256), 0, s, m, xs, ys, nys, owner, angles, n_ang, radius, min_distance, off, rng3, kept, keep);
}

void launch_table_rs(const float *xs, uint32_t nx, const float *ys, uint32_t ny, const float *ts, uint32_t k0, uint32_t k1, double rho,
                     float *out, hipStream_t s)
{
    size_t total = (size_t)(k1 - k0) * ny * nx;
    if (total == 0)
        return;
    const size_t blocks = (total + 255) / 256;
    const dim3 grid((uint32_t)(blocks < (1u << 30) ? blocks : (1u << 30)));
    if (total < (1ull << 32) - 256) // one thread per entry, no index beyond 32 bits
        hipLaunchKernelGGL(k_table_rs<uint32_t>, grid, dim3(256), 0, s, xs, nx, ys, ny, ts, k0, k1, rho, out);
    else
        hipLaunchKernelGGL(k_table_rs<size_t>, grid, dim3(256), 0, s, xs, nx, ys, ny, ts, k0, k1, rho, out);
}

void launch_table_dubins(const float *xs, uint32_t nx, const float *ys, uint32_t ny, const float *ts, uint32_t k0, uint32_t k1, float rho,
                         float *out, hipStream_t s)
{
    size_t total = (size_t)(k1 - k0) * ny * nx;
    if (total == 0)
        return;
    hipLaunchKernelGGL(k_table_dubins, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, s, xs, nx, ys, ny, ts, k0, k1, rho, out);
}

void launch_table_bezier(const float *xs, uint32_t nx, const float *ys, uint32_t ny, const float *ts, uint32_t k0, uint32_t k1, float *out,
                         hipStream_t s)
{
    size_t total = (size_t)(k1 - k0) * ny * nx;
    if (total == 0)
        return;
    hipLaunchKernelGGL(k_table_bezier, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, s, xs, nx, ys, ny, ts, k0, k1, out);
}

void launch_table_install(const DevMap &m, const float *raw, const float *xs, uint32_t nx, const float *ys, uint32_t ny, const float *ts,
                          uint32_t ns, uint32_t *rank, float *table, uint64_t table_size, hipStream_t s)
{
    size_t total = (size_t)ns * ny * nx;
    hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(table_size)), dim3(256), 0, s, rank, (size_t)table_size, 0xFFFFFFFFu);
    hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(table_size)), dim3(256), 0, s, reinterpret_cast<uint32_t *>(table), (size_t)table_size,
                       0x7FC00000u); // quiet NaN = key absent
    if (total == 0)
        return;
    for (int pass = 0; pass < 2; ++pass)
        hipLaunchKernelGGL(k_table_install, dim3(grid_for(total)), dim3(256), 0, s, m, raw, xs, nx, ys, ny, ts, ns, rank, table, table_size,
                           pass);
}

void launch_table_lookup(const DevMap &m, const float *xyt, size_t n, float *cost, hipStream_t s)
{
    if (n)
        hipLaunchKernelGGL(k_table_lookup, dim3(grid_for(n)), dim3(256), 0, s, m, xyt, n, cost);
}

void launch_expand(const DevMap &m, const StepIn *in, uint32_t n, const hyas_goal *goals, const float *twod, uint8_t *count,
                   hyas_succ *succ, unsigned long long *stats, int stats_mode, hipStream_t s)
{
    if (!n)
        return;
    const uint32_t nodes = m.ad_mode == 1 ? ExpandShape<1>::NODES : m.ad_mode == 2 ? ExpandShape<2>::NODES : ExpandShape<0>::NODES;
    const dim3 grid((n + nodes - 1) / nodes);
    if (m.ad_mode == 1) // the adaptive variants keep no work counters
        hipLaunchKernelGGL((k_expand<0, 1>), grid, dim3(EXPAND_THREADS), 0, s, m, in, n, goals, twod, count, succ, stats);
    else if (m.ad_mode == 2)
        hipLaunchKernelGGL((k_expand<0, 2>), grid, dim3(EXPAND_THREADS), 0, s, m, in, n, goals, twod, count, succ, stats);
    else if (stats && stats_mode == 1)
        hipLaunchKernelGGL((k_expand<1, 0>), grid, dim3(EXPAND_THREADS), 0, s, m, in, n, goals, twod, count, succ, stats);
    else
        hipLaunchKernelGGL((k_expand<0, 0>), grid, dim3(EXPAND_THREADS), 0, s, m, in, n, goals, twod, count, succ, stats);
}

// kernel<MODE, CURVE> chosen at run time: MODE 0 plain, 1 work counters, 2 phase clocks (Reeds-Shepp only)
template <template <int, int> class Launch, class... Args>
static void dispatch_mode_curve(const DevMap &m, int mode, Args... args)
{
    const bool rs = m.curve_type == HYAS_CURVE_RS, bz = m.curve_type == HYAS_CURVE_BEZIER;
    if (mode == 0 && rs)
        Launch<0, HYAS_CURVE_RS>::go(m, args...);
    else if (mode == 1 && rs)
        Launch<1, HYAS_CURVE_RS>::go(m, args...);
    else if (mode == 2 && rs)
        Launch<2, HYAS_CURVE_RS>::go(m, args...);
    else if (mode == 1 && bz)
        Launch<1, HYAS_CURVE_BEZIER>::go(m, args...);
    else if (bz)
        Launch<0, HYAS_CURVE_BEZIER>::go(m, args...);
    else if (mode == 1)
        Launch<1, HYAS_CURVE_DUBINS>::go(m, args...);
    else
        Launch<0, HYAS_CURVE_DUBINS>::go(m, args...);
}

template <int MODE, int CURVE>
struct LaunchShot
{
    static void go(const DevMap &m, const StepIn *in, uint32_t n, const hyas_goal *goals, const double *seg_tab, uint32_t seg_cap, uint8_t *shot_ok,
                   float *shot_len, unsigned long long *stats, hipStream_t s)
    {
        hipLaunchKernelGGL((k_shot<MODE, CURVE>), dim3(n), dim3(SHOT_THREADS), 0, s, m, in, n, goals, seg_tab, seg_cap, shot_ok, shot_len, stats);
    }
};
template <int MODE, int CURVE>
struct LaunchRound
{
    template <int ADAPT>
    static void with(const DevMap &m, const StepIn *in, uint32_t n, const hyas_goal *goals, const float *twod, const double *seg_tab, uint32_t seg_cap,
                     uint8_t *shot_ok, float *shot_len, uint8_t *count, hyas_succ *succ, unsigned long long *stats, StepIn *child,
                     uint32_t child_shoot, CellFilter filter, RoundOut ro, hipStream_t s)
    {
        constexpr uint32_t nodes = ExpandShape<ADAPT>::NODES;
        const dim3 grid((filter.n_upd + ROUND_THREADS - 1) / ROUND_THREADS + n + (n + nodes - 1) / nodes);
        hipLaunchKernelGGL((k_round<MODE, CURVE, ADAPT>), grid, dim3(ROUND_THREADS), 0, s, m, in, n, goals, twod, seg_tab, seg_cap, shot_ok, shot_len,
                           count, succ, stats, child, child_shoot, filter, ro);
    }
    static void go(const DevMap &m, const StepIn *in, uint32_t n, const hyas_goal *goals, const float *twod, const double *seg_tab, uint32_t seg_cap,
                   uint8_t *shot_ok, float *shot_len, uint8_t *count, hyas_succ *succ, unsigned long long *stats, StepIn *child, uint32_t child_shoot,
                   CellFilter filter, RoundOut ro, hipStream_t s)
    {
        if constexpr (MODE == 0) // the adaptive variants exist without counters only
        {
            if (m.ad_mode == 1)
                return with<1>(m, in, n, goals, twod, seg_tab, seg_cap, shot_ok, shot_len, count, succ, stats, child, child_shoot, filter, ro, s);
            if (m.ad_mode == 2)
                return with<2>(m, in, n, goals, twod, seg_tab, seg_cap, shot_ok, shot_len, count, succ, stats, child, child_shoot, filter, ro, s);
        }
        with<0>(m, in, n, goals, twod, seg_tab, seg_cap, shot_ok, shot_len, count, succ, stats, child, child_shoot, filter, ro, s);
    }
};

// Resolves the round kernel variant the next launches will use (the runtime loads a kernel's code object and looks the
// function up the first time it is asked for it): called once on the thread that owns the context, so that the driver
// threads, which make their first launches at the same moment, find that work done.
template <int MODE, int CURVE>
struct ResolveRound
{
    template <int ADAPT>
    static void with(hipError_t *e)
    {
        hipFuncAttributes attr;
        *e = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_round<MODE, CURVE, ADAPT>));
    }
    static void go(const DevMap &m, hipError_t *e)
    {
        if constexpr (MODE == 0)
        {
            if (m.ad_mode == 1)
                return with<1>(e);
            if (m.ad_mode == 2)
                return with<2>(e);
        }
        with<0>(e);
    }
};
hipError_t resolve_round_kernel(const DevMap &m, bool stats, int stats_mode)
{
    int mode = (stats && stats_mode >= 1 && stats_mode <= 2) ? stats_mode : 0;
    if (m.ad_mode != 0)
        mode = 0;
    hipError_t e = hipSuccess;
    dispatch_mode_curve<ResolveRound>(m, mode, &e);
    return e;
}

void launch_shot(const DevMap &m, const StepIn *in, uint32_t n, const hyas_goal *goals, const double *seg_tab, uint32_t seg_cap,
                 uint8_t *shot_ok, float *shot_len, unsigned long long *stats, int stats_mode, hipStream_t s)
{
    if (!n)
        return;
    const int mode = (stats && stats_mode >= 1 && stats_mode <= 2) ? stats_mode : 0;
    dispatch_mode_curve<LaunchShot>(m, mode, in, n, goals, seg_tab, seg_cap, shot_ok, shot_len, stats, s);
}

void launch_round(const DevMap &m, const StepIn *in, uint32_t n, const hyas_goal *goals, const float *twod, const double *seg_tab,
                  uint32_t seg_cap, uint8_t *shot_ok, float *shot_len, uint8_t *count, hyas_succ *succ, unsigned long long *stats,
                  int stats_mode, hipStream_t s, StepIn *child, bool child_shoot, CellFilter filter, RoundOut out)
{
    if (!n)
        return;
    int mode = (stats && stats_mode >= 1 && stats_mode <= 2) ? stats_mode : 0;
    if (m.ad_mode != 0)
        mode = 0;
    if (!filter.bound)
        filter = CellFilter{nullptr, 0, nullptr, 0};
    dispatch_mode_curve<LaunchRound>(m, mode, in, n, goals, twod, seg_tab, seg_cap, shot_ok, shot_len, count, succ, stats, child,
                                     (child_shoot ? ROUND_CHILD_SHOOT : 0u) | ROUND_TAG_CANDIDATE, filter, out, s);
}

void launch_bezier_points(const float *anchors, const uint32_t *first, const uint32_t *off, const uint32_t *n, uint32_t n_curves, float *out, hipStream_t s)
{
    if (n_curves)
        hipLaunchKernelGGL(k_bezier_points, dim3(n_curves), dim3(64), 0, s, anchors, first, off, n, out);
}

void launch_shortcut(const DevMap &m, const float *paths, const uint32_t *offs, uint32_t n_paths, uint32_t *out_idx, uint32_t *out_n, hipStream_t s)
{
    if (n_paths)
        hipLaunchKernelGGL(k_shortcut, dim3(n_paths), dim3(256), 0, s, m, paths, offs, out_idx, out_n);
}

void launch_shot_path(const DevMap &m, const StepIn *nodes, uint32_t n, const hyas_goal *goals, const double *seg_tab, uint32_t seg_cap,
                      const uint32_t *offsets, float *out, uint32_t *n_out, hipStream_t s)
{
    if (!n)
        return;
    if (m.curve_type == HYAS_CURVE_RS)
        hipLaunchKernelGGL(k_shot_path<HYAS_CURVE_RS>, dim3(n), dim3(256), 0, s, m, nodes, goals, seg_tab, seg_cap, offsets, out, n_out);
    else if (m.curve_type == HYAS_CURVE_BEZIER)
        hipLaunchKernelGGL(k_shot_path<HYAS_CURVE_BEZIER>, dim3(n), dim3(256), 0, s, m, nodes, goals, seg_tab, seg_cap, offsets, out, n_out);
    else
        hipLaunchKernelGGL(k_shot_path<HYAS_CURVE_DUBINS>, dim3(n), dim3(256), 0, s, m, nodes, goals, seg_tab, seg_cap, offsets, out, n_out);
}

// Plans the launch of k_flood for maps of n_cells cells: how many workgroups (= workspace slots), how much LDS, and
// which variant.  A query's flood is one serial chain of queue moves on one wave, so what the launch takes is the
// chain's latency times the number of queries a compute unit has to take one after the other.  Cell state
// {cost, parent | state} in LDS gives the shortest chain (map_large: 24.4 ms per query) but leaves room for one to
// eight queries per unit; with the cell state in the workspace (L2-resident) the chain is 7 % longer for one query per
// unit and 57 % longer for sixteen, and sixteen queries per unit run side by side (4096 map_large floods: 193 ms with
// the cells in LDS, 38 ms without).  So: cells in LDS while all queries of the call fit the units at once that way,
// otherwise cells in the workspace and as many queries per unit as the call needs, at most sixteen; the queue gets
// what is left of an equal share of the unit's LDS (spill area beyond).  max_per_cu > 0 caps the queries per unit
// (floods that run beside another chunk's rounds).  HYAS_FLOOD_CELLS_IN_HBM=<k> forces the workspace form with k
// queries per unit (test knob).
FloodPlan plan_flood(uint32_t n_cells, uint32_t n_queries, int heap_cap_override, int max_per_cu)
{
    FloodPlan fp{};
    fp.n_cells = n_cells;
    fp.narrow = n_cells <= 16384; // 14-bit parent + 2-bit state in 16 bits
    const size_t ps = fp.narrow ? 2 : 4;
    fp.cells_pad = (n_cells + 7u) & ~7u;
    const size_t cell_bytes = (size_t)fp.cells_pad * (4 + ps);
    const size_t lds_total = 160 * 1024, margin = 1024;
    const int cap = max_per_cu > 0 ? max_per_cu : 16;
    int per_cu = 0;
    size_t heap_cap = 1024;
    for (int k : {8, 4, 2, 1})
    {
        if (k > cap)
            continue;
        const size_t budget = lds_total / k - margin;
        if (cell_bytes + 256 * 16 <= budget)
        {
            per_cu = k;
            heap_cap = std::min<size_t>(1024, (budget - cell_bytes) / 16);
            break;
        }
    }
    fp.cells_in_lds = per_cu > 0 && (size_t)n_queries <= (size_t)256 * per_cu;
    const char *force = std::getenv("HYAS_FLOOD_CELLS_IN_HBM");
    if (force)
        fp.cells_in_lds = false;
    if (!fp.cells_in_lds)
    {
        per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)cap, ((size_t)n_queries + 255) / 256));
        if (force)
            per_cu = std::max(1, std::min(32, std::atoi(force)));
        heap_cap = std::min<size_t>(1024, (lds_total / per_cu - margin) / 16);
    }
    heap_cap = std::min<size_t>(heap_cap, std::max<uint32_t>(n_cells, 1));
    if (heap_cap_override > 0)
        heap_cap = (size_t)heap_cap_override;
    fp.heap_cap = (uint32_t)heap_cap;
    fp.lds_bytes = heap_cap * 16 + (fp.cells_in_lds ? cell_bytes : 0);
    if (max_per_cu == 1)
        fp.lds_bytes = std::max<size_t>(fp.lds_bytes, lds_total / 2 + margin); // no second workgroup on the unit
    fp.grid = (uint32_t)std::max<size_t>(1, std::min<size_t>(n_queries, (size_t)256 * per_cu));
    fp.ws_xy_bytes = (size_t)fp.grid * n_cells * sizeof(float2);
    fp.ws_heap_bytes = (size_t)fp.grid * n_cells * sizeof(uint4);
    fp.ws_g_bytes = fp.cells_in_lds ? 0 : (size_t)fp.grid * n_cells * sizeof(float);
    fp.ws_ps_bytes = fp.cells_in_lds ? 0 : (size_t)fp.grid * n_cells * ps;
    return fp;
}

template <typename PS, bool L>
static hipError_t launch_flood_as(const DevMap &m, const FloodPlan &fp, const float *poses6, const uint32_t *qlist, uint32_t nq, float *twod,
                                  float *start_cost, void *ws_xy, void *ws_heap, void *ws_g, void *ws_ps, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute((const void *)k_flood<PS, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.lds_bytes);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL((k_flood<PS, L>), dim3(fp.grid), dim3(64), fp.lds_bytes, s, m, poses6, qlist, nq, twod, start_cost, (float2 *)ws_xy,
                       (uint4 *)ws_heap, (float *)ws_g, (PS *)ws_ps, fp.heap_cap, fp.cells_pad);
    return hipGetLastError();
}

hipError_t launch_flood(const DevMap &m, const FloodPlan &fp, const float *poses6, const uint32_t *qlist, uint32_t nq, float *twod,
                        float *start_cost, void *ws_xy, void *ws_heap, void *ws_g, void *ws_ps, hipStream_t s)
{
    if (!nq)
        return hipSuccess;
    if (fp.narrow)
        return fp.cells_in_lds ? launch_flood_as<uint16_t, true>(m, fp, poses6, qlist, nq, twod, start_cost, ws_xy, ws_heap, ws_g, ws_ps, s)
                               : launch_flood_as<uint16_t, false>(m, fp, poses6, qlist, nq, twod, start_cost, ws_xy, ws_heap, ws_g, ws_ps, s);
    return fp.cells_in_lds ? launch_flood_as<uint32_t, true>(m, fp, poses6, qlist, nq, twod, start_cost, ws_xy, ws_heap, ws_g, ws_ps, s)
                           : launch_flood_as<uint32_t, false>(m, fp, poses6, qlist, nq, twod, start_cost, ws_xy, ws_heap, ws_g, ws_ps, s);
}

} // namespace hyas
