// hyas_ctx.hip -- C ABI (include/hyas.h): context life cycle, map ingest,
// heuristic-table build and the batched kernel entry points.
#include "hyas_ctx.h"
#include "hyas_adaptive.h"
#include "hyas_prune.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <fcntl.h>
#include <unistd.h>
#include <limits>
#include <mutex>

using namespace hyas;

// Scratch of hyas_table_install_noalloc_dev: one loop rank per table key (see k_table_install).  Every GPU the library
// runs on holds its own copy of this array until the process ends; in host code the name is only a handle for
// hipGetSymbolAddress, the address the kernels get is taken per device in hyas_create.
static __device__ __align__(256) uint32_t scratch[HYAS_NOALLOC_MAX_KEYS];
#define HYAS_MAX_DEVICES 64
static uint32_t *scratch_ptr[HYAS_MAX_DEVICES] = {nullptr}; // device address of `scratch`, by device ordinal

namespace hyas
{

int hip_fail(hyas_ctx *c, hipError_t e, const char *what)
{
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    if (c)
    {
        std::lock_guard<std::mutex> lock(c->err_mutex);
        c->err = buf;
    }
    return (e == hipErrorOutOfMemory) ? HYAS_ENOMEM : HYAS_ENODEV;
}

int ensure_scratch(hyas_ctx *c, size_t dev_bytes, size_t host_bytes)
{
    if (dev_bytes > c->d_scratch_bytes)
    {
        if (c->d_scratch)
            HYAS_HIP(c, hipFree(c->d_scratch));
        c->d_scratch = nullptr, c->d_scratch_bytes = 0;
        size_t want = dev_bytes + dev_bytes / 2;
        HYAS_HIP(c, hipMalloc(&c->d_scratch, want));
        c->d_scratch_bytes = want;
    }
    if (host_bytes > c->h_scratch_bytes)
    {
        if (c->h_scratch)
            HYAS_HIP(c, hipHostFree(c->h_scratch));
        c->h_scratch = nullptr, c->h_scratch_bytes = 0;
        size_t want = host_bytes + host_bytes / 2;
        HYAS_HIP(c, hipHostMalloc(&c->h_scratch, want, hipHostMallocDefault));
        c->h_scratch_bytes = want;
    }
    return HYAS_OK;
}

int ensure_twod(hyas_ctx *c, size_t bytes)
{
    if (bytes > c->d_twod_bytes)
    {
        if (c->d_twod)
            HYAS_HIP(c, hipFree(c->d_twod));
        c->d_twod = nullptr, c->d_twod_bytes = 0;
        HYAS_HIP(c, hipMalloc((void **)&c->d_twod, bytes));
        c->d_twod_bytes = bytes;
    }
    return HYAS_OK;
}

int ensure_buffer(hyas_ctx *c, void **buf, size_t *have, size_t bytes)
{
    if (bytes <= *have)
        return HYAS_OK;
    if (*buf)
        HYAS_HIP(c, hipFree(*buf));
    *buf = nullptr, *have = 0;
    const size_t want = bytes + bytes / 4;
    HYAS_HIP(c, hipMalloc(buf, want));
    *have = want;
    return HYAS_OK;
}

int flood_dev(hyas_ctx *c, const float *poses6, const uint32_t *qlist, uint32_t nq, float *twod, float *start_cost, hipStream_t s, int max_per_cu)
{
    if (!nq)
        return HYAS_OK;
    int cap_override = 0;
    if (const char *e = std::getenv("HYAS_FLOOD_HEAP")) // test knob: a tiny queue in LDS sends the rest through the spill area
        cap_override = std::max(1, std::atoi(e));
    const FloodPlan fp = plan_flood((uint32_t)c->twod_size, nq, cap_override, max_per_cu);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_xy = 0, o_heap = up(fp.ws_xy_bytes), o_g = o_heap + up(fp.ws_heap_bytes), o_ps = o_g + up(fp.ws_g_bytes),
                 total = o_ps + up(fp.ws_ps_bytes);
    int rc = ensure_buffer(c, &c->d_flood_ws, &c->d_flood_ws_bytes, total);
    if (rc != HYAS_OK)
        return rc;
    char *ws = (char *)c->d_flood_ws;
    hipError_t e = launch_flood(c->dm, fp, poses6, qlist, nq, twod, start_cost, ws + o_xy, ws + o_heap, fp.ws_g_bytes ? ws + o_g : nullptr,
                                fp.ws_ps_bytes ? ws + o_ps : nullptr, s);
    if (e != hipSuccess)
        return hip_fail(c, e, "2D flood launch");
    return HYAS_OK;
}

int shortcut_paths(hyas_ctx *c, const float *paths, const uint32_t *offs, size_t n, uint32_t *out_idx, uint32_t *out_n)
{
    if (!n)
        return HYAS_OK;
    const size_t total = offs[n];
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_paths = 0, o_offs = up(total * 3 * sizeof(float)), o_idx = o_offs + up((n + 1) * sizeof(uint32_t)),
                 o_n = o_idx + up((total + n) * sizeof(uint32_t)), bytes = o_n + up(n * sizeof(uint32_t));
    int rc = ensure_buffer(c, &c->d_sc, &c->d_sc_bytes, bytes);
    if (rc != HYAS_OK)
        return rc;
    char *d = (char *)c->d_sc;
    if (total)
        HYAS_HIP(c, hipMemcpyAsync(d + o_paths, paths, total * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HYAS_HIP(c, hipMemcpyAsync(d + o_offs, offs, (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    launch_shortcut(c->dm, (const float *)(d + o_paths), (const uint32_t *)(d + o_offs), (uint32_t)n, (uint32_t *)(d + o_idx), (uint32_t *)(d + o_n),
                    c->stream);
    HYAS_HIP(c, hipGetLastError());
    HYAS_HIP(c, hipMemcpyAsync(out_idx, d + o_idx, (total + n) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HYAS_HIP(c, hipMemcpyAsync(out_n, d + o_n, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HYAS_HIP(c, hipStreamSynchronize(c->stream));
    return HYAS_OK;
}

// Samples of n_curves cubic Bezier curves between consecutive anchor poses (k_bezier_points): anchors = n_anchors x
// {x, y, heading}, curve c from anchor first[c]; out receives the poses of curve c at off[c].
int bezier_points(hyas_ctx *c, const float *anchors, size_t n_anchors, const uint32_t *first, const uint32_t *off, const uint32_t *n, size_t n_curves,
                  size_t total_points, float *out)
{
    if (!n_curves || !total_points)
        return HYAS_OK;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_anc = 0, o_first = up(n_anchors * 3 * sizeof(float)), o_off = o_first + up(n_curves * sizeof(uint32_t)),
                 o_n = o_off + up(n_curves * sizeof(uint32_t)), o_out = o_n + up(n_curves * sizeof(uint32_t)),
                 bytes = o_out + up(total_points * 3 * sizeof(float));
    int rc = ensure_buffer(c, &c->d_sc, &c->d_sc_bytes, bytes);
    if (rc != HYAS_OK)
        return rc;
    char *d = (char *)c->d_sc;
    HYAS_HIP(c, hipMemcpyAsync(d + o_anc, anchors, n_anchors * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HYAS_HIP(c, hipMemcpyAsync(d + o_first, first, n_curves * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HYAS_HIP(c, hipMemcpyAsync(d + o_off, off, n_curves * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HYAS_HIP(c, hipMemcpyAsync(d + o_n, n, n_curves * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    launch_bezier_points((const float *)(d + o_anc), (const uint32_t *)(d + o_first), (const uint32_t *)(d + o_off), (const uint32_t *)(d + o_n),
                         (uint32_t)n_curves, (float *)(d + o_out), c->stream);
    HYAS_HIP(c, hipGetLastError());
    HYAS_HIP(c, hipMemcpyAsync(out, d + o_out, total_points * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HYAS_HIP(c, hipStreamSynchronize(c->stream));
    return HYAS_OK;
}

// Device memory for the round kernel's input that the host writes with plain stores: fine-grained (coherent)
// device memory on a GPU whose whole memory is visible to the CPU.  The kernel then reads its node from local
// memory (0.5 us) instead of fetching it from host memory over PCIe (1.5-2 us).  Leaves d_nodes null where that
// is not possible; the caller then keeps the nodes in pinned host memory.
int ensure_nodes(hyas_ctx *c, size_t bytes)
{
    const bool host_only = std::getenv("HYAS_NODES_IN_HOST_MEMORY") != nullptr;
    if (!host_only && c->d_nodes && bytes <= c->d_nodes_bytes)
        return HYAS_OK;
    if (c->d_nodes)
        (void)hipFree(c->d_nodes);
    c->d_nodes = nullptr, c->d_nodes_bytes = 0;
    if (host_only)
        return HYAS_OK;
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, c->device) != hipSuccess || large_bar != 1)
    {
        (void)hipGetLastError();
        return HYAS_OK;
    }
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess)
    {
        (void)hipGetLastError();
        return HYAS_OK;
    }
    // Is the block really mapped for this process, for reading and for writing?  Asked of the kernel, which answers
    // EFAULT where a plain store would end the process: one byte from the block into a pipe, one byte from
    // /dev/zero into the block, at both ends of the block.
    bool mapped = false;
    {
        int fds[2] = {-1, -1};
        const int zero = open("/dev/zero", O_RDONLY);
        if (zero >= 0 && pipe(fds) == 0)
        {
            char *b = (char *)p;
            mapped = write(fds[1], b, 1) == 1 && write(fds[1], b + bytes - 1, 1) == 1 && read(zero, b, 1) == 1 &&
                     read(zero, b + bytes - 1, 1) == 1;
        }
        if (zero >= 0)
            close(zero);
        if (fds[0] >= 0)
            close(fds[0]), close(fds[1]);
    }
    if (!mapped)
    {
        (void)hipFree(p);
        if (std::getenv("HYAS_TRACE"))
            fprintf(stderr, "hyas: device memory is not mapped for the host, nodes stay in pinned host memory\n");
        return HYAS_OK;
    }
    c->d_nodes = p, c->d_nodes_bytes = bytes;
    if (std::getenv("HYAS_TRACE"))
        fprintf(stderr, "hyas: node block %p, %zu bytes of device memory\n", p, bytes);
    return HYAS_OK;
}

void timer_begin(hyas_ctx *c) { (void)hipEventRecord(c->ev0, c->stream); }
void timer_end(hyas_ctx *c, int which)
{
    (void)hipEventRecord(c->ev1, c->stream);
    (void)hipEventSynchronize(c->ev1);
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess)
        c->k_ms[which] += ms;
    c->k_launches[which] += 1;
}
void reset_timers(hyas_ctx *c)
{
    for (int i = 0; i < 5; ++i)
        c->k_ms[i] = 0, c->k_launches[i] = 0;
}

// ------------------------------------------------------------------ ingest
static Quad make_quad(const float vx[4], const float vy[4])
{
    Quad q;
    for (int i = 0; i < 4; ++i)
        q.vx[i] = vx[i], q.vy[i] = vy[i];
    q.xmin = *std::min_element(vx, vx + 4), q.xmax = *std::max_element(vx, vx + 4);
    q.ymin = *std::min_element(vy, vy + 4), q.ymax = *std::max_element(vy, vy + 4);
    q.axis_rect = (vx[0] == q.xmin && vy[0] == q.ymin && vx[1] == q.xmax && vy[1] == q.ymin && vx[2] == q.xmax && vy[2] == q.ymax &&
                   vx[3] == q.xmin && vy[3] == q.ymax && q.xmin < q.xmax && q.ymin < q.ymax)
                      ? 1u
                      : 0u;
    q.pad[0] = q.pad[1] = q.pad[2] = 0;
    return q;
}

// Utility::IsPolygonInNeighbor, src/utility.cpp:999-1012: any bit-equal vertex
static bool share_vertex(const Quad &a, const Quad &b)
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (a.vx[i] == b.vx[j] && a.vy[i] == b.vy[j])
                return true;
    return false;
}

// Utility::CombinePolygon(cur, prev), src/utility.cpp:1014-1058.  The result is
// a general quadrilateral when the two cells only touch at a corner.
static bool combine(const Quad &p1, const Quad &p2, Quad &out)
{
    float vx[4], vy[4];
    auto pick = [&](int k, const Quad &q, int i) { vx[k] = q.vx[i], vy[k] = q.vy[i]; };
    if (p2.vy[0] > p1.vy[0])
        pick(0, p1, 0), pick(1, p1, 1), pick(2, p2, 2), pick(3, p2, 3);
    else if (p2.vx[0] < p1.vx[0])
        pick(0, p2, 0), pick(1, p1, 1), pick(2, p1, 2), pick(3, p2, 3);
    else if (p2.vy[0] < p1.vy[0])
        pick(0, p2, 0), pick(1, p2, 1), pick(2, p1, 2), pick(3, p1, 3);
    else if (p2.vx[0] > p1.vx[0])
        pick(0, p1, 0), pick(1, p2, 1), pick(2, p2, 2), pick(3, p1, 3);
    else
        return false;
    out = make_quad(vx, vy);
    return true;
}

// The float loop variables of src/collisiondetection.cpp:268-303 / :324-348.
static void cell_axes(const hyas_ctx *c, std::vector<float> &xs, std::vector<float> &ys)
{
    const float res = (float)c->res, ox = (float)c->ox, oy = (float)c->oy;
    const float x_limit = ox + c->w * res, y_limit = oy + c->h * res;
    for (float x = ox; x < x_limit; x = x + res)
        xs.push_back(x);
    for (float y = oy; y < y_limit; y = y + res)
        ys.push_back(y);
}

// CollisionDetection::SetObstacleVec, src/collisiondetection.cpp:262-317.  occ: the nav_msgs::OccupancyGrid values of
// the cells (k_occupancy).  Every step of the scan looks at the polygon the step before closed, so it stays a host loop.
static void extract_obstacles(hyas_ctx *c, const std::vector<int8_t> &occ, const std::vector<float> &xs, const std::vector<float> &ys)
{
    const float res = (float)c->res;
    DevMap hm = c->dm; // only the scalar members are used by grid_key()
    c->quads.clear();
    for (float x : xs)
        for (float y : ys)
        {
            uint32_t gi = grid_key(hm, x, y);
            if (gi >= occ.size() || occ[gi] == 0)
                continue;
            // CreatePolygon(origin, res, res), src/utility.cpp:457-474
            const float vx[4] = {x, res + x, res + x, x};
            const float vy[4] = {y, y, y + res, y + res};
            Quad cur = make_quad(vx, vy);
            if (!c->quads.empty() && share_vertex(c->quads.back(), cur))
            {
                Quad merged;
                if (combine(cur, c->quads.back(), merged))
                    c->quads.back() = merged;
                // an un-mergeable neighbour leaves an EMPTY polygon in the
                // reference; it cannot arise from grid cells, keep the old one
            }
            else
                c->quads.push_back(cur);
        }
    if (c->P.map_boundary_obstacle)
    {
        // CollisionDetection::AddMapBoundaryAsObstacle, src/collisiondetection.cpp:1223-1240: four rectangles around
        // the grid, given in cells whatever the resolution and the origin are; CreatePolygon(origin, width, height),
        // src/utility.cpp:457-474
        auto add = [&](float x, float y, float width, float height) {
            const float vx[4] = {x, width + x, width + x, x};
            const float vy[4] = {y, y, y + height, y + height};
            c->quads.push_back(make_quad(vx, vy));
        };
        const uint32_t w = c->w, h = c->h;
        add(0.f, -1.f, (float)w, 1.f);
        add((float)w, -1.f, 1.f, (float)(h + 2));
        add(-1.f, (float)h, (float)(w + 2), 1.f);
        add(-1.f, 0.f, 1.f, (float)h);
    }
}

// --------------------------------------------------- footprint-list table
// Rasterises the rotated vehicle rectangle into a bb x bb cell box for every
// sub-cell offset and heading, as Lookup::collisionLookup does
// (include/lookup.h:30-284): edges are traced cell by cell with a DDA walk,
// then each row is filled between its outermost marked cells.
namespace fpt
{
struct Pt
{
    float x, y;
};
static inline int sgn(float v) { return v >= 0 ? 1 : -1; }

static void trace_edge(std::vector<char> &box, int bb, Pt a, Pt b) // :153-213
{
    int X = (int)a.x, Y = (int)a.y;
    box[Y * bb + X] = 1;
    const float tx = b.x - a.x, ty = b.y - a.y;
    const int stepX = sgn(tx), stepY = sgn(ty);
    const float tDeltaX = (tx != 0) ? 1.f / std::abs(tx) : 1000;
    const float tDeltaY = (ty != 0) ? 1.f / std::abs(ty) : 1000;
    float tMaxX = (stepX > 0) ? tDeltaX * (1 - (a.x - (long)a.x)) : tDeltaX * (a.x - (long)a.x);
    float tMaxY = (stepY > 0) ? tDeltaY * (1 - (a.y - (long)a.y)) : tDeltaY * (a.y - (long)a.y);
    const int ex = (int)b.x, ey = (int)b.y;
    while (ex != X || ey != Y)
    {
        if (tMaxX < tMaxY && std::abs(X + stepX - ex) < std::abs(X - ex))
        {
            tMaxX = tMaxX + tDeltaX;
            X = X + stepX;
        }
        else if (tMaxY < tMaxX && std::abs(Y + stepY - ey) < std::abs(Y - ey))
        {
            tMaxY = tMaxY + tDeltaY;
            Y = Y + stepY;
        }
        else if (2 >= std::abs(X - ex) + std::abs(Y - ey))
        {
            if (std::abs(X - ex) > std::abs(Y - ey))
                X = X + stepX;
            else
                Y = Y + stepY;
        }
        else
            break;
        box[Y * bb + X] = 1;
    }
}

static void build(const hyas_params &P, std::vector<int32_t> &out)
{
    const float cell = P.fp_cell_size;
    const double width = P.fp_width, length = P.fp_length;
    const int bb = (int)std::ceil((std::sqrt(width * width + length * length) + 4) / cell); // constants.h:56
    const int pr = P.position_resolution, headings = P.headings;
    const float dtheta = (float)(2 * HY_PI / (double)(float)headings); // constants.h:46
    out.assign((size_t)pr * pr * headings * 129, 0);
    std::vector<char> box((size_t)bb * bb);
    // scan-fill state that the reference keeps alive across rows AND entries
    int first_marked = 0, last_marked = 0;
    for (int q = 0; q < pr * pr; ++q)
    {
        const float offx = 1.f / pr * (q % pr), offy = 1.f / pr * (q / pr); // :87-92
        Pt c{(float)bb / 2 + offx, (float)bb / 2 + offy};
        const double hl = length / 2 / (double)cell, hw = width / 2 / (double)cell;
        const Pt corner[4] = {{(float)((double)c.x - hl), (float)((double)c.y - hw)},
                              {(float)((double)c.x - hl), (float)((double)c.y + hw)},
                              {(float)((double)c.x + hl), (float)((double)c.y + hw)},
                              {(float)((double)c.x + hl), (float)((double)c.y - hw)}};
        float theta = 0;
        for (int o = 0; o < headings; ++o)
        {
            std::fill(box.begin(), box.end(), 0);
            Pt r[4];
            const float ct = hy_cosf(theta), st = hy_sinf(theta);
            for (int j = 0; j < 4; ++j)
            {
                const float rx = corner[j].x - c.x, ry = corner[j].y - c.y;
                r[j].x = rx * ct - ry * st + c.x;
                r[j].y = rx * st + ry * ct + c.y;
            }
            theta += dtheta;
            for (int k = 0; k < 4; ++k)
                trace_edge(box, bb, r[k], r[(k + 1) & 3]);
            for (int i = 0; i < bb; ++i) // :216-240
            {
                bool inside = false;
                for (int j = 0; j < bb; ++j)
                {
                    for (int k = 0; k < bb; ++k)
                        if (box[i * bb + k])
                        {
                            if (!inside)
                                first_marked = k, inside = true;
                            last_marked = k;
                        }
                    if (j > first_marked && j < last_marked && inside)
                        box[i * bb + j] = 1;
                }
            }
            int32_t *e = &out[((size_t)q * headings + o) * 129];
            int count = 0;
            for (int i = 0; i < bb; ++i)
                for (int j = 0; j < bb; ++j)
                    if (box[i * bb + j])
                    {
                        if (count < 64)
                            e[1 + 2 * count] = j - (int)c.x, e[2 + 2 * count] = i - (int)c.y;
                        ++count;
                    }
            e[0] = count;
        }
    }
}
} // namespace fpt

static void set_resolution(DevMap &m, double res, double ox, double oy)
{
    m.res = res, m.ox = ox, m.oy = oy;
    m.res_f = (float)res, m.ox_f = (float)ox, m.oy_f = (float)oy; // src/collisiondetection.cpp:13-15
    int e = 0;
    const bool p2 = std::frexp(res, &e) == 0.5 && e > -60 && e < 60 && (double)m.res_f == res;
    m.res_pow2 = p2 ? 1 : 0;
    m.inv_res = p2 ? 1.0 / res : 0.0;
    m.inv_res_f = p2 ? (float)(1.0 / res) : 0.f;
}

static void free_map(hyas_ctx *c)
{
    void *ptrs[] = {c->d_charmap, c->d_quads, c->d_cell_off, c->d_cell_idx, c->d_fp, c->d_table, c->d_tab_xs, c->d_tab_ys, c->d_tab_ts, c->d_seg,
                    c->d_pose_off, c->d_pose_idx, c->d_seg_off, c->d_seg_idx, c->d_occ_tiles, c->d_fp_pack, c->d_seg_db, c->d_fp_win,
                    c->d_ad_off, c->d_ad_rng, c->d_ad_density, c->d_occ_pad, c->d_fp_m8};
    for (void *p : ptrs)
        if (p)
            (void)hipFree(p);
    c->d_charmap = nullptr, c->d_quads = nullptr, c->d_cell_off = nullptr, c->d_cell_idx = nullptr, c->d_fp = nullptr;
    c->d_table = nullptr, c->d_tab_xs = c->d_tab_ys = c->d_tab_ts = nullptr, c->d_seg = nullptr;
    c->d_pose_off = c->d_seg_off = nullptr, c->d_pose_idx = c->d_seg_idx = nullptr;
    c->d_occ_tiles = nullptr, c->d_fp_pack = nullptr, c->d_seg_db = nullptr, c->d_fp_win = nullptr;
    c->d_ad_off = nullptr, c->d_ad_rng = nullptr, c->d_ad_density = nullptr, c->d_occ_pad = nullptr, c->d_fp_m8 = nullptr;
    c->ad_off.clear(), c->ad_rng.clear(), c->ad_density.clear(), c->ad_min_distance.clear();
    c->tab_xs.clear(), c->tab_ys.clear(), c->tab_ts.clear();
    c->table_size = 0, c->twod_size = 0, c->seg_cap = 0;
    c->have_map = c->have_table = false;
}

template <typename T>
static int upload(hyas_ctx *c, T *&dst, const std::vector<T> &src, size_t min_elems = 1)
{
    size_t n = std::max(src.size(), min_elems);
    const size_t slack = 64; // kernels fetch list entries in 16-byte blocks, one block ahead
    HYAS_HIP(c, hipMalloc((void **)&dst, n * sizeof(T) + slack));
    HYAS_HIP(c, hipMemset((char *)dst + src.size() * sizeof(T), 0, n * sizeof(T) + slack - src.size() * sizeof(T)));
    if (!src.empty())
        HYAS_HIP(c, hipMemcpy(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return HYAS_OK;
}

// Derived forms of the grid and the footprint table for the batched footprint kernel (DevMap::occ_tiles, fp_pack).
static int upload_footprint_forms(hyas_ctx *c)
{
    DevMap &m = c->dm;
    const uint32_t w = c->w, h = c->h, tx = (w + 7) / 8, ty = (h + 7) / 8;
    // occupancy bits packed on the GPU from the uploaded cost bytes: plain tiles and the bordered form
    const size_t n_tiles = (size_t)tx * ty, n_pad = (size_t)(tx + 3) * (ty + 3), slack = 64;
    HYAS_HIP(c, hipMalloc((void **)&c->d_occ_tiles, n_tiles * 8 + slack));
    HYAS_HIP(c, hipMalloc((void **)&c->d_occ_pad, n_pad * 8 + slack));
    HYAS_HIP(c, hipMemsetAsync((char *)c->d_occ_tiles + n_tiles * 8, 0, slack, c->stream));
    HYAS_HIP(c, hipMemsetAsync((char *)c->d_occ_pad + n_pad * 8, 0, slack, c->stream));
    launch_occ_tiles(c->d_charmap, w, h, tx, ty, c->d_occ_tiles, c->d_occ_pad, c->stream);
    HYAS_HIP(c, hipGetLastError());
    HYAS_HIP(c, hipStreamSynchronize(c->stream));
    int rc = HYAS_OK;
    m.occ_tiles = c->d_occ_tiles, m.tiles_x = tx, m.tiles_y = ty;
    // window bitmaps: the box of Lookup::collisionLookup (include/lookup.h:56-60) is bb cells wide and the pose's
    // cell sits at box cell ((int)c.x, (int)c.y) with c = bb / 2 + sub-cell offset (:87-92, :262-263)
    m.fp_win = nullptr, m.fp_bb = 0, m.fp_win_words = 0;
    {
        const hyas_params &P = c->P;
        const int bb = (int)std::ceil((std::sqrt(P.fp_width * P.fp_width + P.fp_length * P.fp_length) + 4) / P.fp_cell_size);
        const int pr = P.position_resolution, headings = P.headings;
        const size_t n_entries = c->fp.size() / 129;
        const int words = (bb * bb + 2 + 31) / 32;
        bool ok = bb >= 1 && bb <= 9 && n_entries == (size_t)pr * pr * headings && n_entries * words * 4 <= 150 * 1024;
        std::vector<uint32_t> win(ok ? n_entries * words : 0, 0u);
        for (size_t e = 0; e < n_entries && ok; ++e)
        {
            const int q = (int)(e / headings);
            const int cxi = (int)((float)bb / 2 + 1.f / pr * (q % pr)), cyi = (int)((float)bb / 2 + 1.f / pr * (q / pr));
            if (cxi - bb / 2 < 0 || cxi - bb / 2 > 1 || cyi - bb / 2 < 0 || cyi - bb / 2 > 1)
                ok = false;
            const int32_t *src = &c->fp[e * 129];
            uint32_t *dst = &win[e * words];
            const int len = std::min(src[0], 64);
            for (int k = 0; k < len && ok; ++k)
            {
                const int j = src[1 + 2 * k] + cxi, i = src[2 + 2 * k] + cyi;
                if (j < 0 || j >= bb || i < 0 || i >= bb)
                    ok = false;
                else
                    dst[(i * bb + j) >> 5] |= 1u << ((i * bb + j) & 31);
            }
            const int fb = bb * bb;
            if (ok && cxi - bb / 2)
                dst[fb >> 5] |= 1u << (fb & 31);
            if (ok && cyi - bb / 2)
                dst[(fb + 1) >> 5] |= 1u << ((fb + 1) & 31);
        }
        if (ok && n_entries && !std::getenv("HYAS_NO_FP_WINDOW"))
        {
            if ((rc = upload(c, c->d_fp_win, win)) != HYAS_OK)
                return rc;
            m.fp_win = c->d_fp_win, m.fp_bb = bb, m.fp_win_words = words;
        }
        // boxes of up to 7 cells: the same bitmaps with one byte per box row, and the tiles with an all-ones border
        m.fp_m8 = nullptr, m.occ_pad = nullptr;
        if (m.fp_win && bb <= 7 && n_entries * 8 <= 100 * 1024 && !std::getenv("HYAS_FP_ROWS"))
        {
            std::vector<uint64_t> m8(n_entries, 0);
            for (size_t e = 0; e < n_entries; ++e)
            {
                const uint32_t *src = &win[e * words];
                auto bit = [&](int b) { return (uint64_t)((src[b >> 5] >> (b & 31)) & 1u); };
                for (int i = 0; i < bb; ++i)
                    for (int j = 0; j < bb; ++j)
                        m8[e] |= bit(i * bb + j) << (8 * i + j);
                m8[e] |= bit(bb * bb) << 62 | bit(bb * bb + 1) << 63;
            }
            if ((rc = upload(c, c->d_fp_m8, m8)) != HYAS_OK)
                return rc;
            m.fp_m8 = c->d_fp_m8, m.occ_pad = c->d_occ_pad;
        }
    }
    const size_t entries = c->fp.size() / 129;
    std::vector<FpPack> pack(entries);
    bool fits = entries * sizeof(FpPack) <= 150 * 1024; // the kernel keeps the whole table in LDS
    for (size_t e = 0; e < entries && fits; ++e)
    {
        const int32_t *src = &c->fp[e * 129];
        uint8_t b[16] = {0};
        if (src[0] < 0 || src[0] > 7)
            fits = false;
        else
        {
            b[0] = (uint8_t)src[0];
            for (int i = 0; i < src[0]; ++i)
            {
                const int dx = src[1 + 2 * i], dy = src[2 + 2 * i];
                if (dx < -128 || dx > 127 || dy < -128 || dy > 127)
                    fits = false;
                b[1 + 2 * i] = (uint8_t)(int8_t)dx, b[2 + 2 * i] = (uint8_t)(int8_t)dy;
            }
        }
        std::memcpy(&pack[e], b, 16);
    }
    m.fp_pack = nullptr, m.fp_entries = (uint32_t)entries, m.fp_max_len = 0;
    for (size_t e = 0; e < entries; ++e)
        m.fp_max_len = std::max(m.fp_max_len, c->fp[e * 129]);
    if (fits && entries)
    {
        if ((rc = upload(c, c->d_fp_pack, pack)) != HYAS_OK)
            return rc;
        m.fp_pack = c->d_fp_pack;
    }
    return HYAS_OK;
}

} // namespace hyas

// ============================================================== C ABI
extern "C" {

void hyas_default_params(hyas_params *p)
{
    std::memset(p, 0, sizeof *p);
    p->position_resolution = 10, p->headings = 72, p->cd_min_turning_radius = 6, p->curve_type = 0;
    p->vehicle_width = 1, p->vehicle_length = 2, p->obstacle_detection_range = 5, p->map_boundary_obstacle = 0;
    p->number_of_successors = 3, p->adaptive_steering_angle_and_step_size = 0, p->adaptive_step_size = 1; // include/parameter_manager.h:129-131
    p->analytical_expansion_every_point = 1, p->analytical_expansion = 1;
    p->penalty_turning = 1.05f, p->penalty_reverse = 1, p->penalty_change_of_direction = 2, p->reverse = 0;
    p->max_iterations = 10000, p->steering_angle = 5, p->goal_range = 1e-3f, p->tie_breaker = 0.01f;
    p->min_turning_radius = 6, p->fp_width = 0.18, p->fp_length = 0.22, p->fp_cell_size = 1;
    p->short_cut = 1, p->max_steering_angle = 30, p->add_one_more_successor = 1;
    p->steering_angle_towards_free_angle_range_for_obstacle_angle_range = 1, p->fixed_number_of_steering_angle_in_free_angle_range = 1;
    p->constant_density = 0.5f, p->curve_step_size = 1;
}

void hyas_hot_path_params(hyas_params *p)
{
    hyas_default_params(p);
    p->curve_type = HYAS_CURVE_RS, p->number_of_successors = 5, p->steering_angle = 5, p->reverse = 1, p->short_cut = 0;
    p->adaptive_step_size = 0; // the fixed-step branch, src/hybrid_a_star.cpp:600-621
}

static bool params_ok(const hyas_params *p, std::string &why)
{
    if ((p->adaptive_steering_angle_and_step_size || p->adaptive_step_size) && p->consider_steering_angle_range_for_obstacle_density)
        why = "consider_steering_angle_range_for_obstacle_density is not supported (the obstacle density is taken per grid cell)";
    else if ((p->adaptive_steering_angle_and_step_size || p->adaptive_step_size) && !(p->obstacle_detection_range > 0))
        why = "the adaptive modes need a positive obstacle_detection_range";
    else if (p->headings < 1 || p->headings > 720 || p->position_resolution < 1 || p->position_resolution > 64)
        why = "headings / position_resolution out of range";
    else if (p->curve_type != HYAS_CURVE_RS && p->curve_type != HYAS_CURVE_DUBINS && p->curve_type != HYAS_CURVE_BEZIER)
        why = "curve_type must be 0 (Dubins), 1 (Reeds-Shepp) or 2 (cubic Bezier)";
    else if (!(p->vehicle_width >= 0) || !(p->vehicle_length >= 0))
        why = "vehicle dimensions must not be negative";
    else if (!(p->min_turning_radius > 0) || !(p->cd_min_turning_radius > 0))
        why = "turning radius must be positive";
    else if (p->number_of_successors < 1 || 1 + 2 * ((p->number_of_successors - 1) / 2) + (p->reverse ? 1 : 0) > HYAS_MAX_SUCC)
        why = "number_of_successors out of range";
    else
        return true;
    return false;
}

hyas_ctx *hyas_create(const hyas_params *params, int device)
{
    if (!params)
        return nullptr;
    std::string why;
    if (!params_ok(params, why))
    {
        fprintf(stderr, "hyas_create: %s\n", why.c_str());
        return nullptr;
    }
    int ndev = 0;
    // kernel arguments in device memory (2-3 us less per launch); read by the HIP runtime when it starts up, so this
    // only takes effect in a process that has made no HIP call yet; a value that is already set is left alone
    (void)setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    {
        fprintf(stderr, "hyas_create: no HIP device %d (found %d)\n", device, ndev);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess)
        return nullptr;
    static std::mutex scratch_mu; // contexts may be created from several threads
    std::lock_guard<std::mutex> scratch_lock(scratch_mu);
    if (device < HYAS_MAX_DEVICES && !scratch_ptr[device])
    {
        // this device's copy of the static scratch array (the device is current)
        uint32_t *p = nullptr;
        const hipError_t e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
        {
            fprintf(stderr, "hyas_create: hipGetSymbolAddress(scratch) failed on device %d: %s\n", device, hipGetErrorString(e));
            return nullptr;
        }
        scratch_ptr[device] = p;
    }
    hyas_ctx *c = new hyas_ctx();
    c->P = *params;
    c->device = device;
    bool ok = hipStreamCreate(&c->stream) == hipSuccess;
    c->round_stream[0] = c->stream;
    for (int g = 0; g < HYAS_MAX_GROUPS && ok; ++g)
    {
        if (g > 0)
            ok = hipStreamCreate(&c->round_stream[g]) == hipSuccess;
        for (int k = 0; k < 2 && ok; ++k)
            ok = hipEventCreate(&c->round_ev[g][k]) == hipSuccess;
    }
    if (!ok || hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipMalloc((void **)&c->d_stats, 24 * sizeof(unsigned long long)) != hipSuccess)
    {
        fprintf(stderr, "hyas_create: stream/event creation failed\n");
        delete c;
        return nullptr;
    }
    (void)hipMemset(c->d_stats, 0, 24 * sizeof(unsigned long long));
    return c;
}

void hyas_destroy(hyas_ctx *c)
{
    if (!c)
        return;
    const bool trace = std::getenv("HYAS_TRACE") != nullptr; // where a teardown that does not come back stands
    auto mark = [trace](const char *what) {
        if (trace)
            fprintf(stderr, "hyas destroy: %s\n", what), fflush(stderr);
    };
    mark("begin");
    (void)hipSetDevice(c->device);
    free_search_pools(c);
    mark("worker threads joined");
    for (int g = 0; g < HYAS_MAX_GROUPS; ++g) // round_stream[0] is c->stream
        if (c->round_stream[g])
            (void)hipStreamSynchronize(c->round_stream[g]);
    if (c->prep_stream)
        (void)hipStreamSynchronize(c->prep_stream);
    mark("streams drained");
    free_map(c);
    if (c->d_flood_ws)
        (void)hipFree(c->d_flood_ws);
    if (c->d_pts)
        (void)hipFree(c->d_pts);
    if (c->d_sc)
        (void)hipFree(c->d_sc);
    if (c->d_scratch)
        (void)hipFree(c->d_scratch);
    if (c->h_scratch)
        (void)hipHostFree(c->h_scratch);
    if (c->d_twod)
        (void)hipFree(c->d_twod);
    if (c->d_nodes)
        (void)hipFree(c->d_nodes);
    if (c->d_bound)
        (void)hipFree(c->d_bound);
    if (c->d_stats)
        (void)hipFree(c->d_stats);
    if (c->ev0)
        (void)hipEventDestroy(c->ev0);
    if (c->ev1)
        (void)hipEventDestroy(c->ev1);
    for (int g = 0; g < HYAS_MAX_GROUPS; ++g)
    {
        for (int k = 0; k < 2; ++k)
            if (c->round_ev[g][k])
                (void)hipEventDestroy(c->round_ev[g][k]);
        if (g > 0 && c->round_stream[g])
            (void)hipStreamDestroy(c->round_stream[g]);
    }
    if (c->prep_stream)
        (void)hipStreamDestroy(c->prep_stream);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    mark("device memory, events and streams released");
    delete c;
    // the searches' host memory goes back to the system, unless another context still holds results or is planning
    (void)HugePool::instance().trim();
    mark("done");
}

const char *hyas_last_error(const hyas_ctx *c) { return c ? c->err.c_str() : "null context"; }

int hyas_num_successors(const hyas_ctx *c) { return c ? c->dm.n_slots : HYAS_EINVAL; }

static void fill_motion_set(hyas_ctx *c)
{
    // Utility::FormSteeringAngleVec (src/utility.cpp:2073-2089) followed by the
    // single straight reverse step (src/hybrid_a_star.cpp:629-654)
    DevMap &m = c->dm;
    int k = 0;
    m.succ_steer[k] = 0, m.succ_reverse[k] = 0, ++k;
    if (c->P.number_of_successors > 1)
        for (int i = 1; i <= (c->P.number_of_successors - 1) / 2; i++)
        {
            m.succ_steer[k] = i * deg2rad(c->P.steering_angle), m.succ_reverse[k] = 0, ++k;
            m.succ_steer[k] = -(i * deg2rad(c->P.steering_angle)), m.succ_reverse[k] = 0, ++k;
        }
    if (c->P.reverse)
        m.succ_steer[k] = 0, m.succ_reverse[k] = 1, ++k;
    m.n_succ = k;
    // CreateSuccessor's branch (src/hybrid_a_star.cpp:547-600): the combined mode wins over the step-only mode
    m.ad_mode = c->P.adaptive_steering_angle_and_step_size ? 1 : c->P.adaptive_step_size ? 2 : 0;
    m.n_slots = m.ad_mode == 1 ? HYAS_ADAPTIVE_SLOTS : k;
    m.ad_range = c->P.obstacle_detection_range, m.ad_constant_density = c->P.constant_density;
    m.ad_one_more = c->P.add_one_more_successor != 0, m.ad_one_more_free_only = c->P.add_one_more_successor_only_in_free_angle_range != 0;
    m.ad_towards_free = c->P.steering_angle_towards_free_angle_range_for_obstacle_angle_range != 0;
    m.ad_fixed_number = c->P.fixed_number_of_steering_angle_in_free_angle_range != 0;
    m.pen_turn = c->P.penalty_turning, m.pen_rev = c->P.penalty_reverse, m.pen_cod = c->P.penalty_change_of_direction;
    m.shot_rho = c->P.min_turning_radius;
    m.max_steer = c->P.max_steering_angle;
    m.curve_type = c->P.curve_type;
}

int hyas_table_dims(const hyas_ctx *c, int32_t *nx, int32_t *ny, int32_t *ns)
{
    if (!c || !c->have_map || c->grid_only)
        return HYAS_ESTATE;
    if (!nx || !ny || !ns)
        return HYAS_EINVAL;
    *nx = (int32_t)c->tab_xs.size(), *ny = (int32_t)c->tab_ys.size(), *ns = (int32_t)c->tab_ts.size();
    return HYAS_OK;
}

static bool known_curve(int curve) { return curve == HYAS_CURVE_RS || curve == HYAS_CURVE_DUBINS || curve == HYAS_CURVE_BEZIER; }
// the cost of a key the table does not hold: src/lookup_table.cpp:59 (Dubins), :85 (Reeds-Shepp), :101 (cubic Bezier)
static float table_miss_of(int curve) { return curve == HYAS_CURVE_RS ? 100000000.f : curve == HYAS_CURVE_BEZIER ? 10000000.f : 1000000.f; }

int hyas_table_build_dev(hyas_ctx *c, int curve, int k0, int k1, float *out_dev)
{
    if (!c || !c->have_map)
        return HYAS_ESTATE;
    if (c->grid_only)
        return c->fail(HYAS_ESTATE, "the heuristic table needs hyas_set_costmap, not hyas_set_grid");
    if (!known_curve(curve))
        return c->fail(HYAS_EINVAL, "curve must be 0 (Dubins), 1 (Reeds-Shepp) or 2 (cubic Bezier)");
    if (k0 < 0 || k1 > (int)c->tab_ts.size() || k0 > k1 || !out_dev)
        return c->fail(HYAS_EINVAL, "slice range outside the heading loop");
    HYAS_HIP(c, hipSetDevice(c->device));
    timer_begin(c);
    if (curve == HYAS_CURVE_BEZIER)
        launch_table_bezier(c->d_tab_xs, (uint32_t)c->tab_xs.size(), c->d_tab_ys, (uint32_t)c->tab_ys.size(), c->d_tab_ts, (uint32_t)k0, (uint32_t)k1,
                            out_dev, c->stream);
    else if (curve == HYAS_CURVE_RS)
        launch_table_rs(c->d_tab_xs, (uint32_t)c->tab_xs.size(), c->d_tab_ys, (uint32_t)c->tab_ys.size(), c->d_tab_ts, (uint32_t)k0,
                        (uint32_t)k1, (double)c->P.cd_min_turning_radius, out_dev, c->stream);
    else
        launch_table_dubins(c->d_tab_xs, (uint32_t)c->tab_xs.size(), c->d_tab_ys, (uint32_t)c->tab_ys.size(), c->d_tab_ts, (uint32_t)k0,
                            (uint32_t)k1, c->P.cd_min_turning_radius, out_dev, c->stream);
    HYAS_HIP(c, hipGetLastError());
    HYAS_HIP(c, hipStreamSynchronize(c->stream));
    timer_end(c, 3);
    return HYAS_OK;
}

int hyas_table_install_dev(hyas_ctx *c, int curve, const float *raw_dev)
{
    if (!c || !c->have_map)
        return HYAS_ESTATE;
    if (c->grid_only || !c->d_table)
        return c->fail(HYAS_ESTATE, "the heuristic table needs hyas_set_costmap, not hyas_set_grid");
    if (!known_curve(curve))
        return c->fail(HYAS_EINVAL, "curve must be 0 (Dubins), 1 (Reeds-Shepp) or 2 (cubic Bezier)");
    if (!raw_dev)
        return c->fail(HYAS_EINVAL, "null table");
    HYAS_HIP(c, hipSetDevice(c->device));
    uint32_t *rank = nullptr;
    HYAS_HIP(c, hipMalloc((void **)&rank, c->table_size * sizeof(uint32_t)));
    launch_table_install(c->dm, raw_dev, c->d_tab_xs, (uint32_t)c->tab_xs.size(), c->d_tab_ys, (uint32_t)c->tab_ys.size(), c->d_tab_ts,
                         (uint32_t)c->tab_ts.size(), rank, c->d_table, c->table_size, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    (void)hipFree(rank);
    HYAS_HIP(c, e);
    c->dm.table_miss = table_miss_of(curve);
    c->have_table = true;
    return HYAS_OK;
}

int hyas_table_install_noalloc_dev(hyas_ctx *c, int curve, const float *raw_dev)
{
    if (!c || !c->have_map)
        return HYAS_ESTATE;
    if (c->grid_only || !c->d_table)
        return c->fail(HYAS_ESTATE, "the heuristic table needs hyas_set_costmap, not hyas_set_grid");
    if (!known_curve(curve))
        return c->fail(HYAS_EINVAL, "curve must be 0 (Dubins), 1 (Reeds-Shepp) or 2 (cubic Bezier)");
    if (!raw_dev)
        return c->fail(HYAS_EINVAL, "null table");
    if (c->table_size > HYAS_NOALLOC_MAX_KEYS)
    {
        char buf[200];
        snprintf(buf, sizeof buf, "hyas_table_install_noalloc_dev: the table has %llu keys, the static scratch holds %u; use hyas_table_install_dev",
                 (unsigned long long)c->table_size, HYAS_NOALLOC_MAX_KEYS);
        return c->fail(HYAS_ERANGE, buf);
    }
    uint32_t *rank = c->device < HYAS_MAX_DEVICES ? scratch_ptr[c->device] : nullptr;
    if (!rank)
        return c->fail(HYAS_ENODEV, "no static scratch on this device");
    HYAS_HIP(c, hipSetDevice(c->device));
    // launch_table_install starts by setting every rank the table can address to "no position yet"
    launch_table_install(c->dm, raw_dev, c->d_tab_xs, (uint32_t)c->tab_xs.size(), c->d_tab_ys, (uint32_t)c->tab_ys.size(), c->d_tab_ts,
                         (uint32_t)c->tab_ts.size(), rank, c->d_table, c->table_size, c->stream);
    HYAS_HIP(c, hipGetLastError());
    HYAS_HIP(c, hipStreamSynchronize(c->stream));
    c->dm.table_miss = table_miss_of(curve);
    c->have_table = true;
    return HYAS_OK;
}

// The per-costmap part of the adaptive step size / steering modes: the hybrid_astar branch of
// CollisionDetection::UpdateGrid (src/collisiondetection.cpp:39-46).  For every grid key the minimum obstacle distance,
// the sweep of its owner corner (361 rays against the corner's in-range obstacles), the run-length pass over the sweep
// and the cut to the circle run in one kernel (k_adaptive_ranges: lengths, prefix sum, ranges); the density
// normalisation is two passes over the list lengths here.  owner[key] = the first loop position (x outer, y inner)
// that maps to a grid key: the reference's maps keep what that position emplaced.  The host keeps copies for
// hyas_adaptive_cell.
static int build_adaptive_maps(hyas_ctx *c, const float *d_xs, const float *d_ys, uint32_t nys, const uint32_t *d_owner, const std::vector<int64_t> &owner)
{
    DevMap &m = c->dm;
    const size_t cells = (size_t)c->w * c->h;
    const std::vector<float> angles = ad::sweep_angles();
    if (angles.size() > HYAS_SWEEP_MAX)
        return c->fail(HYAS_ERANGE, "more sweep angles than the kernel holds");
    const size_t slack = 64;
    float *d_ang = nullptr, *d_min = nullptr, *d_kept = nullptr;
    unsigned long long *d_total = nullptr, total = 0;
    // the first ranges of every key are kept between the two passes (a cell has three or four on average, rarely more
    // than eight), unless the map is so large that the room for them would be
    const uint32_t keep = std::getenv("HYAS_ADAPTIVE_KEEP") ? (uint32_t)std::max(0, std::min(21, std::atoi(std::getenv("HYAS_ADAPTIVE_KEEP")))) : 8u; // (test knob)
    const size_t kept_bytes = cells * keep * 3 * sizeof(float);
    int rc = upload(c, d_ang, angles);
    auto step = [&](hipError_t err, const char *what) {
        if (rc == HYAS_OK && err != hipSuccess)
            rc = hip_fail(c, err, what);
        return rc == HYAS_OK;
    };
    if (rc == HYAS_OK)
        step(hipMalloc((void **)&d_min, std::max<size_t>(cells, 1) * sizeof(float)), "hipMalloc(minimum distances)");
    if (rc == HYAS_OK)
        step(hipMalloc((void **)&d_total, sizeof(unsigned long long)), "hipMalloc(range total)");
    if (rc == HYAS_OK && kept_bytes != 0 && kept_bytes <= ((size_t)512 << 20) && hipMalloc((void **)&d_kept, kept_bytes) != hipSuccess)
    {
        (void)hipGetLastError(); // (without the room both passes sweep)
        d_kept = nullptr;
    }
    if (rc == HYAS_OK)
        step(hipMalloc((void **)&c->d_ad_off, (cells + 1) * sizeof(uint32_t) + slack), "hipMalloc(range offsets)");
    if (rc == HYAS_OK)
        step(hipMemsetAsync((char *)c->d_ad_off + (cells + 1) * sizeof(uint32_t), 0, slack, c->stream), "range offsets");
    if (rc == HYAS_OK)
    {
        launch_adaptive_ranges(m, d_xs, d_ys, nys, d_owner, d_ang, (uint32_t)angles.size(), c->P.obstacle_detection_range, d_min, c->d_ad_off, nullptr,
                               d_kept, keep, d_total, 0, c->stream);
        step(hipGetLastError(), "angle range lengths");
    }
    if (rc == HYAS_OK)
        step(hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, c->stream), "angle range lengths");
    if (rc == HYAS_OK)
        step(hipStreamSynchronize(c->stream), "angle range lengths");
    if (rc == HYAS_OK && total >= (1ull << 32) / 3)
        rc = c->fail(HYAS_ERANGE, "too many angle ranges");
    if (rc == HYAS_OK)
        step(hipMalloc((void **)&c->d_ad_rng, std::max<size_t>((size_t)total, 1) * 3 * sizeof(float) + slack), "hipMalloc(angle ranges)");
    if (rc == HYAS_OK)
        step(hipMemsetAsync((char *)c->d_ad_rng + (size_t)total * 3 * sizeof(float), 0, slack + (total ? 0 : 3 * sizeof(float)), c->stream), "angle ranges");
    if (rc == HYAS_OK)
    {
        launch_adaptive_ranges(m, d_xs, d_ys, nys, d_owner, d_ang, (uint32_t)angles.size(), c->P.obstacle_detection_range, d_min, c->d_ad_off,
                               c->d_ad_rng, d_kept, keep, d_total, 1, c->stream);
        step(hipGetLastError(), "angle ranges");
    }
    c->ad_min_distance.assign(cells, 1000.f);
    c->ad_off.assign(cells + 1, 0);
    c->ad_rng.assign((size_t)total * 3, 0.f);
    if (rc == HYAS_OK)
        step(hipMemcpyAsync(c->ad_min_distance.data(), d_min, cells * sizeof(float), hipMemcpyDeviceToHost, c->stream), "minimum distances");
    if (rc == HYAS_OK)
        step(hipMemcpyAsync(c->ad_off.data(), c->d_ad_off, (cells + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream), "angle range offsets");
    if (rc == HYAS_OK && total)
        step(hipMemcpyAsync(c->ad_rng.data(), c->d_ad_rng, (size_t)total * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream), "angle ranges");
    if (rc == HYAS_OK)
        step(hipStreamSynchronize(c->stream), "angle ranges");
    for (void *p : {(void *)d_ang, (void *)d_min, (void *)d_total, (void *)d_kept})
        if (p)
            (void)hipFree(p);
    if (rc != HYAS_OK)
        return rc;
    // BuildObstacleDensityMap (the per-cell count, :1034-1049) and BuildNormalizedObstacleDensityMap (:1110-1151)
    float max_density = 0, min_density = 1000000;
    for (size_t key = 0; key < cells; ++key)
        if (owner[key] >= 0)
        {
            const float d = (float)(c->cell_off[key + 1] - c->cell_off[key]);
            if (d > max_density)
                max_density = d;
            if (d < min_density)
                min_density = d;
        }
    const float delta = 0.01f;
    c->ad_density.assign(cells, 0.f);
    for (size_t key = 0; key < cells; ++key)
        if (owner[key] >= 0)
            c->ad_density[key] = ((float)(c->cell_off[key + 1] - c->cell_off[key]) - min_density) / (max_density - min_density + delta);
    if ((rc = upload(c, c->d_ad_density, c->ad_density)) != HYAS_OK)
        return rc;
    m.ad_off = c->d_ad_off, m.ad_rng = c->d_ad_rng, m.ad_density = c->d_ad_density;
    return HYAS_OK;
}

// The adaptive maps of one grid key (0 <= key < w * h), for inspection: minimum obstacle distance, normalised density
// and up to cap {label distance, start, width} angle ranges as FindFreeAngleRangeAndObstacleAngleRange returns them
// for a node in that cell.  Returns the number of ranges.
int hyas_adaptive_cell(hyas_ctx *c, uint32_t key, float *min_distance, float *density, float *ranges3, size_t cap)
{
    if (!c || !c->have_map || c->grid_only)
        return HYAS_ESTATE;
    if (c->dm.ad_mode == 0)
        return c->fail(HYAS_ESTATE, "the context was created without an adaptive mode");
    if ((uint64_t)key >= (uint64_t)c->w * c->h)
        return c->fail(HYAS_EINVAL, "grid key outside the map");
    if (min_distance)
        *min_distance = c->ad_min_distance[key];
    if (density)
        *density = c->ad_density[key];
    const uint32_t b = c->ad_off[key], n = c->ad_off[key + 1] - b;
    if (ranges3)
        std::memcpy(ranges3, &c->ad_rng[3 * (size_t)b], std::min<size_t>(n, cap) * 3 * sizeof(float));
    return (int)n;
}

int hyas_set_costmap(hyas_ctx *c, const uint8_t *charmap, uint32_t w, uint32_t h, double res, double ox, double oy, int build_table)
{
    if (!c)
        return HYAS_EINVAL;
    if (!charmap || w == 0 || h == 0 || !(res > 0))
        return c->fail(HYAS_EINVAL, "empty costmap or non-positive resolution");
    if ((uint64_t)w * h >= (1u << 24))
        return c->fail(HYAS_ERANGE, "polygon-mode indices are float arithmetic (include/node3d.h:67): w*h must stay below 2^24; "
                                    "use hyas_set_grid for footprint-only grids");
    HYAS_HIP(c, hipSetDevice(c->device));
    free_map(c);
    c->grid_only = false;
    c->w = w, c->h = h, c->res = res, c->ox = ox, c->oy = oy;
    c->charmap.assign(charmap, charmap + (size_t)w * h);
    c->twod_size = (size_t)w * h + w + 1;

    DevMap &m = c->dm;
    m = DevMap{};
    m.w = w, m.h = h;
    set_resolution(m, res, ox, oy);
    m.veh_len = c->P.vehicle_length, m.veh_wid = c->P.vehicle_width;
    m.half_wid = 0.5 * (double)c->P.vehicle_width;
    // the on-line / on-segment snaps act within 1e-3 rad or 1e-3 m of an edge
    // no longer than the map diagonal
    m.cull_margin = (float)(0.01 + 2e-3 * std::sqrt((double)w * w + (double)h * h) * res);
    if (c->P.map_boundary_obstacle) // the boundary rectangles' corners lie up to a rectangle's length beyond the detection range
        m.cull_margin = (float)(0.01 + 2e-3 * (std::sqrt((double)w * w + (double)h * h) * res + (double)w + (double)h + 4 +
                                                 std::max(0.0, (double)c->P.obstacle_detection_range)));
    m.pr = c->P.position_resolution, m.headings = c->P.headings;
    m.twod_size = (uint32_t)c->twod_size;
    m.delta_heading = (float)(2 * HY_PI / (double)(float)c->P.headings); // src/lookup_table.cpp:127
    fill_motion_set(c);

    // HYAS_TRACE: where the time of the ingest goes
    const bool trace_ingest = std::getenv("HYAS_TRACE") != nullptr;
    auto t_mark = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!trace_ingest)
            return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "hyas ingest: %-44s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_mark).count());
        t_mark = now;
    };
    // obstacles + per-key lists
    std::vector<float> xs, ys;
    cell_axes(c, xs, ys);
    int rc;
    if ((rc = upload(c, c->d_charmap, c->charmap)) != HYAS_OK)
        return rc;
    const size_t cells = (size_t)w * h;
    std::vector<int8_t> occ(cells);
    {
        // costmap -> occupancy values on the GPU (src/utility.cpp:185-204)
        int8_t *d_occ = nullptr;
        HYAS_HIP(c, hipMalloc((void **)&d_occ, cells));
        launch_occupancy(c->d_charmap, cells, d_occ, c->stream);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess)
            e = hipMemcpyAsync(occ.data(), d_occ, cells, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(c->stream);
        (void)hipFree(d_occ);
        if (e != hipSuccess)
            return hip_fail(c, e, "occupancy values");
    }
    mark("cost bytes up, occupancy values back");
    extract_obstacles(c, occ, xs, ys);
    mark("rectangle scan (host)");
    if (c->quads.size() > 32767)
        return c->fail(HYAS_ERANGE, "more than 32767 obstacle polygons");
    if ((rc = upload(c, c->d_quads, c->quads)) != HYAS_OK)
        return rc;
    c->cell_off.assign(cells + 1, 0);
    c->cell_idx.clear();
    // in_range_obstacle_map_.emplace(key, list): the first loop position that maps to a key keeps it
    // (src/collisiondetection.cpp:345); -1: no corner maps to the key
    std::vector<int64_t> owner(cells, -1);
    float *d_xs = nullptr, *d_ys = nullptr; // the corner axes and the keys' owners on the device: the adaptive maps use them too
    uint32_t *d_owner = nullptr;
    struct FreeAxes
    {
        float *&xs, *&ys;
        uint32_t *&owner;
        ~FreeAxes()
        {
            for (void *p : {(void *)xs, (void *)ys, (void *)owner})
                if (p)
                    (void)hipFree(p);
        }
    } free_axes{d_xs, d_ys, d_owner};
    {
        // Lists on the GPU in two passes over (key, polygon): lengths, prefix sum, ids.  Nothing of size
        // cells x polygons is ever stored.
        const size_t slack = 64; // kernels fetch list entries in 16-byte blocks, one block ahead
        const size_t nq = c->quads.size();
        unsigned long long *d_total = nullptr, total = 0;
        std::vector<uint32_t> owner_u(cells);
        rc = upload(c, d_xs, xs);
        if (rc == HYAS_OK)
            rc = upload(c, d_ys, ys);
        auto step = [&](hipError_t err, const char *what) {
            if (rc == HYAS_OK && err != hipSuccess)
                rc = hip_fail(c, err, what);
            return rc == HYAS_OK;
        };
        if (rc == HYAS_OK)
            step(hipMalloc((void **)&d_owner, cells * sizeof(uint32_t)), "hipMalloc(list owners)");
        if (rc == HYAS_OK)
            step(hipMalloc((void **)&d_total, sizeof(unsigned long long)), "hipMalloc(list total)");
        if (rc == HYAS_OK)
            step(hipMalloc((void **)&c->d_cell_off, (cells + 1) * sizeof(uint32_t) + slack), "hipMalloc(list offsets)");
        if (rc == HYAS_OK)
            step(hipMemsetAsync((char *)c->d_cell_off + (cells + 1) * sizeof(uint32_t), 0, slack, c->stream), "list offsets");
        // bins of polygons per tile of grid keys first (lengths, prefix sum, ids), then the keys' lists out of them
        RangeBins bins{};
        bins.tiles_x = ((uint32_t)w + HYAS_BIN_TILE - 1) / HYAS_BIN_TILE, bins.tiles_y = ((uint32_t)h + HYAS_BIN_TILE - 1) / HYAS_BIN_TILE;
        bins.extra = std::getenv("HYAS_ALL_POLYGONS_IN_BINS") ? INFINITY : 0.f; // test aid: every key is tested against every polygon
        const size_t tiles = (size_t)bins.tiles_x * bins.tiles_y;
        unsigned long long bin_total = 0;
        if (rc == HYAS_OK)
            step(hipMalloc((void **)&bins.off, (tiles + 1) * sizeof(uint32_t)), "hipMalloc(bin offsets)");
        if (rc == HYAS_OK)
        {
            launch_range_bins(m, c->d_quads, (uint32_t)nq, d_xs, (uint32_t)xs.size(), d_ys, (uint32_t)ys.size(), c->P.obstacle_detection_range, d_owner, bins,
                              d_total, 0, c->stream);
            step(hipGetLastError(), "polygon bin lengths");
        }
        if (rc == HYAS_OK)
            step(hipMemcpyAsync(&bin_total, d_total, sizeof(bin_total), hipMemcpyDeviceToHost, c->stream), "polygon bin lengths");
        if (rc == HYAS_OK)
            step(hipStreamSynchronize(c->stream), "polygon bin lengths");
        if (rc == HYAS_OK && bin_total >= (1ull << 32))
            rc = c->fail(HYAS_ERANGE, "the polygon bins hold 2^32 entries or more");
        if (rc == HYAS_OK)
            step(hipMalloc((void **)&bins.idx, ((size_t)bin_total + 1) * sizeof(uint16_t)), "hipMalloc(bin entries)");
        if (rc == HYAS_OK)
        {
            launch_range_bins(m, c->d_quads, (uint32_t)nq, d_xs, (uint32_t)xs.size(), d_ys, (uint32_t)ys.size(), c->P.obstacle_detection_range, d_owner, bins,
                              d_total, 1, c->stream);
            launch_in_range_lists(m, c->d_quads, d_xs, d_ys, (uint32_t)ys.size(), c->P.obstacle_detection_range, d_owner, bins, c->d_cell_off, nullptr, d_total,
                                  0, c->stream);
            step(hipGetLastError(), "in-range list lengths");
        }
        if (rc == HYAS_OK)
            step(hipMemcpyAsync(&total, d_total, sizeof(total), hipMemcpyDeviceToHost, c->stream), "in-range list lengths");
        if (rc == HYAS_OK)
            step(hipStreamSynchronize(c->stream), "in-range list lengths");
        if (rc == HYAS_OK && total >= (1ull << 32))
            rc = c->fail(HYAS_ERANGE, "the in-range obstacle lists hold 2^32 entries or more");
        if (rc == HYAS_OK)
            step(hipMalloc((void **)&c->d_cell_idx, (size_t)total * sizeof(uint16_t) + slack), "hipMalloc(list entries)");
        if (rc == HYAS_OK)
            step(hipMemsetAsync((char *)c->d_cell_idx + (size_t)total * sizeof(uint16_t), 0, slack, c->stream), "list entries");
        if (rc == HYAS_OK)
        {
            launch_in_range_lists(m, c->d_quads, d_xs, d_ys, (uint32_t)ys.size(), c->P.obstacle_detection_range, d_owner, bins, c->d_cell_off, c->d_cell_idx,
                                  d_total, 1, c->stream);
            step(hipGetLastError(), "in-range lists");
        }
        // the host keeps copies: list pruning, the adaptive maps and the getters read them
        c->cell_idx.resize((size_t)total);
        if (rc == HYAS_OK)
            step(hipMemcpyAsync(c->cell_off.data(), c->d_cell_off, (cells + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream), "in-range lists");
        if (rc == HYAS_OK && total)
            step(hipMemcpyAsync(c->cell_idx.data(), c->d_cell_idx, (size_t)total * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream), "in-range lists");
        if (rc == HYAS_OK)
            step(hipMemcpyAsync(owner_u.data(), d_owner, cells * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream), "in-range lists");
        if (rc == HYAS_OK)
            step(hipStreamSynchronize(c->stream), "in-range lists");
        for (void *p : {(void *)d_total, (void *)bins.off, (void *)bins.idx})
            if (p)
                (void)hipFree(p);
        if (rc != HYAS_OK)
            return rc;
        for (size_t key = 0; key < cells; ++key)
            owner[key] = owner_u[key] == 0xFFFFFFFFu ? -1 : (int64_t)owner_u[key];
    }
    mark("polygon bins and in-range lists, copies back");
    m.charmap = c->d_charmap, m.quads = c->d_quads, m.n_quads = (uint32_t)c->quads.size();
    m.cell_off = c->d_cell_off, m.cell_idx = c->d_cell_idx;
    if (m.ad_mode != 0 && (rc = build_adaptive_maps(c, d_xs, d_ys, (uint32_t)ys.size(), d_owner, owner)) != HYAS_OK)
        return rc;
    mark("adaptive maps");
    {
        // successor steps are 0.2 m, curve samples 0.1 x resolution apart
        const float lmax = std::max(0.3f, 0.2f * (float)res);
        // pruned lists on the GPU, straight from the full lists in HBM: lengths, prefix sums, ids; neither list can be
        // longer than the full one, so the ids are allocated before their number is known
        const size_t slack = 64, full = c->cell_idx.size();
        const PruneMargins pm = prune_margins(m, lmax);
        PruneArgs pa{};
        pa.res = m.res, pa.ox = m.ox, pa.oy = m.oy, pa.diag = pm.diag, pa.reach = pm.reach, pa.lmax = (double)lmax, pa.slack = pm.slack;
        pa.w = m.w, pa.cells = (uint32_t)cells;
        unsigned long long *d_totals = nullptr, totals[2] = {0, 0};
        HYAS_HIP(c, hipMalloc((void **)&d_totals, 2 * sizeof(unsigned long long)));
        hipError_t e = hipMalloc((void **)&c->d_pose_off, (cells + 1) * sizeof(uint32_t) + slack);
        if (e == hipSuccess)
            e = hipMalloc((void **)&c->d_seg_off, (cells + 1) * sizeof(uint32_t) + slack);
        if (e == hipSuccess)
            e = hipMalloc((void **)&c->d_pose_idx, std::max<size_t>(full, 1) * sizeof(uint16_t) + slack);
        if (e == hipSuccess)
            e = hipMalloc((void **)&c->d_seg_idx, std::max<size_t>(full, 1) * sizeof(uint16_t) + slack);
        if (e == hipSuccess)
            e = hipMemsetAsync((char *)c->d_pose_off + (cells + 1) * sizeof(uint32_t), 0, slack, c->stream);
        if (e == hipSuccess)
            e = hipMemsetAsync((char *)c->d_seg_off + (cells + 1) * sizeof(uint32_t), 0, slack, c->stream);
        if (e == hipSuccess)
            e = hipMemsetAsync(c->d_pose_idx, 0, std::max<size_t>(full, 1) * sizeof(uint16_t) + slack, c->stream);
        if (e == hipSuccess)
            e = hipMemsetAsync(c->d_seg_idx, 0, std::max<size_t>(full, 1) * sizeof(uint16_t) + slack, c->stream);
        if (e == hipSuccess)
        {
            launch_prune_lists(c->d_quads, c->d_cell_off, c->d_cell_idx, pa, c->d_pose_off, c->d_pose_idx, c->d_seg_off, c->d_seg_idx, d_totals, 0, c->stream);
            launch_prune_lists(c->d_quads, c->d_cell_off, c->d_cell_idx, pa, c->d_pose_off, c->d_pose_idx, c->d_seg_off, c->d_seg_idx, d_totals, 1, c->stream);
            e = hipGetLastError();
        }
        if (e == hipSuccess)
            e = hipMemcpyAsync(totals, d_totals, sizeof totals, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(c->stream);
        (void)hipFree(d_totals);
        if (e != hipSuccess)
            return hip_fail(c, e, "list pruning");
        c->n_pose_entries = (size_t)totals[0], c->n_seg_entries = (size_t)totals[1];
        if (std::getenv("HYAS_CHECK_PRUNE")) // test aid: the host form of the same rules must give the same lists
        {
            std::vector<uint32_t> pose_off, seg_off, g_pose_off(cells + 1), g_seg_off(cells + 1);
            std::vector<uint16_t> pose_idx, seg_idx;
            prune_lists(m, c->quads.data(), c->cell_off.data(), c->cell_idx.data(), lmax, pose_off, pose_idx, seg_off, seg_idx);
            std::vector<uint16_t> g_pose_idx(pose_idx.size()), g_seg_idx(seg_idx.size());
            bool same = pose_idx.size() == totals[0] && seg_idx.size() == totals[1];
            if (same)
            {
                HYAS_HIP(c, hipMemcpy(g_pose_off.data(), c->d_pose_off, (cells + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
                HYAS_HIP(c, hipMemcpy(g_seg_off.data(), c->d_seg_off, (cells + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
                if (!pose_idx.empty())
                    HYAS_HIP(c, hipMemcpy(g_pose_idx.data(), c->d_pose_idx, pose_idx.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
                if (!seg_idx.empty())
                    HYAS_HIP(c, hipMemcpy(g_seg_idx.data(), c->d_seg_idx, seg_idx.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
                same = g_pose_off == pose_off && g_seg_off == seg_off && g_pose_idx == pose_idx && g_seg_idx == seg_idx;
            }
            if (!same)
                return c->fail(HYAS_ESTATE, "HYAS_CHECK_PRUNE: the pruned lists of the GPU differ from the host form's");
        }
        m.seg_lmax = std::getenv("HYAS_NO_PRUNE") ? 0.f : lmax;
    }

    mark("list pruning");
    // footprint lists
    fpt::build(c->P, c->fp);
    if ((rc = upload(c, c->d_fp, c->fp)) != HYAS_OK)
        return rc;
    mark("footprint table (host) and upload");

    // heuristic-table axes: the float accumulators of src/lookup_table.cpp:197-222
    c->tab_xs.clear(), c->tab_ys.clear(), c->tab_ts.clear();
    for (float x = 0; x < (float)(int)w; x += 1.0f / c->P.position_resolution)
        c->tab_xs.push_back(x);
    for (float y = 0; y < (float)(int)h; y += 1.0f / c->P.position_resolution)
        c->tab_ys.push_back(y);
    for (float th = 0; (double)th < 2 * HY_PI; th += m.delta_heading)
        c->tab_ts.push_back(th);
    if ((rc = upload(c, c->d_tab_xs, c->tab_xs)) != HYAS_OK || (rc = upload(c, c->d_tab_ys, c->tab_ys)) != HYAS_OK ||
        (rc = upload(c, c->d_tab_ts, c->tab_ts)) != HYAS_OK)
        return rc;
    c->table_size = (uint64_t)(c->P.headings + 1) * w * h * c->P.position_resolution * c->P.position_resolution +
                    (uint64_t)w * c->P.position_resolution;
    if (c->table_size >= (1ull << 31))
        return c->fail(HYAS_ERANGE, "heuristic table keys exceed int range (src/lookup_table.cpp:143)");
    HYAS_HIP(c, hipMalloc((void **)&c->d_table, c->table_size * sizeof(float)));

    // arc-length samples of the shot: seg after k additions of 0.1*resolution
    // (src/ReedsShepp.cpp:566, src/hybrid_a_star.cpp:439)
    {
        const double step = 0.1 * (double)m.res_f;
        const double longest = 8.0 * (std::sqrt((double)w * w + (double)h * h) * res + 2 * HY_PI * (double)c->P.min_turning_radius);
        size_t cap = (size_t)(longest / step) + 16;
        cap = std::min<size_t>(std::max<size_t>(cap, 1024), (size_t)1 << 22);
        std::vector<double> seg(cap);
        double s = 0;
        for (size_t k = 0; k < cap; ++k, s += step)
            seg[k] = s;
        c->seg_cap = (uint32_t)cap;
        if ((rc = upload(c, c->d_seg, seg)) != HYAS_OK)
            return rc;
        // Dubins shot: x after k additions of 0.2*resolution, float with the sum formed in double
        // (src/hybrid_a_star.cpp:396-424)
        // and the cubic Bezier shot: the same with 0.1*resolution (src/hybrid_a_star.cpp:477-499), behind it in one buffer
        const size_t cap_db = cap / 2 + 16;
        std::vector<float> seg_db(cap_db + cap);
        float x = 0.f;
        for (size_t k = 0; k < cap_db; ++k)
        {
            seg_db[k] = x;
            x = (float)((double)x + 0.2 * (double)m.res_f);
        }
        x = 0.f;
        for (size_t k = 0; k < cap; ++k)
        {
            seg_db[cap_db + k] = x;
            x = (float)((double)x + 0.1 * (double)m.res_f);
        }
        if ((rc = upload(c, c->d_seg_db, seg_db)) != HYAS_OK)
            return rc;
        m.seg_db = c->d_seg_db, m.seg_db_cap = (uint32_t)cap_db;
        m.seg_bz = c->d_seg_db + cap_db, m.seg_bz_cap = (uint32_t)cap;
    }

    m.charmap = c->d_charmap, m.quads = c->d_quads, m.n_quads = (uint32_t)c->quads.size();
    m.cell_off = c->d_cell_off, m.cell_idx = c->d_cell_idx, m.fp = c->d_fp;
    if ((rc = upload_footprint_forms(c)) != HYAS_OK)
        return rc;
    m.pose_off = c->d_pose_off, m.pose_idx = c->d_pose_idx, m.seg_off = c->d_seg_off, m.seg_idx = c->d_seg_idx;
    m.table = c->d_table, m.table_size = c->table_size;
    m.table_miss = table_miss_of(c->P.curve_type);
    c->have_map = true;
    mark("table axes, shot samples, footprint forms");

    if (build_table)
    {
        const size_t total = c->tab_xs.size() * c->tab_ys.size() * c->tab_ts.size();
        float *raw = nullptr;
        HYAS_HIP(c, hipMalloc((void **)&raw, std::max<size_t>(total, 1) * sizeof(float)));
        rc = hyas_table_build_dev(c, c->P.curve_type, 0, (int)c->tab_ts.size(), raw);
        if (rc == HYAS_OK)
            rc = hyas_table_install_dev(c, c->P.curve_type, raw);
        (void)hipFree(raw);
        if (rc != HYAS_OK)
            return rc;
    }
    return HYAS_OK;
}

// Footprint-list mode only: a bare occupancy grid of any size (no obstacle
// polygons, no heuristic table; resolution 1, origin 0, as the footprint table
// assumes unit cells).
int hyas_set_grid(hyas_ctx *c, const uint8_t *charmap, uint32_t w, uint32_t h)
{
    if (!c)
        return HYAS_EINVAL;
    if (!charmap || w == 0 || h == 0)
        return c->fail(HYAS_EINVAL, "empty grid");
    HYAS_HIP(c, hipSetDevice(c->device));
    free_map(c);
    c->w = w, c->h = h, c->res = 1, c->ox = 0, c->oy = 0;
    c->charmap.assign(charmap, charmap + (size_t)w * h);
    c->quads.clear(), c->cell_off.clear(), c->cell_idx.clear();
    c->twod_size = 0;
    DevMap &m = c->dm;
    m = DevMap{};
    m.w = w, m.h = h;
    set_resolution(m, 1.0, 0.0, 0.0);
    m.pr = c->P.position_resolution, m.headings = c->P.headings;
    m.delta_heading = (float)(2 * HY_PI / (double)(float)c->P.headings);
    int rc;
    if ((rc = upload(c, c->d_charmap, c->charmap)) != HYAS_OK)
        return rc;
    fpt::build(c->P, c->fp);
    if ((rc = upload(c, c->d_fp, c->fp)) != HYAS_OK)
        return rc;
    m.charmap = c->d_charmap, m.fp = c->d_fp;
    if ((rc = upload_footprint_forms(c)) != HYAS_OK)
        return rc;
    c->have_map = true, c->have_table = false;
    c->grid_only = true;
    return HYAS_OK;
}

int hyas_footprint_table(hyas_ctx *c, int32_t *out, size_t cap)
{
    if (!c || !c->have_map)
        return HYAS_ESTATE;
    size_t n = c->fp.size() / 129;
    if (out)
        std::memcpy(out, c->fp.data(), std::min(cap, n) * 129 * sizeof(int32_t));
    return (int)n;
}

int hyas_collision_batch_dev(hyas_ctx *c, int mode, const float *xyt, const float *pred_xy, size_t n, uint8_t *out)
{
    if (!c || !c->have_map)
        return HYAS_ESTATE;
    if (mode != HYAS_MODE_POLYGON && mode != HYAS_MODE_FOOTPRINT)
        return c->fail(HYAS_EINVAL, "unknown collision mode");
    if (mode == HYAS_MODE_POLYGON && c->grid_only)
        return c->fail(HYAS_ESTATE, "polygon mode needs hyas_set_costmap, not hyas_set_grid");
    if (n == 0)
        return HYAS_OK;
    if (!xyt || !out)
        return c->fail(HYAS_EINVAL, "null buffer");
    HYAS_HIP(c, hipSetDevice(c->device));
    timer_begin(c);
    launch_collision(c->dm, mode, xyt, pred_xy, n, out, c->stream);
    HYAS_HIP(c, hipGetLastError());
    HYAS_HIP(c, hipStreamSynchronize(c->stream));
    timer_end(c, 2);
    return HYAS_OK;
}

int hyas_collision_batch(hyas_ctx *c, int mode, const float *xyt, const float *pred_xy, size_t n, uint8_t *out)
{
    if (!c || !c->have_map)
        return HYAS_ESTATE;
    if (n == 0)
        return HYAS_OK;
    if (!xyt || !out)
        return c->fail(HYAS_EINVAL, "null buffer");
    HYAS_HIP(c, hipSetDevice(c->device));
    const size_t b_xyt = n * 3 * sizeof(float), b_pred = pred_xy ? n * 2 * sizeof(float) : 0, b_out = (n + 3) & ~(size_t)3;
    int rc = ensure_scratch(c, b_xyt + b_pred + b_out + 64, 0);
    if (rc != HYAS_OK)
        return rc;
    char *d = (char *)c->d_scratch;
    float *d_xyt = (float *)d;
    float *d_pred = pred_xy ? (float *)(d + b_xyt) : nullptr;
    uint8_t *d_out = (uint8_t *)(d + b_xyt + b_pred);
    HYAS_HIP(c, hipMemcpyAsync(d_xyt, xyt, b_xyt, hipMemcpyHostToDevice, c->stream));
    if (pred_xy)
        HYAS_HIP(c, hipMemcpyAsync(d_pred, pred_xy, b_pred, hipMemcpyHostToDevice, c->stream));
    rc = hyas_collision_batch_dev(c, mode, d_xyt, d_pred, n, d_out);
    if (rc != HYAS_OK)
        return rc;
    HYAS_HIP(c, hipMemcpy(out, d_out, n, hipMemcpyDeviceToHost));
    return HYAS_OK;
}

int hyas_table_lookup(hyas_ctx *c, const float *xyt, size_t n, float *cost)
{
    if (!c || !c->have_map || !c->have_table)
        return HYAS_ESTATE;
    if (n == 0)
        return HYAS_OK;
    if (!xyt || !cost)
        return c->fail(HYAS_EINVAL, "null buffer");
    HYAS_HIP(c, hipSetDevice(c->device));
    int rc = ensure_scratch(c, n * 4 * sizeof(float), 0);
    if (rc != HYAS_OK)
        return rc;
    float *d_xyt = (float *)c->d_scratch, *d_cost = d_xyt + 3 * n;
    HYAS_HIP(c, hipMemcpyAsync(d_xyt, xyt, n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    launch_table_lookup(c->dm, d_xyt, n, d_cost, c->stream);
    HYAS_HIP(c, hipGetLastError());
    HYAS_HIP(c, hipMemcpyAsync(cost, d_cost, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HYAS_HIP(c, hipStreamSynchronize(c->stream));
    return HYAS_OK;
}

int hyas_twod_size(const hyas_ctx *c) { return (c && c->have_map) ? (int)c->twod_size : HYAS_ESTATE; }

int hyas_flood(hyas_ctx *c, const hyas_goal *goal, float *out)
{
    if (!c || !c->have_map || c->grid_only)
        return HYAS_ESTATE;
    if (!goal || !out)
        return c->fail(HYAS_EINVAL, "null buffer");
    HYAS_HIP(c, hipSetDevice(c->device));
    const size_t N = c->twod_size;
    int rc = ensure_twod(c, N * sizeof(float) + 256);
    if (rc != HYAS_OK)
        return rc;
    // one query {start (unused), goal}; the poses sit behind the cost map in the same buffer
    float *d_pose = (float *)((char *)c->d_twod + ((N * sizeof(float) + 63) & ~(size_t)63));
    const float pose[6] = {0, 0, 0, goal->x, goal->y, goal->t};
    HYAS_HIP(c, hipMemcpyAsync(d_pose, pose, sizeof pose, hipMemcpyHostToDevice, c->stream));
    rc = flood_dev(c, d_pose, nullptr, 1, c->d_twod, nullptr, c->stream);
    if (rc != HYAS_OK)
        return rc;
    HYAS_HIP(c, hipMemcpyAsync(out, c->d_twod, N * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HYAS_HIP(c, hipStreamSynchronize(c->stream));
    return HYAS_OK;
}

// Shared body of hyas_expand_batch / hyas_shot_batch: one kernel launch over
// n independent (node, goal) pairs.
static int run_step(hyas_ctx *c, const hyas_node *nodes, const hyas_goal *goals, const hyas_goal *starts, const float *twod, size_t n, int shoot, uint8_t *ok,
                    float *len, hyas_succ *succ, uint8_t *count)
{
    if (!c || !c->have_map || !c->have_table)
        return HYAS_ESTATE;
    if (n == 0)
        return HYAS_OK;
    if (n > 0x7fffffffu)
        return c->fail(HYAS_ERANGE, "too many nodes in one call");
    HYAS_HIP(c, hipSetDevice(c->device));
    const int K = c->dm.n_slots;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_in = take(n * sizeof(StepIn)), o_goal = take(n * sizeof(hyas_goal)), o_succ = take(n * K * sizeof(hyas_succ)),
                 o_len = take(n * sizeof(float)), o_ok = take(n), o_cnt = take(n);
    int rc = ensure_scratch(c, off, off);
    if (rc != HYAS_OK)
        return rc;
    if (twod && (rc = ensure_twod(c, n * c->twod_size * sizeof(float))) != HYAS_OK)
        return rc;
    char *hs = (char *)c->h_scratch, *ds = (char *)c->d_scratch;
    StepIn *h_in = (StepIn *)(hs + o_in);
    for (size_t i = 0; i < n; ++i)
    {
        uint32_t dsg = 0;
        if (starts) // Utility::GetDistance(start, goal)
        {
            const float d = normf(sub(f2(starts[i].x, starts[i].y), f2(goals[i].x, goals[i].y)));
            std::memcpy(&dsg, &d, sizeof d);
        }
        h_in[i] = StepIn{nodes[i].x, nodes[i].y, nodes[i].t, nodes[i].g, nodes[i].prim, (uint32_t)i, (uint32_t)shoot, dsg};
    }
    std::memcpy(hs + o_goal, goals, n * sizeof(hyas_goal));
    HYAS_HIP(c, hipMemcpyAsync(ds + o_in, hs + o_in, o_succ - o_in, hipMemcpyHostToDevice, c->stream));
    if (twod)
        HYAS_HIP(c, hipMemcpyAsync(c->d_twod, twod, n * c->twod_size * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HYAS_HIP(c, hipMemsetAsync(ds + o_succ, 0, off - o_succ, c->stream));
    timer_begin(c);
    if (shoot)
        launch_shot(c->dm, (const StepIn *)(ds + o_in), (uint32_t)n, (const hyas_goal *)(ds + o_goal), c->d_seg, c->seg_cap,
                       (uint8_t *)(ds + o_ok), (float *)(ds + o_len), c->stats_on ? c->d_stats : nullptr, c->stats_mode, c->stream);
    else
        launch_expand(c->dm, (const StepIn *)(ds + o_in), (uint32_t)n, (const hyas_goal *)(ds + o_goal), twod ? c->d_twod : nullptr,
                      (uint8_t *)(ds + o_cnt), (hyas_succ *)(ds + o_succ), c->stats_on ? c->d_stats : nullptr, c->stats_mode, c->stream);
    HYAS_HIP(c, hipGetLastError());
    HYAS_HIP(c, hipStreamSynchronize(c->stream));
    timer_end(c, shoot ? 0 : 1);
    HYAS_HIP(c, hipMemcpy(hs + o_succ, ds + o_succ, off - o_succ, hipMemcpyDeviceToHost));
    if (succ)
        std::memcpy(succ, hs + o_succ, n * K * sizeof(hyas_succ));
    if (count)
        for (size_t i = 0; i < n; ++i)
        {
            count[i] = ((const uint8_t *)(hs + o_cnt))[i];
        }
    if (ok)
        std::memcpy(ok, hs + o_ok, n);
    if (len)
        std::memcpy(len, hs + o_len, n * sizeof(float));
    return HYAS_OK;
}

int hyas_expand_batch(hyas_ctx *c, const hyas_node *preds, const hyas_goal *goals, const float *twod, size_t n, hyas_succ *out, uint8_t *count)
{
    if (c && n && (!preds || !goals || !out || !count))
        return c->fail(HYAS_EINVAL, "null buffer");
    if (c && c->dm.ad_mode == 1)
        return c->fail(HYAS_EINVAL, "adaptive_steering_angle_and_step_size needs the start pose of each search: use hyas_expand_batch_from");
    return run_step(c, preds, goals, nullptr, twod, n, 0, nullptr, nullptr, out, count);
}

int hyas_expand_batch_from(hyas_ctx *c, const hyas_node *preds, const hyas_goal *goals, const hyas_goal *starts, const float *twod, size_t n,
                           hyas_succ *out, uint8_t *count)
{
    if (c && n && (!preds || !goals || !starts || !out || !count))
        return c->fail(HYAS_EINVAL, "null buffer");
    return run_step(c, preds, goals, starts, twod, n, 0, nullptr, nullptr, out, count);
}

int hyas_shot_batch(hyas_ctx *c, const hyas_node *nodes, const hyas_goal *goals, size_t n, uint8_t *ok, float *len)
{
    if (c && n && (!nodes || !goals || !ok || !len))
        return c->fail(HYAS_EINVAL, "null buffer");
    return run_step(c, nodes, goals, nullptr, nullptr, n, 1, ok, len, nullptr, nullptr);
}

int hyas_shortcut_batch(hyas_ctx *c, const float *paths_xyt, const uint32_t *offsets, size_t n_paths, uint32_t *out_idx, uint32_t *out_n)
{
    if (!c || !c->have_map || c->grid_only)
        return HYAS_ESTATE;
    if (n_paths == 0)
        return HYAS_OK;
    if (!offsets || !out_idx || !out_n || (offsets[n_paths] && !paths_xyt))
        return c->fail(HYAS_EINVAL, "null buffer");
    for (size_t i = 0; i < n_paths; ++i)
        if (offsets[i] > offsets[i + 1])
            return c->fail(HYAS_EINVAL, "path offsets must not decrease");
    HYAS_HIP(c, hipSetDevice(c->device));
    std::vector<uint32_t> raw((size_t)offsets[n_paths] + n_paths);
    int rc = shortcut_paths(c, paths_xyt, offsets, n_paths, raw.data(), out_n);
    if (rc != HYAS_OK)
        return rc;
    // the kernel lists a path's nodes from its last one backwards (the order ShortCut pushes them in): reverse (:1177)
    for (size_t i = 0; i < n_paths; ++i)
        for (uint32_t j = 0; j < out_n[i]; ++j)
            out_idx[(size_t)offsets[i] + i + j] = raw[(size_t)offsets[i] + i + out_n[i] - 1 - j];
    return HYAS_OK;
}

int hyas_plan_batch(hyas_ctx *c, const float *starts, const float *goals, size_t n, hyas_result *out)
{
    if (!c || !c->have_map || !c->have_table)
        return HYAS_ESTATE;
    if (n && (!starts || !goals || !out))
        return c->fail(HYAS_EINVAL, "null buffer");
    return plan_batch_impl(c, starts, goals, n, out);
}

int hyas_get_path(const hyas_ctx *c, size_t i, float *xyt, size_t cap)
{
    if (!c || i >= c->paths.size())
        return HYAS_EINVAL;
    size_t n = c->paths[i].size() / 3;
    if (xyt)
        std::memcpy(xyt, c->paths[i].data(), std::min(cap, n) * 3 * sizeof(float));
    return (int)n;
}

int hyas_get_expanded(const hyas_ctx *c, size_t i, int32_t *idx, size_t cap)
{
    if (!c || i >= c->expanded.size())
        return HYAS_EINVAL;
    size_t n = c->expanded[i].size();
    if (idx)
        std::memcpy(idx, c->expanded[i].data(), std::min(cap, n) * sizeof(int32_t));
    return (int)n;
}

double hyas_last_kernel_ms(const hyas_ctx *c, int which, int64_t *launches)
{
    if (!c || which < 0 || which >= 5)
        return -1;
    if (launches)
        *launches = c->k_launches[which];
    return c->k_ms[which];
}

void hyas_reset_kernel_timers(hyas_ctx *c)
{
    if (c)
        reset_timers(c);
}

int hyas_set_stats(hyas_ctx *c, int enable)
{
    if (!c)
        return HYAS_EINVAL;
    c->stats_on = enable != 0;
    c->stats_mode = enable;
    HYAS_HIP(c, hipSetDevice(c->device));
    HYAS_HIP(c, hipMemset(c->d_stats, 0, 24 * sizeof(unsigned long long)));
    return HYAS_OK;
}

int hyas_set_round_timing(hyas_ctx *c, int every)
{
    if (!c || every < 0)
        return HYAS_EINVAL;
    c->round_timing = every;
    return HYAS_OK;
}

int hyas_nodes_in_device_memory(const hyas_ctx *c) { return c ? (c->d_nodes ? 1 : 0) : HYAS_EINVAL; }

int hyas_get_stats(hyas_ctx *c, uint64_t out[24])
{
    if (!c || !out)
        return HYAS_EINVAL;
    HYAS_HIP(c, hipSetDevice(c->device));
    HYAS_HIP(c, hipMemcpy(out, c->d_stats, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return HYAS_OK;
}

// obstacle polygons (8 floats each) and per-key list sizes, for tests
int hyas_get_obstacles(const hyas_ctx *c, float *out, size_t cap)
{
    if (!c || !c->have_map || c->grid_only)
        return HYAS_ESTATE;
    size_t n = std::min(cap, c->quads.size());
    for (size_t i = 0; i < n && out; ++i)
        for (int k = 0; k < 4; ++k)
            out[i * 8 + 2 * k] = c->quads[i].vx[k], out[i * 8 + 2 * k + 1] = c->quads[i].vy[k];
    return (int)c->quads.size();
}
int hyas_in_range_count(const hyas_ctx *c, uint32_t key)
{
    if (!c || !c->have_map || c->grid_only || key >= (uint64_t)c->w * c->h || (size_t)key + 1 >= c->cell_off.size())
        return -1;
    return (int)(c->cell_off[key + 1] - c->cell_off[key]);
}
int hyas_in_range_list(const hyas_ctx *c, uint32_t key, float *out, size_t cap)
{
    const int n = hyas_in_range_count(c, key);
    for (int i = 0; i < n && (size_t)i < cap && out; ++i)
    {
        const Quad &q = c->quads[c->cell_idx[c->cell_off[key] + i]];
        for (int k = 0; k < 4; ++k)
            out[i * 8 + 2 * k] = q.vx[k], out[i * 8 + 2 * k + 1] = q.vy[k];
    }
    return n;
}
}
