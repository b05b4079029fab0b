// hyas_ctx.h -- the context behind the C ABI: host copies of the map, device
// buffers, staging memory and kernel timers.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "hyas_hugepool.h"
#include "hyas_kernels.h"

#define HYAS_MAX_GROUPS 16

namespace hyas
{
struct SearchPools;
}

struct hyas_ctx
{
    hyas_params P{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t prep_stream = nullptr; // set-up of the next chunk of a large batch, beside the rounds of the current one
    // search rounds: up to HYAS_MAX_GROUPS query groups in flight, one stream each (round_stream[0] is `stream`)
    // and a begin/end event pair around the group's kernel
    hipStream_t round_stream[HYAS_MAX_GROUPS] = {};
    hipEvent_t round_ev[HYAS_MAX_GROUPS][2] = {};
    std::string err;
    std::mutex err_mutex; // a batch's set-up thread and its main thread may both report
    bool have_map = false, have_table = false, grid_only = false;

    // host-side map state
    uint32_t w = 0, h = 0;
    double res = 1, ox = 0, oy = 0;
    std::vector<uint8_t> charmap;
    std::vector<hyas::Quad> quads;
    std::vector<uint32_t> cell_off;
    std::vector<uint16_t> cell_idx;
    std::vector<int32_t> fp; // footprint table, 129 ints per entry
    std::vector<float> tab_xs, tab_ys, tab_ts;
    uint64_t table_size = 0;
    uint32_t seg_cap = 0;
    size_t twod_size = 0; // floats per 2D cost map

    // device mirrors
    hyas::DevMap dm{}; // pointers below, as the kernels see them
    uint8_t *d_charmap = nullptr;
    hyas::Quad *d_quads = nullptr;
    uint32_t *d_cell_off = nullptr;
    uint16_t *d_cell_idx = nullptr;
    int32_t *d_fp = nullptr;
    uint64_t *d_occ_tiles = nullptr;
    hyas::FpPack *d_fp_pack = nullptr;
    uint32_t *d_fp_win = nullptr;
    uint64_t *d_occ_pad = nullptr, *d_fp_m8 = nullptr;
    float *d_table = nullptr;
    float *d_tab_xs = nullptr, *d_tab_ys = nullptr, *d_tab_ts = nullptr;
    double *d_seg = nullptr;
    float *d_seg_db = nullptr;
    uint32_t *d_pose_off = nullptr, *d_seg_off = nullptr;
    uint16_t *d_pose_idx = nullptr, *d_seg_idx = nullptr;
    // adaptive step size / steering: per grid key the angle ranges (CSR), normalised density, minimum obstacle distance
    uint32_t *d_ad_off = nullptr;
    float *d_ad_rng = nullptr, *d_ad_density = nullptr;
    std::vector<uint32_t> ad_off;
    std::vector<float> ad_rng, ad_density, ad_min_distance;
    size_t n_pose_entries = 0, n_seg_entries = 0;

    // growable scratch: device + pinned host, reused by every batched call
    void *d_scratch = nullptr;
    size_t d_scratch_bytes = 0;
    void *h_scratch = nullptr;
    size_t h_scratch_bytes = 0;
    float *d_twod = nullptr;
    // successor filter of the search rounds: per query of a chunk and per cell the cost a successor has to beat (hyas::CellFilter)
    int32_t *d_bound = nullptr;
    size_t d_bound_bytes = 0;
    // popped nodes of the search rounds: device memory the host stores into directly (large BAR), null if unavailable
    void *d_nodes = nullptr;
    size_t d_nodes_bytes = 0;
    int nodes_tried = 0;
    size_t d_twod_bytes = 0;
    // workspace of the 2D flood kernel (cell coordinates, queue spill, cell state of maps too large for LDS)
    void *d_flood_ws = nullptr;
    size_t d_flood_ws_bytes = 0;
    // sampled curves of the queries an analytic shot closed (hyas_plan_batch)
    float *d_pts = nullptr;
    size_t d_pts_bytes = 0;
    void *d_sc = nullptr; // paths, offsets and results of the shortcut kernel
    size_t d_sc_bytes = 0;
    // host worker threads of the search driver, kept between calls (created on first use)
    hyas::SearchPools *pools = nullptr;

    // results of the last hyas_plan_batch
    std::vector<std::vector<float>> paths;
    std::vector<std::vector<int32_t>> expanded;

    // kernel timing, HIP events on `stream`
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double k_ms[5] = {0, 0, 0, 0, 0};
    int64_t k_launches[5] = {0, 0, 0, 0, 0};
    unsigned long long *d_stats = nullptr; // 8 work counters of the step kernel
    int round_timing = 0; // hyas_set_round_timing
    bool stats_on = false;
    int stats_mode = 0; // 1 work counters, 2 phase clocks

    int fail(int code, const std::string &msg)
    {
        std::lock_guard<std::mutex> lock(err_mutex);
        err = msg;
        return code;
    }
};

namespace hyas
{
// hipError -> HYAS_E*, records the message in the context
int hip_fail(hyas_ctx *c, hipError_t e, const char *what);
#define HYAS_HIP(c, call)                                    \
    do                                                       \
    {                                                        \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess)                                \
            return hyas::hip_fail((c), e_, #call);           \
    } while (0)

int ensure_scratch(hyas_ctx *c, size_t dev_bytes, size_t host_bytes);
int ensure_twod(hyas_ctx *c, size_t bytes);
int ensure_nodes(hyas_ctx *c, size_t bytes);
int ensure_buffer(hyas_ctx *c, void **buf, size_t *have, size_t bytes); // device buffer that only grows
void free_search_pools(hyas_ctx *c);
// HybridAStar::ShortCut for n paths on the GPU (host pointers; layout as launch_shortcut); synchronous on c->stream
int bezier_points(hyas_ctx *c, const float *anchors, size_t n_anchors, const uint32_t *first, const uint32_t *off, const uint32_t *n, size_t n_curves,
                  size_t total_points, float *out);
int shortcut_paths(hyas_ctx *c, const float *paths, const uint32_t *offs, size_t n, uint32_t *out_idx, uint32_t *out_n);
// 2D flood of `nq` queries on the GPU (see k_flood): poses6_dev / qlist_dev / twod_dev / start_cost_dev are device
// pointers; launches on `s` and returns without waiting
int flood_dev(hyas_ctx *c, const float *poses6_dev, const uint32_t *qlist_dev, uint32_t nq, float *twod_dev, float *start_cost_dev,
              hipStream_t s, int max_per_cu = 0);
void timer_begin(hyas_ctx *c);
void timer_end(hyas_ctx *c, int which); // call after the stream has been synchronised
void reset_timers(hyas_ctx *c);

int plan_batch_impl(hyas_ctx *c, const float *starts, const float *goals, size_t n, hyas_result *out);
} // namespace hyas
