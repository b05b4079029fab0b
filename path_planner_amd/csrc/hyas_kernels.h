// hyas_kernels.h -- launch wrappers of hyas_kernels.hip (device pointers only).
#pragma once
#include <hip/hip_runtime.h>

#include "hyas_dev.h"

namespace hyas
{

// One popped node handed to the expansion step.
struct StepIn
{
    float x, y, t, g;
    int32_t prim;
    uint32_t query; // which goal / 2D cost map
    uint32_t shoot; // bit 0: evaluate the analytic shot for this node; STEP_DEAD: no node in this slot
    uint32_t pad;
};
#define STEP_DEAD 2u
// flags of the successor role inside a search round
#define ROUND_CHILD_SHOOT 1u   // the records written for a deeper launch ask for the analytic shot
#define ROUND_TAG_CANDIDATE 2u // hyas_succ::pad, high byte: 1 + index of the motion-set entry (fixed motion set) the successor comes from

// Successors the search would drop anyway (src/hybrid_a_star.cpp:264, :271: cell closed, or cell open with a cost that the
// new one does not beat) are dropped by the round kernel before a record is written.  bound[query * cells + idx] holds,
// as the bits of a float, an UPPER bound of the cost a successor in that cell has to beat: +inf while the cell is
// unseen, the cell's cost while it is open, -inf once it is closed.  The true value only ever falls, every value the
// host sends was true at some time, and the kernel keeps the minimum of what it was sent, so the bound never lies below
// the true value and a test against it drops no successor the host would keep; what it lets through the host tests again.
struct CellUpd
{
    uint32_t query; // row of `bound`
    int32_t idx;    // Node3D::setIdx of the cell
    int32_t bits;   // new value: bits of the pushed node's cost, or of -inf when the cell was closed
    uint32_t pad;
};
struct CellFilter
{
    int32_t *bound;      // null: no filter
    uint32_t cells;      // entries per query
    const CellUpd *upd;  // changes since the last launch, applied by extra workgroups of this one
    uint32_t n_upd;
};

// Results of a search round as the host reads them while the kernel is still running: one block of `units` 16-byte units
// per node of the launch, every unit carrying the round's tag in its last word, so that the host can tell, unit by unit
// and without any completion signal, whether what it reads is this round's (a unit is stored by one lane in one 16-byte
// store).  Unit 0, the analytic shot: {length bits, verdict, 0, tag}; unit 1, the successors: {count byte as in the
// count array, 0, 0, tag}; units 2 + 2j and 3 + 2j, successor j: {x, y, t, tag << 9 | child mark << 8 | (motion-set
// entry + 1) << 3 | primitive} and {g, h, idx, tag}.  Tags are 23 bits and never 0.
struct RoundOut
{
    uint4 *base;    // null: results go to the plain arrays
    uint32_t units; // per node: 2 + 2 * successor slots
    uint32_t tag;
};
#define ROUND_TAG_MASK 0x7FFFFFu

void launch_collision(const DevMap &m, int mode, const float *xyt, const float *pred_xy, size_t n, uint8_t *out, hipStream_t s);
// map ingest: occupancy values, occupancy bit tiles (padded may be null), per-key in-range lists.
void launch_occupancy(const uint8_t *cost, size_t n, int8_t *occ, hipStream_t s);
void launch_occ_tiles(const uint8_t *cost, uint32_t w, uint32_t h, uint32_t tx, uint32_t ty, uint64_t *tiles, uint64_t *padded, hipStream_t s);
// In-range lists in two steps of two passes each, so that nothing is tested against every polygon: bins of polygons
// per tile of HYAS_BIN_TILE x HYAS_BIN_TILE grid keys, then each key's list from the bin of its tile.
#define HYAS_BIN_TILE 16u
struct RangeBins
{
    uint32_t tiles_x, tiles_y;
    float extra;   // added to the growth of the boxes: 0, or infinity to put every polygon into every bin
    uint32_t *off; // tiles_x * tiles_y + 1 entries
    uint16_t *idx; // polygon ids, tile after tile, ascending inside a tile
};
// pass 0: owner[key] (w*h entries), bins.off = offsets of the bins, *total = their sum; pass 1: bins.idx filled.
void launch_range_bins(const DevMap &m, const Quad *quads, uint32_t n_quads, const float *xs, uint32_t nxs, const float *ys, uint32_t nys, float range,
                       uint32_t *owner, RangeBins bins, unsigned long long *total, int pass, hipStream_t s);
// pass 0: off[0..w*h] = offsets of the lists, *total = their sum; pass 1: idx filled (total entries).
void launch_in_range_lists(const DevMap &m, const Quad *quads, const float *xs, const float *ys, uint32_t nys, float range, const uint32_t *owner,
                           RangeBins bins, uint32_t *off, uint16_t *idx, unsigned long long *total, int pass, hipStream_t s);
// Exact pruning of the per-key lists on the GPU (hyas_prune.h states the rules): pass 0 leaves the offsets in
// pose_off / seg_off (cells + 1 entries each) and the two totals in totals2[0..1]; pass 1 fills pose_idx / seg_idx,
// which hold at most as many entries as the full lists.
struct PruneArgs
{
    double res, ox, oy;
    double diag, reach; // growth of the cell box for the pose list and for the segment list
    double lmax, slack; // the ray-only rule of the segment list
    uint32_t w, cells;
};
void launch_prune_lists(const Quad *quads, const uint32_t *cell_off, const uint16_t *cell_idx, const PruneArgs &a, uint32_t *pose_off, uint16_t *pose_idx,
                        uint32_t *seg_off, uint16_t *seg_idx, unsigned long long *totals2, int pass, hipStream_t s);
// Adaptive maps of every grid key (w*h keys; owner[key] = 0xFFFFFFFF: no corner maps to it): pass 0 leaves
// min_distance[key], the offsets of the keys' range lists in off[0..w*h] and their total in *total; pass 1 writes the
// ranges {label, start, width} into rng3.  n_ang <= HYAS_SWEEP_MAX.  kept (optional, w*h*keep*3 floats, keep <= 21): where
// pass 0 leaves every key's first `keep` ranges, so that pass 1 only sweeps again for the keys that have more.
#define HYAS_SWEEP_MAX 368u
void launch_adaptive_ranges(const DevMap &m, const float *xs, const float *ys, uint32_t nys, const uint32_t *owner, const float *angles, uint32_t n_ang,
                            float radius, float *min_distance, uint32_t *off, float *rng3, float *kept, uint32_t keep, unsigned long long *total, int pass,
                            hipStream_t s);
void launch_table_rs(const float *xs, uint32_t nx, const float *ys, uint32_t ny, const float *ts, uint32_t k0, uint32_t k1, double rho,
                     float *out, hipStream_t s);
void launch_table_dubins(const float *xs, uint32_t nx, const float *ys, uint32_t ny, const float *ts, uint32_t k0, uint32_t k1, float rho,
                         float *out, hipStream_t s);
void launch_table_bezier(const float *xs, uint32_t nx, const float *ys, uint32_t ny, const float *ts, uint32_t k0, uint32_t k1, float *out,
                         hipStream_t s);
void launch_table_install(const DevMap &m, const float *raw, const float *xs, uint32_t nx, const float *ys, uint32_t ny, const float *ts,
                          uint32_t ns, uint32_t *rank, float *table, uint64_t table_size, hipStream_t s);
void launch_table_lookup(const DevMap &m, const float *xyt, size_t n, float *cost, hipStream_t s);
void launch_expand(const DevMap &m, const StepIn *in, uint32_t n, const hyas_goal *goals, const float *twod, uint8_t *count,
                   hyas_succ *succ, unsigned long long *stats, int stats_mode, hipStream_t s);
void launch_shot(const DevMap &m, const StepIn *in, uint32_t n, const hyas_goal *goals, const double *seg_tab, uint32_t seg_cap,
                    uint8_t *shot_ok, float *shot_len, unsigned long long *stats, int stats_mode, hipStream_t s);
hipError_t resolve_round_kernel(const DevMap &m, bool stats, int stats_mode);
void launch_round(const DevMap &m, const StepIn *in, uint32_t n, const hyas_goal *goals, const float *twod, const double *seg_tab,
                  uint32_t seg_cap, uint8_t *shot_ok, float *shot_len, uint8_t *count, hyas_succ *succ, unsigned long long *stats,
                  int stats_mode, hipStream_t s, StepIn *child = nullptr, bool child_shoot = false, CellFilter filter = CellFilter{nullptr, 0, nullptr, 0},
                  RoundOut out = RoundOut{nullptr, 0, 0});
// child (optional, n records in device memory): the launch also writes, for every node, the input record of its
// lowest-f successor (shoot = child_shoot; STEP_DEAD when the node has none), ready to be the `in` of a further launch
// on the same stream, and marks that successor's record with bit 0 of pad.  A search round tags every successor
// record with its motion-set entry (ROUND_TAG_CANDIDATE); hyas_expand_batch leaves pad = 0.
// paths: concatenated poses {x, y, t}; offs: n_paths + 1 pose offsets; out_idx: offs[n_paths] + n_paths entries; out_n: n_paths
void launch_bezier_points(const float *anchors, const uint32_t *first, const uint32_t *off, const uint32_t *n, uint32_t n_curves, float *out, hipStream_t s);
void launch_shortcut(const DevMap &m, const float *paths, const uint32_t *offs, uint32_t n_paths, uint32_t *out_idx, uint32_t *out_n, hipStream_t s);
void launch_shot_path(const DevMap &m, const StepIn *nodes, uint32_t n, const hyas_goal *goals, const double *seg_tab, uint32_t seg_cap,
                      const uint32_t *offsets, float *out, uint32_t *n_out, hipStream_t s);


// Launch geometry and workspace sizes of the 2D flood kernel for a map of n_cells (= twod_size) cells.
struct FloodPlan
{
    uint32_t n_cells, cells_pad, heap_cap, grid;
    bool narrow, cells_in_lds;
    size_t lds_bytes, ws_xy_bytes, ws_heap_bytes, ws_g_bytes, ws_ps_bytes;
};
// max_per_cu > 0 caps the workgroups per compute unit (1: the LDS request is padded so that no second one fits)
FloodPlan plan_flood(uint32_t n_cells, uint32_t n_queries, int heap_cap_override, int max_per_cu = 0);
// poses6: n x {start x, y, t, goal x, y, t}; qlist: which queries to flood (null: 0..nq-1); twod: [query][n_cells];
// start_cost (optional): the cost at each query's start cell, -1 if the flood never closed it.
hipError_t launch_flood(const DevMap &m, const FloodPlan &fp, const float *poses6, const uint32_t *qlist, uint32_t nq, float *twod,
                        float *start_cost, void *ws_xy, void *ws_heap, void *ws_g, void *ws_ps, hipStream_t s);

} // namespace hyas
