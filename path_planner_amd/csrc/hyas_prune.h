// hyas_prune.h -- exact pruning of the per-key obstacle lists: the rules, and their host form (k_prune_lists is the
// device form the product runs; the host form checks it and serves the host-side tests).
#pragma once
#include <cmath>
#include <vector>

#include "hyas_dev.h"

namespace hyas
{
// Removes from the per-key obstacle lists every entry that cannot influence a
// verdict, WITHOUT changing any verdict:
//  * pose list (vehicle polygon, keyed by the cell of corner 0): the polygon
//    lies within one vehicle diagonal of corner 0, and IsIntersect rejects edge
//    pairs whose boxes are apart, so a quad whose box misses the cell box grown
//    by the diagonal can never intersect;
//  * segment list (keyed by the cell of a segment end, segments no longer than
//    lmax per axis): by seg_hits_quad()'s argument a quad farther than
//    lmax + half_wid + cull_margin from the cell box can only matter through
//    the ray-casting inside test, which needs an end point in the quad's y-range
//    and left of its xmax; such quads stay in the list flagged "ray only";
//    quads failing that too are dropped.
// A point maps to cell (col,row) when round((v-o)/res*1000)/1000 floors to it,
// i.e. it can sit up to 0.0005 cells outside; `slack` covers that and rounding.
// The growths and margins of the rules below, for a map and a longest segment extent lmax.
struct PruneMargins
{
    double slack, diag, reach;
};
inline PruneMargins prune_margins(const DevMap &m, float lmax)
{
    PruneMargins p;
    p.slack = 0.01 * m.res + 1e-3;
    p.diag = std::sqrt((double)m.veh_len * m.veh_len + (double)m.veh_wid * m.veh_wid) + p.slack;
    p.reach = (double)lmax + m.half_wid + (double)m.cull_margin + p.slack;
    return p;
}

inline void prune_lists(const DevMap &m, const Quad *quads, const uint32_t *cell_off, const uint16_t *cell_idx, float lmax,
                        std::vector<uint32_t> &pose_off, std::vector<uint16_t> &pose_idx, std::vector<uint32_t> &seg_off,
                        std::vector<uint16_t> &seg_idx)
{
    const size_t cells = (size_t)m.w * m.h;
    const PruneMargins pm = prune_margins(m, lmax);
    const double res = m.res, slack = pm.slack, diag = pm.diag, reach = pm.reach;
    pose_off.assign(cells + 1, 0), seg_off.assign(cells + 1, 0);
    pose_idx.clear(), seg_idx.clear();
    for (size_t key = 0; key < cells; ++key)
    {
        pose_off[key] = (uint32_t)pose_idx.size(), seg_off[key] = (uint32_t)seg_idx.size();
        const double x0 = m.ox + res * (double)(key % m.w), y0 = m.oy + res * (double)(key / m.w);
        const double x1 = x0 + res, y1 = y0 + res;
        for (uint32_t k = cell_off[key]; k < cell_off[key + 1]; ++k)
        {
            const uint16_t qi = cell_idx[k];
            const Quad &q = quads[qi];
            auto overlaps = [&](double grow) {
                return !(q.xmax < x0 - grow || q.xmin > x1 + grow || q.ymax < y0 - grow || q.ymin > y1 + grow);
            };
            if (overlaps(diag))
                pose_idx.push_back(qi);
            if (overlaps(reach))
                seg_idx.push_back(qi);
            else if (!(q.ymax < y0 - lmax - slack || q.ymin > y1 + lmax + slack) && q.xmax >= x0 - lmax - slack)
                seg_idx.push_back((uint16_t)(qi | 0x8000));
        }
    }
    pose_off[cells] = (uint32_t)pose_idx.size(), seg_off[cells] = (uint32_t)seg_idx.size();
}

} // namespace hyas
