/*
 * hyas.h -- C ABI of the MI355X Hybrid A* expansion path.
 *
 * Every entry point replaces one piece of the reference planner
 * (JialiangHan/path_planner); the file:line after "replaces" is the reference
 * interface it stands in for.  Conventions: extern "C", plain pointers and
 * sizes, caller-allocated outputs, return 0 on success or a negative
 * HYAS_E* code, no exceptions cross the boundary, one context per GPU, a
 * context is not thread-safe, every call is synchronous on the context's
 * stream.  Pointers are HOST pointers unless the name ends in _dev.
 */
#ifndef HYAS_H
#define HYAS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HYAS_OK 0
#define HYAS_EINVAL (-22)   /* bad argument / unsupported parameter */
#define HYAS_ENOMEM (-12)   /* device or host allocation failed */
#define HYAS_ENODEV (-19)   /* no usable GPU / HIP error */
#define HYAS_ESTATE (-1)    /* call order violated (e.g. no costmap yet) */
#define HYAS_ERANGE (-34)   /* size beyond a documented maximum */

#define HYAS_MODE_POLYGON 0   /* live collision test of the reference */
#define HYAS_MODE_FOOTPRINT 1 /* per-heading footprint cell lists */

#define HYAS_CURVE_DUBINS 0
#define HYAS_CURVE_RS 1
#define HYAS_CURVE_BEZIER 2 /* cubic Bezier heuristic + shot (src/cubic_bezier.cpp, include/bezier.h) */

#define HYAS_MAX_SUCC 8 /* most successors one node can generate (fixed steering set) */
#define HYAS_ADAPTIVE_SLOTS 32 /* successor slots per node with adaptive_steering_angle_and_step_size: up to 31 forward
                                * {step, steering} pairs and the reverse step */

/* Planner parameters.  Field names, meaning and defaults follow
 * ParameterCollisionDetection and ParameterHybridAStar
 * (include/parameter_manager.h:10-42, :124-175).  hyas_default_params()
 * fills the struct defaults of the reference; hyas_hot_path_params() the
 * fixed-step Reeds-Shepp configuration the benchmarks are quoted on. */
typedef struct hyas_params
{
    /* ParameterCollisionDetection */
    int32_t position_resolution;      /* 10 */
    int32_t headings;                 /* 72 */
    float cd_min_turning_radius;      /* 6: radius of the heuristic table */
    int32_t curve_type;               /* 0 Dubins, 1 Reeds-Shepp, 2 cubic Bezier */
    float vehicle_width;              /* 1 */
    float vehicle_length;             /* 2 */
    float obstacle_detection_range;   /* 5 */
    int32_t map_boundary_obstacle;    /* 0; 1: four rectangles around the grid join the obstacles (src/collisiondetection.cpp:305-308, :1223-1240) */
    /* ParameterHybridAStar */
    int32_t number_of_successors;     /* 3 */
    int32_t adaptive_steering_angle_and_step_size; /* 0; wins over adaptive_step_size when both are set */
    int32_t adaptive_step_size;       /* 1 */
    int32_t analytical_expansion_every_point;      /* 1 */
    int32_t analytical_expansion;     /* 1 */
    float penalty_turning;            /* 1.05 */
    float penalty_reverse;            /* 1 */
    float penalty_change_of_direction;/* 2 */
    int32_t reverse;                  /* 0 */
    int32_t max_iterations;           /* 10000 */
    float steering_angle;             /* 5 [deg] */
    float goal_range;                 /* 1e-3 */
    float tie_breaker;                /* 0.01 */
    float min_turning_radius;         /* 6: radius of the analytic shot */
    /* footprint-lookup mode, Constants:: (include/constants.h:35-58) */
    double fp_width;                  /* 0.18 */
    double fp_length;                 /* 0.22 */
    float fp_cell_size;               /* 1 */
    /* path post-processing and the remaining ParameterHybridAStar / ParameterCollisionDetection switches
     * (include/parameter_manager.h:28-40, :128, :162, :170-172) */
    int32_t short_cut;                /* 1: HybridAStar::ShortCut on the traced path */
    float max_steering_angle;         /* 30 */
    int32_t consider_steering_angle_range_for_obstacle_density;                  /* 0 */
    int32_t add_one_more_successor_only_in_free_angle_range;                     /* 0 */
    int32_t add_one_more_successor;                                              /* 1 */
    int32_t steering_angle_towards_free_angle_range_for_obstacle_angle_range;    /* 1 */
    int32_t fixed_number_of_steering_angle_in_free_angle_range;                  /* 1 */
    float constant_density;           /* 0.5 */
    int32_t piecewise_cubic_bezier_interpolation; /* 0; 1: the traced part of a path closed by an analytic shot is replaced by
                                                   * cubic Bezier curves between its anchor nodes (src/hybrid_a_star.cpp:830-883) */
    float curve_step_size;            /* 1 */
    int32_t reserved;
} hyas_params;

/* A search node as the expansion kernels see it: the fields of Node3D
 * (include/node3d.h:100-124) that CreateSuccessor/UpdateCostSoFar read. */
typedef struct hyas_node
{
    float x, y, t; /* pose, t in [0, 2pi) */
    float g;       /* cost so far */
    int32_t prim;  /* motion primitive 0..5 */
} hyas_node;

typedef struct hyas_goal
{
    float x, y, t;
} hyas_goal;

/* One traversable successor, in the order the reference emits them. */
typedef struct hyas_succ
{
    float x, y, t;
    float g;       /* UpdateCostSoFar result */
    float h;       /* UpdateHeuristic result */
    int32_t idx;   /* Node3D::setIdx */
    int8_t prim;
    uint8_t free;  /* always 1 for an emitted successor */
    uint16_t pad;
} hyas_succ;

#define HYAS_STATUS_NO_PATH 0   /* open list ran empty */
#define HYAS_STATUS_GOAL 1      /* goal test passed on a popped node */
#define HYAS_STATUS_SHOT 2      /* analytic expansion connected */
#define HYAS_STATUS_MAX_ITER 3  /* max_iterations reached: partial path */
#define HYAS_STATUS_BAD_POSE 4  /* start or goal rejected (makePlan pre-check) */
#define HYAS_STATUS_OVERFLOW 5  /* adaptive steering: a node produced more than HYAS_ADAPTIVE_SLOTS - 1 distinct {step, steering}
                                 * pairs (pairs differ by a degree of steering at least; the in-tree maps stay at or below 20) */

typedef struct hyas_result
{
    int32_t status;
    int32_t nodes_explored; /* pops incl. lazily deleted ones */
    int32_t nodes_expanded; /* nodes closed and expanded */
    int32_t successors;     /* successors that received a verdict + g + h */
    int32_t path_len;       /* poses in the final path */
    float g_final;          /* g of the node the path was closed from */
    float shot_length;      /* analytic-curve length, 0 if none */
    float path_length;      /* sum of pose-to-pose distances */
} hyas_result;

typedef struct hyas_ctx hyas_ctx;

void hyas_default_params(hyas_params *p);
void hyas_hot_path_params(hyas_params *p);

/* replaces: HybridAStarPlanner ctor + parameter load,
 * src/hybrid_a_star_planner.cpp:17-35, :59.  device = HIP device ordinal.
 * Side effect on the calling process: hyas_create calls
 * setenv("HIP_FORCE_DEV_KERNARG", "1", overwrite = 0) before its first HIP call
 * (kernel arguments in device memory: 2-3 us less per launch).  The HIP runtime
 * reads the variable when it starts up, so the setting only takes effect in a
 * process that has made no HIP call yet; a value the caller has set is kept. */
hyas_ctx *hyas_create(const hyas_params *params, int device);
void hyas_destroy(hyas_ctx *ctx);
const char *hyas_last_error(const hyas_ctx *ctx);

/* replaces: HybridAStarPlanner::initialize(name, Costmap2D*, frame)
 * src/hybrid_a_star_planner.cpp:54-87 = CollisionDetection::UpdateGrid
 * (src/collisiondetection.cpp:26-54) + LookupTable::Initialize
 * (src/lookup_table.cpp:20-46).  charmap is costmap_2d's row-major cost
 * array (0 free .. 254 lethal, 255 unknown).  build_table != 0 also builds
 * the heuristic table for params.curve_type on this GPU; pass 0 when the
 * table is installed afterwards with hyas_table_install_dev. */
int hyas_set_costmap(hyas_ctx *ctx, const uint8_t *charmap, uint32_t w, uint32_t h,
                     double resolution, double origin_x, double origin_y, int build_table);

/* Footprint-list mode only: a bare occupancy grid of any size with unit
 * cells and origin 0 (what the table of include/lookup.h assumes); no obstacle
 * polygons and no heuristic table are built.  No counterpart call in the
 * reference: its footprint table (include/collisiondetection.h:306) is never
 * attached to a grid. */
int hyas_set_grid(hyas_ctx *ctx, const uint8_t *charmap, uint32_t w, uint32_t h);

/* Obstacle polygons of CollisionDetection::SetObstacleVec
 * (src/collisiondetection.cpp:262-317), 8 floats (4 vertices) each; returns the
 * polygon count.  hyas_in_range_count: size of in_range_obstacle_map_[key]
 * (:319-354), 0 for an absent key, -1 for a key outside the grid. */
int hyas_get_obstacles(const hyas_ctx *ctx, float *out, size_t cap_polygons);
int hyas_in_range_count(const hyas_ctx *ctx, uint32_t key);
/* The list itself: in_range_obstacle_map_[key] (src/collisiondetection.cpp:335-345) in list order, 8 floats per polygon;
 * writes at most cap_polygons polygons, returns the list length (-1 for a key outside the grid). */
int hyas_in_range_list(const hyas_ctx *ctx, uint32_t key, float *out, size_t cap_polygons);

/* replaces: CollisionDetection::IsTraversable(const Node3D&)
 * src/collisiondetection.cpp:93-127 (mode POLYGON) or the footprint-list
 * configurationTest the table of include/lookup.h:30-284 was built for
 * (mode FOOTPRINT).  xyt = n x {x,y,t}; pred_xy = n x {x,y} of the
 * predecessor or NULL (pose test only); free_out[i] = 1 iff traversable. */
int hyas_collision_batch(hyas_ctx *ctx, int mode, const float *xyt, const float *pred_xy,
                         size_t n, uint8_t *free_out);
/* The same on DEVICE buffers.  Any alignment a float / byte array may have is
 * accepted; mode FOOTPRINT runs its vectorised kernels when xyt_dev is 16-byte
 * and free_out_dev 4-byte aligned (what hipMalloc returns) and a
 * scalar-access kernel otherwise. */
int hyas_collision_batch_dev(hyas_ctx *ctx, int mode, const float *xyt_dev,
                             const float *pred_xy_dev, size_t n, uint8_t *free_out_dev);

/* Footprint table of mode FOOTPRINT as the reference lays it out
 * (Constants::config, include/constants.h:70-80): entries =
 * position_resolution^2 * headings, each {int32 length; int32 pos[64][2]}.
 * Copies min(cap, entries) entries; returns the entry count. */
int hyas_footprint_table(hyas_ctx *ctx, int32_t *out, size_t cap_entries);

/* replaces: HybridAStar::CreateSuccessor + UpdateCostSoFar + UpdateHeuristic
 * src/hybrid_a_star.cpp:540-756, :761-811, :316-366 for n independent
 * (node, goal) pairs.  UpdateCostSoFar compares each successor's primitive
 * with preds[i].prim.  twod_cost = n x hyas_twod_size() floats, the 2D cost
 * map of each goal (hyas_flood), or NULL for all-zero.  out = n x
 * hyas_num_successors() slots, compacted per node; count[i] = emitted. */
int hyas_expand_batch(hyas_ctx *ctx, const hyas_node *preds, const hyas_goal *goals,
                      const float *twod_cost, size_t n, hyas_succ *out, uint8_t *count);
/* The same with the start pose of the search each pair belongs to: with
 * adaptive_steering_angle_and_step_size the steering towards the goal is
 * weighted by the start-goal distance (CollisionDetection::FindStepSizeAndSteeringAngle,
 * src/collisiondetection.cpp:1383-1401).  hyas_expand_batch refuses that mode.
 * count[i] = 0xFF: node i produced more than HYAS_ADAPTIVE_SLOTS - 1 distinct pairs. */
int hyas_expand_batch_from(hyas_ctx *ctx, const hyas_node *preds, const hyas_goal *goals, const hyas_goal *starts,
                           const float *twod_cost, size_t n, hyas_succ *out, uint8_t *count);
/* replaces: the per-costmap maps of the adaptive modes, CollisionDetection::SetMinDistanceMap,
 * SetDistanceAngleRangeMap, BuildObstacleDensityMap, BuildNormalizedObstacleDensityMap
 * (src/collisiondetection.cpp:387-500, :1013-1151, :1403-1447), built inside hyas_set_costmap; this call reads
 * one grid key back: minimum obstacle distance, normalised density and up to cap {distance, start, width} angle
 * ranges as FindFreeAngleRangeAndObstacleAngleRange (:688-725) returns them.  Returns the number of ranges. */
int hyas_adaptive_cell(hyas_ctx *ctx, uint32_t key, float *min_distance, float *density, float *ranges3, size_t cap);
int hyas_num_successors(const hyas_ctx *ctx);

/* replaces: HybridAStar::AnalyticExpansions src/hybrid_a_star.cpp:371-534.
 * ok[i] = 1 iff every sample of the curve from nodes[i] to goals[i] is
 * traversable; len[i] = curve length. */
int hyas_shot_batch(hyas_ctx *ctx, const hyas_node *nodes, const hyas_goal *goals, size_t n,
                    uint8_t *ok, float *len);

/* replaces: HybridAStar::BuildAStarCostMap src/hybrid_a_star.cpp:1020-1074.
 * out = hyas_twod_size() floats (w*h + w + 1: Node2D::setIdx can land one row
 * past the grid), indexed by Node2D::setIdx, -1 where the flood never arrived. */
int hyas_twod_size(const hyas_ctx *ctx);
int hyas_flood(hyas_ctx *ctx, const hyas_goal *goal, float *out);

/* replaces: LookupTable::CalculateReedsSheppLookup / CalculateDubinsLookup
 * src/lookup_table.cpp:147-224, restricted to heading-loop iterations
 * [slice_begin, slice_end).  out_dev (DEVICE memory) receives
 * (slice_end-slice_begin) x ny x nx costs in loop order; nx, ny and the
 * slice count come from hyas_table_dims. */
int hyas_table_dims(const hyas_ctx *ctx, int32_t *nx, int32_t *ny, int32_t *nslices);
int hyas_table_build_dev(hyas_ctx *ctx, int curve, int slice_begin, int slice_end,
                         float *out_dev);
/* Applies the reference's first-write-wins emplace (src/lookup_table.cpp:176,
 * :215) of all nslices raw slices into the context's keyed table. */
int hyas_table_install_dev(hyas_ctx *ctx, int curve, const float *raw_dev);
/* hyas_table_install_dev for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc
 * once they are running): the same first-write-wins install, with its per-key scratch (one loop rank per
 * table key) in a static device array instead of a temporary allocation.  No counterpart in the reference.  Calls no
 * hipMalloc / hipFree.  Like every call here it launches on the context's own stream and waits for it, so it is not
 * for use inside a HIP graph capture.  Limit: tables of at most HYAS_NOALLOC_MAX_KEYS keys (hyas_table_dims: (headings + 1) * w * h *
 * position_resolution^2 + w * position_resolution), HYAS_ERANGE beyond.  All contexts on one GPU share that GPU's
 * array: calls that overlap on one GPU race; consecutive calls are safe (every call resets the ranks it uses). */
#define HYAS_NOALLOC_MAX_KEYS (1u << 20)
int hyas_table_install_noalloc_dev(hyas_ctx *ctx, int curve, const float *raw_dev);
/* replaces: LookupTable::Get{ReedsShepp,Dubins}Cost src/lookup_table.cpp:56-96
 * for n relative poses {|dx|,|dy|,dt}. */
int hyas_table_lookup(hyas_ctx *ctx, const float *xyt, size_t n, float *cost);

/* replaces: Expander::calculatePath / HybridAStar::calculatePath
 * src/hybrid_a_star.cpp:982-1018 (GetPath :41-311) for n independent
 * queries, preceded by the makePlan pose checks
 * (src/hybrid_a_star_planner.cpp:126-155).  Paths and expansion orders of the
 * last batch stay in the context until the next call. */
int hyas_plan_batch(hyas_ctx *ctx, const float *starts_xyt, const float *goals_xyt, size_t n,
                    hyas_result *out);
/* replaces: HybridAStar::ShortCut(path, true) src/hybrid_a_star.cpp:1076-1179 (with
 * CollisionDetection::IsTraversable(current, previous), src/collisiondetection.cpp:129-145) for n_paths
 * independent paths.  paths_xyt: the poses {x,y,t} of all paths one after the other; path i = poses
 * [offsets[i], offsets[i+1]).  out_idx + offsets[i] + i receives the indices (into path i) of the nodes ShortCut
 * keeps, in path order, out_n[i] <= len_i + 1 of them.  A path of fewer than two nodes comes back unchanged (the
 * reference reads input[size-2] there).  hyas_plan_batch applies this itself when params.short_cut is set. */
int hyas_shortcut_batch(hyas_ctx *ctx, const float *paths_xyt, const uint32_t *offsets, size_t n_paths,
                        uint32_t *out_idx, uint32_t *out_n);
/* Copies min(cap, len) poses {x,y,t} / expanded idx of query i; returns len. */
int hyas_get_path(const hyas_ctx *ctx, size_t i, float *xyt, size_t cap);
int hyas_get_expanded(const hyas_ctx *ctx, size_t i, int32_t *idx, size_t cap);

/* Time in milliseconds the GPU spent in each kernel family during the last
 * hyas_plan_batch / *_batch call, measured with HIP events on the streams the
 * kernels were launched on, summed over launches; *launches receives their
 * number.  which: 0 search round (hyas_plan_batch) or shot (hyas_shot_batch),
 * 1 successors (hyas_expand_batch), 2 collision batch, 3 table build, 4 unused.
 * The search rounds of hyas_plan_batch carry timing events only after
 * hyas_set_round_timing(ctx, k) with k > 0 (every 2k-th launch of a query group
 * is then timed and the sum scaled up; k = 0, the default, records none: two
 * event records lengthen a launch call by 2-3 us); the launch count is kept
 * either way.  No counterpart in the reference. */
double hyas_last_kernel_ms(const hyas_ctx *ctx, int which, int64_t *launches);
int hyas_set_round_timing(hyas_ctx *ctx, int every);
void hyas_reset_kernel_timers(hyas_ctx *ctx);

/* Work counters of the expansion-step kernel, for roofline accounting (no
 * counterpart in the reference).  hyas_set_stats(ctx, mode) zeroes them; mode 1
 * makes hyas_plan_batch run the counting (slower) kernel, mode 2 the kernel
 * with phase clocks only, 0 the plain one; hyas_get_stats reads:
 * 0 curve samples tested, 1 samples a serial scan would test (up to and
 * including the first colliding one), 2 obstacle-list entries behind (1),
 * 3 successors that received a verdict, 4 obstacle-list entries behind (3),
 * 5 successors emitted, 6 nodes expanded (mode 2: shot workgroups), 7 shots
 * evaluated; mode 2, shader clock sums over the shot workgroups: 8 curve
 * solve, 9 sample interpolation, 10 sample collision tests, 11..14 time from
 * the start of a workgroup to: goal frame normalised, candidates evaluated,
 * winner known, prefix states stored; 15 arrival of node and goal, 16 whole
 * workgroup, 17 / 18 pose / swept-segment tests of one wave each, 19 sample
 * passes.  The same modes apply to hyas_shot_batch and hyas_expand_batch. */
int hyas_set_stats(hyas_ctx *ctx, int enable);
/* 1 if the last hyas_plan_batch handed its popped nodes to the GPU through device memory the host stores into
 * directly (large-BAR GPUs), 0 if through pinned host memory (HYAS_NODES_IN_HOST_MEMORY=1 or no such mapping);
 * no counterpart in the reference. */
int hyas_nodes_in_device_memory(const hyas_ctx *ctx);
int hyas_get_stats(hyas_ctx *ctx, uint64_t out[24]);

#ifdef __cplusplus
}
#endif
#endif /* HYAS_H */
