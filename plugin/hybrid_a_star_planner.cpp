// See hybrid_a_star_planner.h.  Reference behaviour followed:
// src/hybrid_a_star_planner.cpp:9 (plugin export), :27-87 (constructors,
// initialize), :89-97 (make_plan service), :99-124 (makePlan), :157-183
// (publishPlan); pose conversion src/utility.cpp:7-35.
#include "hybrid_a_star_planner.h"

#include <cstdio>

#ifdef HYAS_HAVE_ROS
#include <pluginlib/class_list_macros.h>
PLUGINLIB_EXPORT_CLASS(HybridAStar::HybridAStarPlanner, nav_core::BaseGlobalPlanner)
#define HYAS_LOG_ERROR(...) ROS_ERROR(__VA_ARGS__)
#define HYAS_LOG_WARN(...) ROS_WARN(__VA_ARGS__)
#else
#define HYAS_LOG_ERROR(...) (std::fprintf(stderr, "[HybridAStarPlanner] " __VA_ARGS__), std::fputc('\n', stderr))
#define HYAS_LOG_WARN(...) (std::fprintf(stderr, "[HybridAStarPlanner] " __VA_ARGS__), std::fputc('\n', stderr))
#endif

namespace HybridAStar
{

// The struct defaults of the reference (include/parameter_manager.h:10-42, :124-175) until setParams or the ROS
// parameter server says otherwise, as ParameterManager starts from them.
HybridAStarPlanner::HybridAStarPlanner() { hyas_default_params(&params_); }

HybridAStarPlanner::HybridAStarPlanner(std::string name, costmap_2d::Costmap2DROS *costmap_ros)
{
    hyas_default_params(&params_);
    initialize(name, costmap_ros);
}

#ifdef HYAS_HAVE_ROS
// ParameterManager::LoadHybridAStarParams + LoadCollisionDetectionParams (src/parameter_manager.cpp:117-165, :187-252):
// every parameter is looked up under the fixed prefix /move_base/HybridAStar/ and keeps its value when absent.
namespace
{
template <typename T>
void get_param(const std::string &name, T &value)
{
    ros::param::get("/move_base/HybridAStar/" + name, value);
}
void get_flag(const std::string &name, int32_t &value)
{
    bool b = value != 0;
    ros::param::get("/move_base/HybridAStar/" + name, b);
    value = b ? 1 : 0;
}
void get_float(const std::string &name, float &value)
{
    double d = value;
    ros::param::get("/move_base/HybridAStar/" + name, d);
    value = (float)d;
}
} // namespace

void HybridAStarPlanner::loadRosParams()
{
    hyas_params &p = params_;
    // ParameterCollisionDetection
    get_float("obstacle_detection_range", p.obstacle_detection_range);
    get_flag("map_boundary_obstacle", p.map_boundary_obstacle);
    get_flag("consider_steering_angle_range_for_obstacle_density", p.consider_steering_angle_range_for_obstacle_density);
    get_flag("add_one_more_successor_only_in_free_angle_range", p.add_one_more_successor_only_in_free_angle_range);
    get_flag("steering_angle_towards_free_angle_range_for_obstacle_angle_range", p.steering_angle_towards_free_angle_range_for_obstacle_angle_range);
    get_flag("fixed_number_of_steering_angle_in_free_angle_range", p.fixed_number_of_steering_angle_in_free_angle_range);
    get_flag("add_one_more_successor", p.add_one_more_successor);
    get_param("position_resolution", p.position_resolution);
    get_param("headings", p.headings);
    get_param("curve_type", p.curve_type);
    get_float("vehicle_width", p.vehicle_width);
    get_float("vehicle_length", p.vehicle_length);
    get_float("max_steering_angle", p.max_steering_angle);
    // "min_turning_radius" and "reverse" feed both structs (:166-170, :218-219, :233-234)
    get_float("min_turning_radius", p.cd_min_turning_radius);
    get_float("min_turning_radius", p.min_turning_radius);
    get_flag("reverse", p.reverse);
    // ParameterHybridAStar
    get_param("number_of_successors", p.number_of_successors);
    get_flag("adaptive_step_size", p.adaptive_step_size);
    get_flag("piecewise_cubic_bezier_interpolation", p.piecewise_cubic_bezier_interpolation);
    get_flag("adaptive_steering_angle_and_step_size", p.adaptive_steering_angle_and_step_size);
    get_flag("analytical_expansion_every_point", p.analytical_expansion_every_point);
    get_flag("analytical_expansion", p.analytical_expansion);
    get_float("penalty_turning", p.penalty_turning);
    get_float("penalty_reverse", p.penalty_reverse);
    get_float("penalty_change_of_direction", p.penalty_change_of_direction);
    get_float("steering_angle", p.steering_angle);
    get_param("max_iterations", p.max_iterations);
    get_float("tie_breaker", p.tie_breaker);
    get_float("goal_range", p.goal_range);
    get_float("curve_step_size", p.curve_step_size);
    get_flag("short_cut", p.short_cut);
    get_float("constant_density", p.constant_density);
}
#endif

HybridAStarPlanner::~HybridAStarPlanner()
{
    if (ctx_)
        hyas_destroy(ctx_);
}

void HybridAStarPlanner::initialize(std::string name, costmap_2d::Costmap2DROS *costmap_ros)
{
    initialize(name, costmap_ros->getCostmap(), costmap_ros->getGlobalFrameID());
}

void HybridAStarPlanner::initialize(std::string name, costmap_2d::Costmap2D *_costmap, std::string frame_id)
{
    if (initialized_)
    {
        HYAS_LOG_WARN("This planner has already been initialized, you can't call it twice, doing nothing");
        return;
    }
    costmap = _costmap;
    frame_id_ = frame_id;
#ifdef HYAS_HAVE_ROS
    if (!params_set_) // params_.loadRosParamFromNodeHandle(nh), src/hybrid_a_star_planner.cpp:59
        loadRosParams();
#endif
    ctx_ = hyas_create(&params_, device_);
    if (!ctx_)
    {
        HYAS_LOG_ERROR("no usable HIP device %d or unsupported parameters: the planner stays uninitialised", device_);
        return;
    }
    int rc = hyas_set_costmap(ctx_, costmap->getCharMap(), costmap->getSizeInCellsX(), costmap->getSizeInCellsY(),
                              costmap->getResolution(), costmap->getOriginX(), costmap->getOriginY(), /*build_table=*/1);
    if (rc != HYAS_OK)
    {
        HYAS_LOG_ERROR("hyas_set_costmap failed (%d): %s", rc, hyas_last_error(ctx_));
        hyas_destroy(ctx_);
        ctx_ = nullptr;
        return;
    }
#ifdef HYAS_HAVE_ROS
    ros::NodeHandle nh("~/" + name);
    plan_pub_ = nh.advertise<nav_msgs::Path>("plan", 1);
    path_vehicles_pub_ = nh.advertise<visualization_msgs::MarkerArray>("pathVehicle", 1); // :71
    make_plan_srv_ = nh.advertiseService("make_plan", &HybridAStarPlanner::makePlanService, this);
#else
    (void)name;
#endif
    initialized_ = true;
}

const char *HybridAStarPlanner::lastError() const { return ctx_ ? hyas_last_error(ctx_) : "planner not initialised"; }

static void to_xyt(const geometry_msgs::PoseStamped &p, float *xyt)
{
    xyt[0] = (float)p.pose.position.x, xyt[1] = (float)p.pose.position.y;
    xyt[2] = (float)tf::getYaw(p.pose.orientation); // normalised to [0, 2 pi) by the library (src/utility.cpp:7-13)
}

bool HybridAStarPlanner::makePlans(const std::vector<geometry_msgs::PoseStamped> &starts,
                                   const std::vector<geometry_msgs::PoseStamped> &goals,
                                   std::vector<std::vector<geometry_msgs::PoseStamped>> &plans, std::vector<hyas_result> *results)
{
    plans.clear();
    if (!initialized_)
    {
        HYAS_LOG_ERROR("The planner has not been initialized, please call initialize() to use the planner");
        return false;
    }
    const size_t n = starts.size();
    if (goals.size() != n)
        return false;
    std::vector<float> s(3 * n), g(3 * n);
    for (size_t i = 0; i < n; ++i)
        to_xyt(starts[i], &s[3 * i]), to_xyt(goals[i], &g[3 * i]);
    std::vector<hyas_result> res(n);
    if (hyas_plan_batch(ctx_, s.data(), g.data(), n, res.data()) != HYAS_OK)
    {
        HYAS_LOG_ERROR("hyas_plan_batch failed: %s", hyas_last_error(ctx_));
        return false;
    }
    plans.resize(n);
    std::vector<float> xyt;
    for (size_t i = 0; i < n; ++i)
    {
        if (res[i].status == HYAS_STATUS_BAD_POSE || res[i].status == HYAS_STATUS_NO_PATH || res[i].path_len <= 0)
            continue;
        xyt.resize(3 * (size_t)res[i].path_len);
        hyas_get_path(ctx_, i, xyt.data(), (size_t)res[i].path_len);
        plans[i].reserve((size_t)res[i].path_len);
        for (int k = 0; k < res[i].path_len; ++k)
        {
            geometry_msgs::PoseStamped p = starts[i]; // header as the start's (src/utility.cpp:21-35)
            p.header.frame_id = frame_id_;
            p.pose.position.x = xyt[3 * k], p.pose.position.y = xyt[3 * k + 1], p.pose.position.z = 0;
            p.pose.orientation = tf::createQuaternionMsgFromYaw(xyt[3 * k + 2]);
            plans[i].push_back(p);
        }
    }
    if (results)
        *results = res;
    if (n)
        last_ = res[n - 1];
    return true;
}

bool HybridAStarPlanner::makePlan(const geometry_msgs::PoseStamped &start, const geometry_msgs::PoseStamped &goal,
                                  std::vector<geometry_msgs::PoseStamped> &plan)
{
    plan.clear();
    std::vector<std::vector<geometry_msgs::PoseStamped>> plans;
    if (!makePlans({start}, {goal}, plans) || plans[0].empty())
        return false;
    plan.swap(plans[0]);
    publishPlan(plan);
    return true;
}

bool HybridAStarPlanner::makePlanService(nav_msgs::GetPlan::Request &req, nav_msgs::GetPlan::Response &resp)
{
    makePlan(req.start, req.goal, resp.plan.poses);
    resp.plan.header.frame_id = frame_id_;
    return true; // the service always answers; the plan may be empty (:89-97)
}

void HybridAStarPlanner::publishPlan(const std::vector<geometry_msgs::PoseStamped> &path)
{
    if (!initialized_)
        return;
    published_.poses = path;
    published_.header.frame_id = frame_id_;
#ifdef HYAS_HAVE_ROS
    published_.header.stamp = ros::Time::now();
    plan_pub_.publish(published_);
#endif
}

// One cube of the vehicle's size per pose (PathPublisher::AddVehicle, src/path_publisher.cpp:84-119) on ~/pathVehicle.
// The reference hands the publisher to calculatePath (:116), which never publishes on it; callers that want the
// markers call this after makePlan.
void HybridAStarPlanner::publishPathNodes(const std::vector<geometry_msgs::PoseStamped> &path)
{
    if (!initialized_)
        return;
#ifdef HYAS_HAVE_ROS
    visualization_msgs::MarkerArray vehicles;
    int i = 0;
    for (const geometry_msgs::PoseStamped &p : path)
    {
        visualization_msgs::Marker v;
        if (i == 1)
            v.action = 3; // delete all previous markers
        v.header.frame_id = "map";
        v.header.stamp = ros::Time(0);
        v.id = i++;
        v.type = visualization_msgs::Marker::CUBE;
        v.scale.x = params_.vehicle_length, v.scale.y = params_.vehicle_width, v.scale.z = 1;
        v.color.a = 0.1, v.color.r = 1, v.color.g = 0, v.color.b = 0;
        v.pose = p.pose;
        vehicles.markers.push_back(v);
    }
    path_vehicles_pub_.publish(vehicles);
#else
    path_nodes_published_ = path.size();
#endif
}

} // namespace HybridAStar
