// HybridAStar::HybridAStarPlanner -- the reference's nav_core global-planner
// plugin class (include/hybrid_a_star_planner.h:26-77) with the search behind
// it replaced by libhyas (include/hyas.h).  Same class name, base class, method
// signatures and plugin manifest entry, so move_base / pluginlib load it in
// place of the reference's library.
//
// Builds against real ROS headers when HYAS_HAVE_ROS is defined, otherwise
// against the minimal headers in plugin/compat (this image has no ROS).
#pragma once
#include <string>
#include <vector>

#ifdef HYAS_HAVE_ROS
#include <costmap_2d/costmap_2d_ros.h>
#include <geometry_msgs/PoseStamped.h>
#include <nav_core/base_global_planner.h>
#include <nav_msgs/GetPlan.h>
#include <nav_msgs/Path.h>
#include <ros/ros.h>
#include <tf/tf.h>
#include <visualization_msgs/MarkerArray.h>
#else
#include "costmap_2d/costmap_2d.h"
#include "geometry_msgs/PoseStamped.h"
#include "nav_core/base_global_planner.h"
#include "nav_msgs/GetPlan.h"
#include "nav_msgs/Path.h"
#include "tf/yaw.h"
#endif

#include "hyas.h"

namespace HybridAStar
{

class HybridAStarPlanner : public nav_core::BaseGlobalPlanner
{
  public:
    HybridAStarPlanner();
    HybridAStarPlanner(std::string name, costmap_2d::Costmap2DROS *costmap_ros);
    ~HybridAStarPlanner();

    // nav_core::BaseGlobalPlanner (include/hybrid_a_star_planner.h:35-41)
    void initialize(std::string name, costmap_2d::Costmap2DROS *costmap_ros) override;
    void initialize(std::string name, costmap_2d::Costmap2D *costmap, std::string frame_id);
    bool makePlan(const geometry_msgs::PoseStamped &start, const geometry_msgs::PoseStamped &goal,
                  std::vector<geometry_msgs::PoseStamped> &plan) override;
    void publishPlan(const std::vector<geometry_msgs::PoseStamped> &path);
    void publishPathNodes(const std::vector<geometry_msgs::PoseStamped> &path); // include/hybrid_a_star_planner.h:51
    bool makePlanService(nav_msgs::GetPlan::Request &req, nav_msgs::GetPlan::Response &resp);

    // Additions: parameters without a ROS parameter server, the GPU to use, many queries at once, and what
    // the last makePlan did.  Call setParams / setDevice before initialize.
    void setParams(const hyas_params &p) { params_ = p, params_set_ = true; } // overrides the ROS parameter load
    void setDevice(int device) { device_ = device; }
    bool makePlans(const std::vector<geometry_msgs::PoseStamped> &starts, const std::vector<geometry_msgs::PoseStamped> &goals,
                   std::vector<std::vector<geometry_msgs::PoseStamped>> &plans, std::vector<hyas_result> *results = nullptr);
    const hyas_result &lastResult() const { return last_; }
    const nav_msgs::Path &lastPublishedPlan() const { return published_; }
    const char *lastError() const;

  protected:
    bool initialized_ = false;
    std::string frame_id_;
    costmap_2d::Costmap2D *costmap = nullptr;

  private:
    hyas_params params_;
    bool params_set_ = false;
    size_t path_nodes_published_ = 0;
    int device_ = 0;
    hyas_ctx *ctx_ = nullptr;
    hyas_result last_{};
    nav_msgs::Path published_;
#ifdef HYAS_HAVE_ROS
    void loadRosParams();
    ros::Publisher plan_pub_;
    ros::Publisher path_vehicles_pub_;
    ros::ServiceServer make_plan_srv_;
#endif
};

} // namespace HybridAStar
