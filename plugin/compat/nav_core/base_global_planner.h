// nav_core::BaseGlobalPlanner as ROS navigation declares it, for a build without ROS.
#pragma once
#include <string>
#include <vector>

#include "costmap_2d/costmap_2d.h"
#include "geometry_msgs/PoseStamped.h"

namespace nav_core
{
class BaseGlobalPlanner
{
  public:
    virtual bool makePlan(const geometry_msgs::PoseStamped &start, const geometry_msgs::PoseStamped &goal,
                          std::vector<geometry_msgs::PoseStamped> &plan) = 0;
    virtual void initialize(std::string name, costmap_2d::Costmap2DROS *costmap_ros) = 0;
    virtual ~BaseGlobalPlanner() {}

  protected:
    BaseGlobalPlanner() {}
};
} // namespace nav_core
