#pragma once
#include "nav_msgs/Path.h"

namespace nav_msgs
{
struct GetPlan
{
    struct Request
    {
        geometry_msgs::PoseStamped start, goal;
        float tolerance = 0;
    };
    struct Response
    {
        Path plan;
    };
};
} // namespace nav_msgs
