// tf::getYaw / tf::createQuaternionMsgFromYaw for a build without ROS.
#pragma once
#include <cmath>

#include "geometry_msgs/PoseStamped.h"

namespace tf
{
inline double getYaw(const geometry_msgs::Quaternion &q)
{
    return std::atan2(2.0 * (q.w * q.z + q.x * q.y), 1.0 - 2.0 * (q.y * q.y + q.z * q.z));
}
inline geometry_msgs::Quaternion createQuaternionMsgFromYaw(double yaw)
{
    geometry_msgs::Quaternion q;
    q.x = 0, q.y = 0, q.z = std::sin(0.5 * yaw), q.w = std::cos(0.5 * yaw);
    return q;
}
} // namespace tf
