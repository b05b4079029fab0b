// Minimal geometry_msgs types for building the planner plugin without ROS
// (same member names as the generated ROS messages).
#pragma once
#include <cstdint>
#include <string>

namespace std_msgs
{
struct Header
{
    uint32_t seq = 0;
    double stamp = 0; // ros::Time in a ROS build
    std::string frame_id;
};
} // namespace std_msgs

namespace geometry_msgs
{
struct Point
{
    double x = 0, y = 0, z = 0;
};
struct Quaternion
{
    double x = 0, y = 0, z = 0, w = 1;
};
struct Pose
{
    Point position;
    Quaternion orientation;
};
struct PoseStamped
{
    std_msgs::Header header;
    Pose pose;
};
} // namespace geometry_msgs
