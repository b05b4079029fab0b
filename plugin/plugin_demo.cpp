// Drives the plugin class the way move_base does: construct, initialize with a
// costmap, makePlan.  Usage:
//   plugin_demo MAP.bin W H RES OX OY POSITION_RESOLUTION  sx sy syaw  gx gy gyaw [more start/goal sextuples]
// MAP.bin: W*H cost bytes, row-major (costmap_2d order).  Prints one block per
// query: "status explored expanded path_len" then path_len lines "x y yaw".
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hybrid_a_star_planner.h"

int main(int argc, char **argv)
{
    if (argc < 14 || (argc - 8) % 6 != 0)
    {
        std::fprintf(stderr, "usage: %s MAP.bin W H RES OX OY PR  sx sy syaw gx gy gyaw ...\n", argv[0]);
        return 2;
    }
    const unsigned w = (unsigned)std::atoi(argv[2]), h = (unsigned)std::atoi(argv[3]);
    costmap_2d::Costmap2D map(w, h, std::atof(argv[4]), std::atof(argv[5]), std::atof(argv[6]));
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(map.getCharMap(), 1, (size_t)w * h, f) != (size_t)w * h)
    {
        std::fprintf(stderr, "cannot read %s\n", argv[1]);
        return 2;
    }
    std::fclose(f);
    costmap_2d::Costmap2DROS ros_map(&map, "map");

    HybridAStar::HybridAStarPlanner planner;
    hyas_params p;
    hyas_hot_path_params(&p);
    p.position_resolution = std::atoi(argv[7]);
    planner.setParams(p);
    nav_core::BaseGlobalPlanner *base = &planner; // what pluginlib hands to move_base
    base->initialize("HybridAStarPlanner", &ros_map);

    int failures = 0;
    for (int a = 8; a + 5 < argc; a += 6)
    {
        geometry_msgs::PoseStamped s, g;
        s.header.frame_id = g.header.frame_id = "map";
        s.pose.position.x = std::atof(argv[a]), s.pose.position.y = std::atof(argv[a + 1]);
        s.pose.orientation = tf::createQuaternionMsgFromYaw(std::atof(argv[a + 2]));
        g.pose.position.x = std::atof(argv[a + 3]), g.pose.position.y = std::atof(argv[a + 4]);
        g.pose.orientation = tf::createQuaternionMsgFromYaw(std::atof(argv[a + 5]));
        std::vector<geometry_msgs::PoseStamped> plan;
        const bool ok = base->makePlan(s, g, plan);
        const hyas_result &r = planner.lastResult();
        std::printf("%d %d %d %zu\n", ok ? r.status : -1, r.nodes_explored, r.nodes_expanded, plan.size());
        for (const auto &q : plan)
            std::printf("%.9g %.9g %.9g\n", q.pose.position.x, q.pose.position.y, tf::getYaw(q.pose.orientation));
        failures += ok ? 0 : 1;
    }
    return failures ? 1 : 0;
}
