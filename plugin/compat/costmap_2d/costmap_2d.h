// The part of costmap_2d::Costmap2D the planner reads (size, resolution,
// origin, the cost bytes), for a build without ROS.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace costmap_2d
{
static const unsigned char NO_INFORMATION = 255;
static const unsigned char LETHAL_OBSTACLE = 254;
static const unsigned char INSCRIBED_INFLATED_OBSTACLE = 253;
static const unsigned char FREE_SPACE = 0;

class Costmap2D
{
  public:
    Costmap2D(unsigned int cells_x, unsigned int cells_y, double resolution, double origin_x, double origin_y, unsigned char fill = 0)
        : sx_(cells_x), sy_(cells_y), res_(resolution), ox_(origin_x), oy_(origin_y), cost_((size_t)cells_x * cells_y, fill)
    {
    }
    unsigned int getSizeInCellsX() const { return sx_; }
    unsigned int getSizeInCellsY() const { return sy_; }
    double getResolution() const { return res_; }
    double getOriginX() const { return ox_; }
    double getOriginY() const { return oy_; }
    unsigned char *getCharMap() { return cost_.data(); }
    const unsigned char *getCharMap() const { return cost_.data(); }
    unsigned char getCost(unsigned int mx, unsigned int my) const { return cost_[(size_t)my * sx_ + mx]; }
    void setCost(unsigned int mx, unsigned int my, unsigned char c) { cost_[(size_t)my * sx_ + mx] = c; }
    bool worldToMap(double wx, double wy, unsigned int &mx, unsigned int &my) const
    {
        if (wx < ox_ || wy < oy_)
            return false;
        mx = (unsigned int)((wx - ox_) / res_);
        my = (unsigned int)((wy - oy_) / res_);
        return mx < sx_ && my < sy_;
    }
    void mapToWorld(unsigned int mx, unsigned int my, double &wx, double &wy) const
    {
        wx = ox_ + (mx + 0.5) * res_;
        wy = oy_ + (my + 0.5) * res_;
    }

  private:
    unsigned int sx_, sy_;
    double res_, ox_, oy_;
    std::vector<unsigned char> cost_;
};

// costmap_2d::Costmap2DROS: owner of a Costmap2D plus its frame
class Costmap2DROS
{
  public:
    Costmap2DROS(Costmap2D *map, std::string global_frame) : map_(map), frame_(std::move(global_frame)) {}
    Costmap2D *getCostmap() { return map_; }
    std::string getGlobalFrameID() const { return frame_; }

  private:
    Costmap2D *map_;
    std::string frame_;
};
} // namespace costmap_2d
