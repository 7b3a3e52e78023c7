// Stand-in: utils/root_finder.hpp includes <ros/ros.h> but only uses it in commented-out logging; generateESDF3d
// declares two unused ros::Time variables (Gridmap3D.cpp:451); the A* front end (front_end_Astar.hpp:205-221,243,280,351-353)
// times its neighbour test with ros::Time::now() for a log line only - the clock here always reads zero.
// TEST INFRASTRUCTURE ONLY.
#pragma once
namespace ros {
struct Duration { double toSec() const { return 0.0; } };
struct Time {
    static Time now() { return Time{}; }
    Duration operator-(const Time &) const { return Duration{}; }
};
}
