// Stand-in: utils/root_finder.hpp includes <ros/ros.h> but only uses it in commented-out logging; generateESDF3d
// declares two unused ros::Time variables (Gridmap3D.cpp:451).  TEST INFRASTRUCTURE ONLY.
#pragma once
namespace ros { struct Time {}; }
