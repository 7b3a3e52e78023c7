// Empty stand-in: utils/root_finder.hpp includes <ros/ros.h> but only uses it in commented-out logging.
// TEST INFRASTRUCTURE ONLY.
#pragma once
