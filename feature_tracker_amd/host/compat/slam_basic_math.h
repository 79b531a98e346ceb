// slam_basic_math.h — stand-in for Slam_Utility's math header (only what the tracker API needs).
#ifndef _SLAM_UTILITY_BASIC_MATH_H_
#define _SLAM_UTILITY_BASIC_MATH_H_
#include <algorithm>
#include <cmath>

#include "basic_type.h"
#endif
