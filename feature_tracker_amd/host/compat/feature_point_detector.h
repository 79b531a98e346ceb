// feature_point_detector.h — umbrella header of the Feature_Detector stand-in.
#ifndef _FEATURE_POINT_DETECTOR_H_
#define _FEATURE_POINT_DETECTOR_H_
#include "feature_point_harris_detector.h"
#endif
