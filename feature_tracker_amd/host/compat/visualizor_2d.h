// visualizor_2d.h — headless stand-in for Visualizor2D (un-vendored GUI library).
// LoadImage decodes 8-bit PNGs (gray, gray+alpha, RGB, RGBA; non-interlaced) with zlib, or binary PGM
// (P5).  The Show* calls draw nothing; when $FTK_VIS_DIR is set they dump the tracked features as
// CSV there so that a run can be inspected.  WaitKey returns immediately.
#ifndef _SLAM_VISUALIZOR_2D_H_
#define _SLAM_VISUALIZOR_2D_H_

#include <cstdint>
#include <string>
#include <vector>

#include "basic_type.h"
#include "datatype_image.h"

namespace slam_visualizor {

class Visualizor2D {
public:
    // Decodes `file` into a buffer owned by `image`.  Returns false if the file cannot be read/decoded.
    static bool LoadImage(const std::string &file, GrayImage &image);
    static bool SaveImage(const std::string &file, const GrayImage &image);  // binary PGM

    static void ShowImageWithDetectedFeatures(const std::string &title, const GrayImage &image, const std::vector<Vec2> &pixel_uv);
    static void ShowImageWithTrackedFeatures(const std::string &title, const GrayImage &cur_image, const std::vector<Vec2> &ref_pixel_uv,
                                             const std::vector<Vec2> &cur_pixel_uv, const std::vector<uint8_t> &status,
                                             uint8_t min_valid_status_value = 2);
    static void ShowImageWithTrackedFeatures(const std::string &title, const GrayImage &ref_image, const GrayImage &cur_image,
                                             const std::vector<Vec2> &ref_pixel_uv, const std::vector<Vec2> &cur_pixel_uv,
                                             const std::vector<uint8_t> &status, uint8_t min_valid_status_value = 2);
    static void WaitKey(int32_t delay_ms) { (void)delay_ms; }
};

}  // namespace slam_visualizor

#endif  // _SLAM_VISUALIZOR_2D_H_
