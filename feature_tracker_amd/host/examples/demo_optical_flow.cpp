// demo_optical_flow — headless counterpart of the reference's test_optical_flow program: load an
// image pair, detect Harris corners, build pyramids, run the three trackers, print time and yield.
//   demo_optical_flow <ref.png|pgm> <cur.png|pgm> [levels=4] [half=6] [method=2]
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "feature_point_harris_detector.h"
#include "optical_flow_affine_klt.h"
#include "optical_flow_basic_klt.h"
#include "optical_flow_lssd_klt.h"
#include "slam_log_reporter.h"
#include "slam_memory.h"
#include "tick_tock.h"
#include "visualizor_2d.h"

using namespace slam_visualizor;

template <typename Tracker>
static void Run(const char *name, GrayImage &ref_image, GrayImage &cur_image, int levels, int half, int method) {
    ImagePyramid ref_pyramid, cur_pyramid;
    ref_pyramid.SetPyramidBuff((uint8_t *)SlamMemory::Malloc(sizeof(uint8_t) * ref_image.rows() * ref_image.cols()), true);
    cur_pyramid.SetPyramidBuff((uint8_t *)SlamMemory::Malloc(sizeof(uint8_t) * cur_image.rows() * cur_image.cols()), true);
    ref_pyramid.SetRawImage(ref_image.data(), ref_image.rows(), ref_image.cols());
    cur_pyramid.SetRawImage(cur_image.data(), cur_image.rows(), cur_image.cols());

    std::vector<Vec2> ref_pixel_uv, cur_pixel_uv;
    std::vector<uint8_t> status;
    feature_detector::FeaturePointHarrisDetector detector;
    detector.options().kMinFeatureDistance = 25;
    detector.options().kMinValidResponse = 40.0f;
    detector.DetectGoodFeatures(ref_image, 300, ref_pixel_uv);

    Tracker klt;
    klt.options().kPatchRowHalfSize = half;
    klt.options().kPatchColHalfSize = half;
    klt.options().kMethod = static_cast<feature_tracker::OpticalFlowMethod>(method);

    for (int pass = 0; pass < 2; ++pass) {  // pass 0 pays context creation + kernel load
        cur_pixel_uv.clear();
        status.clear();
        TickTock timer;
        ref_pyramid.CreateImagePyramid(levels);
        cur_pyramid.CreateImagePyramid(levels);
        const bool ok = klt.TrackFeatures(ref_pyramid, cur_pyramid, ref_pixel_uv, cur_pixel_uv, status);
        const float ms = timer.TockTickInMillisecond();
        int tracked = 0;
        for (uint8_t s : status) tracked += s == static_cast<uint8_t>(feature_tracker::TrackStatus::kTracked);
        ReportInfo(name << " pass " << pass << ": ok " << ok << ", " << tracked << " / " << ref_pixel_uv.size() << " tracked, " << ms << " ms");
    }
    Visualizor2D::ShowImageWithTrackedFeatures(std::string(name) + " after multi tracking", cur_image, ref_pixel_uv, cur_pixel_uv, status);
}

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: demo_optical_flow ref cur [levels] [half] [method]\n");
        return 2;
    }
    GrayImage ref_image, cur_image;
    if (!Visualizor2D::LoadImage(argv[1], ref_image) || !Visualizor2D::LoadImage(argv[2], cur_image)) {
        ReportError("cannot load " << argv[1] << " / " << argv[2]);
        return 2;
    }
    const int levels = argc > 3 ? std::atoi(argv[3]) : 4, half = argc > 4 ? std::atoi(argv[4]) : 6, method = argc > 5 ? std::atoi(argv[5]) : 2;
    Run<feature_tracker::OpticalFlowBasicKlt>("Basic klt", ref_image, cur_image, levels, half, method);
    Run<feature_tracker::OpticalFlowAffineKlt>("Affine klt", ref_image, cur_image, levels, half, method);
    Run<feature_tracker::OpticalFlowLssdKlt>("Lssd klt", ref_image, cur_image, levels, half, method);
    return 0;
}
