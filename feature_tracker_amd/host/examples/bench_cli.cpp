// bench_cli — latency of the C++ drop-in path as a caller of the reference's classes sees it (no Python in the loop):
// pyramids built once, then TrackFeatures timed per call with the pyramids' device twins already resident.
//   bench_cli <ref.png|pgm> <cur.png|pgm> [n=300] [levels=4] [half=6] [reps=200]
// Features: Harris corners first, topped up with a uniform grid to n; every model x method; median / min of `reps` calls.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "feature_point_harris_detector.h"
#include "optical_flow_affine_klt.h"
#include "optical_flow_basic_klt.h"
#include "optical_flow_lssd_klt.h"
#include "slam_memory.h"
#include "visualizor_2d.h"

using namespace slam_visualizor;

template <typename Tracker>
static void Run(const char *name, const ImagePyramid &ref_pyramid, const ImagePyramid &cur_pyramid, const std::vector<Vec2> &ref_pixel_uv, int half,
                int method, int reps) {
    Tracker klt;
    klt.options().kPatchRowHalfSize = half;
    klt.options().kPatchColHalfSize = half;
    klt.options().kMaxTrackPointsNumber = static_cast<uint32_t>(ref_pixel_uv.size());
    klt.options().kMethod = static_cast<feature_tracker::OpticalFlowMethod>(method);
    std::vector<Vec2> cur_pixel_uv;
    std::vector<uint8_t> status;
    std::vector<double> us;
    int tracked = 0;
    for (int r = 0; r < reps + 3; ++r) {
        cur_pixel_uv.clear();
        status.clear();
        const auto t0 = std::chrono::steady_clock::now();
        const bool ok = klt.TrackFeatures(ref_pyramid, cur_pyramid, ref_pixel_uv, cur_pixel_uv, status);
        const auto t1 = std::chrono::steady_clock::now();
        if (!ok) {
            std::printf("%s method %d: TrackFeatures failed: %s\n", name, method, klt.last_error().c_str());
            return;
        }
        if (r >= 3) {  // the first calls pay the context, the code objects and the pyramid upload
            us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        tracked = 0;
        for (uint8_t s : status) tracked += s == static_cast<uint8_t>(feature_tracker::TrackStatus::kTracked);
    }
    std::sort(us.begin(), us.end());
    std::printf("%-8s method %d: %zu features, %d tracked, TrackFeatures median %.1f us, min %.1f us\n", name, method, ref_pixel_uv.size(), tracked,
                us[us.size() / 2], us.front());
}

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: bench_cli ref cur [n] [levels] [half] [reps]\n");
        return 2;
    }
    const int n = argc > 3 ? std::atoi(argv[3]) : 300;
    const int levels = argc > 4 ? std::atoi(argv[4]) : 4;
    const int half = argc > 5 ? std::atoi(argv[5]) : 6;
    const int reps = argc > 6 ? std::atoi(argv[6]) : 200;
    GrayImage ref_image, cur_image;
    if (!Visualizor2D::LoadImage(argv[1], ref_image) || !Visualizor2D::LoadImage(argv[2], cur_image)) {
        std::fprintf(stderr, "cannot load the images\n");
        return 1;
    }
    ImagePyramid ref_pyramid, cur_pyramid;
    ref_pyramid.SetPyramidBuff((uint8_t *)SlamMemory::Malloc(sizeof(uint8_t) * ref_image.rows() * ref_image.cols()), true);
    cur_pyramid.SetPyramidBuff((uint8_t *)SlamMemory::Malloc(sizeof(uint8_t) * cur_image.rows() * cur_image.cols()), true);
    ref_pyramid.SetRawImage(ref_image.data(), ref_image.rows(), ref_image.cols());
    cur_pyramid.SetRawImage(cur_image.data(), cur_image.rows(), cur_image.cols());
    ref_pyramid.CreateImagePyramid(levels);
    cur_pyramid.CreateImagePyramid(levels);

    std::vector<Vec2> features;
    feature_detector::FeaturePointHarrisDetector detector;
    detector.options().kMinFeatureDistance = 15;
    detector.options().kMinValidResponse = 40.0f;
    detector.DetectGoodFeatures(ref_image, static_cast<uint32_t>(n), features);
    const int margin = 4 * half + 8;
    for (int k = 0; static_cast<int>(features.size()) < n; ++k) {  // top up on a jittered grid
        const float u = margin + (k * 37) % (ref_image.cols() - 2 * margin) + 0.25f * (k % 4);
        const float v = margin + (k * 53) % (ref_image.rows() - 2 * margin) + 0.5f * (k % 2);
        features.emplace_back(u, v);
    }
    std::printf("image %d x %d, %d levels, %d x %d patch, %zu features, %d timed calls each\n", ref_image.cols(), ref_image.rows(), levels, 2 * half + 1,
                2 * half + 1, features.size(), reps);
    for (int method = 0; method < 3; ++method) {
        Run<feature_tracker::OpticalFlowBasicKlt>("basic", ref_pyramid, cur_pyramid, features, half, method, reps);
        Run<feature_tracker::OpticalFlowAffineKlt>("affine", ref_pyramid, cur_pyramid, features, half, method, reps);
        Run<feature_tracker::OpticalFlowLssdKlt>("lssd", ref_pyramid, cur_pyramid, features, half, method, reps);
    }
    return 0;
}
