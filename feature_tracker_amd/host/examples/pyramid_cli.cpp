// pyramid_cli — self-checks of the host-side ImagePyramid stand-in with the device runtime linked
// (tests/test_host_cpp_gpu.py runs it on the GPU box):
//   1. CreateImagePyramid builds levels >= 1 in HBM (no host loop, ONE upload of level 0) and a caller that reads a level
//      image gets, lazily, exactly the truncating 2x2 box mean;
//   2. a frame written IN PLACE into the buffer level 0 aliases is noticed (content stamp) and tracked, not the stale copy;
//   3. two threads with their own tracker objects read the same const pyramids at once and get the single-threaded result.
//
//   pyramid_cli <ref.pgm|png> <cur.pgm|png> <levels>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "optical_flow_basic_klt.h"
#include "optical_flow_lssd_klt.h"
#include "slam_memory.h"
#include "visualizor_2d.h"

using namespace feature_tracker;

namespace {

void MakePyramid(ImagePyramid &pyr, GrayImage &image, uint32_t levels) {
    pyr.SetPyramidBuff((uint8_t *)SlamMemory::Malloc(sizeof(uint8_t) * image.rows() * image.cols()), true);
    pyr.SetRawImage(image.data(), image.rows(), image.cols());
    pyr.CreateImagePyramid(levels);
}

std::vector<Vec2> GridFeatures(const GrayImage &image, int step) {
    std::vector<Vec2> uv;
    for (int r = 40; r < image.rows() - 40; r += step) {
        for (int c = 40; c < image.cols() - 40; c += step) {
            uv.emplace_back(c + 0.25f, r + 0.75f);
        }
    }
    return uv;
}

bool SameBits(const std::vector<Vec2> &a, const std::vector<Vec2> &b) {
    return a.size() == b.size() && (a.empty() || std::memcmp(a[0].data(), b[0].data(), sizeof(float) * 2 * a.size()) == 0);
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: pyramid_cli ref cur levels\n");
        return 2;
    }
    GrayImage ref_image, cur_image;
    if (!slam_visualizor::Visualizor2D::LoadImage(argv[1], ref_image) || !slam_visualizor::Visualizor2D::LoadImage(argv[2], cur_image)) {
        std::fprintf(stderr, "cannot load images\n");
        return 2;
    }
    const uint32_t levels = static_cast<uint32_t>(std::atoi(argv[3]));
    int failures = 0;

    // ---- 1. device-built levels, lazy host copies ----
    ImagePyramid ref_pyramid, cur_pyramid;
    MakePyramid(ref_pyramid, ref_image, levels);
    MakePyramid(cur_pyramid, cur_image, levels);
    std::printf("built_on_device %d\n", (ref_pyramid.device_twin() && !ref_pyramid.host_levels_valid()) ? 1 : 0);
    const std::vector<Vec2> ref_uv = GridFeatures(ref_image, 23);
    OpticalFlowBasicKlt klt;
    klt.options().kMethod = OpticalFlowMethod::kInverse;
    klt.options().kMaxTrackPointsNumber = 100000;
    std::vector<Vec2> cur_uv;
    std::vector<uint8_t> status;
    const bool ok = klt.TrackFeatures(ref_pyramid, cur_pyramid, ref_uv, cur_uv, status);
    std::printf("tracked_without_host_levels %d\n", (ok && !ref_pyramid.host_levels_valid() && !cur_pyramid.host_levels_valid()) ? 1 : 0);
    bool levels_equal = true;
    for (uint32_t i = 1; i < ref_pyramid.level(); ++i) {
        const GrayImage &lo = ref_pyramid.GetImageConst(i);      // materialises the host copies (download from the twin)
        const GrayImage &hi = ref_pyramid.GetImageConst(i - 1);
        for (int32_t r = 0; r < lo.rows() && levels_equal; ++r) {
            for (int32_t c = 0; c < lo.cols(); ++c) {
                const uint32_t want = (uint32_t(hi.GetPixelValueNoCheck(2 * r, 2 * c)) + hi.GetPixelValueNoCheck(2 * r, 2 * c + 1) +
                                       hi.GetPixelValueNoCheck(2 * r + 1, 2 * c) + hi.GetPixelValueNoCheck(2 * r + 1, 2 * c + 1)) >> 2;
                if (lo.GetPixelValueNoCheck(r, c) != want) {
                    levels_equal = false;
                    break;
                }
            }
        }
    }
    std::printf("levels_equal_box_mean %d\n", levels_equal ? 1 : 0);
    failures += (ok && levels_equal) ? 0 : 1;
    // reading the host copies must not have invalidated the twins: the same call again, the same bits
    std::vector<Vec2> again;
    std::vector<uint8_t> again_status;
    klt.TrackFeatures(ref_pyramid, cur_pyramid, ref_uv, again, again_status);
    std::printf("same_after_host_read %d\n", SameBits(cur_uv, again) ? 1 : 0);
    failures += SameBits(cur_uv, again) ? 0 : 1;

    // ---- 2. in-place frame reuse (single-level pyramids: the reference reads whatever the buffer holds) ----
    {
        std::vector<uint8_t> frame(static_cast<size_t>(ref_image.rows()) * ref_image.cols());
        std::memcpy(frame.data(), ref_image.data(), frame.size());
        ImagePyramid in_place, fixed;
        uint8_t dummy_a[16], dummy_b[16];
        in_place.SetPyramidBuff(dummy_a, false);
        in_place.SetRawImage(frame.data(), ref_image.rows(), ref_image.cols());
        in_place.CreateImagePyramid(1);
        fixed.SetPyramidBuff(dummy_b, false);
        fixed.SetRawImage(cur_image.data(), cur_image.rows(), cur_image.cols());
        fixed.CreateImagePyramid(1);
        OpticalFlowBasicKlt one;
        one.options().kMaxTrackPointsNumber = 100000;
        std::vector<Vec2> before, after, fresh;
        std::vector<uint8_t> st;
        one.TrackFeatures(in_place, fixed, ref_uv, before, st);  // ref -> cur
        std::memcpy(frame.data(), cur_image.data(), frame.size());  // the "next frame" lands in the same buffer
        st.clear();
        one.TrackFeatures(in_place, fixed, ref_uv, after, st);   // must be cur -> cur now
        ImagePyramid same;
        uint8_t dummy_c[16];
        same.SetPyramidBuff(dummy_c, false);
        same.SetRawImage(cur_image.data(), cur_image.rows(), cur_image.cols());
        same.CreateImagePyramid(1);
        st.clear();
        one.TrackFeatures(same, fixed, ref_uv, fresh, st);
        const bool noticed = SameBits(after, fresh) && !SameBits(after, before);
        std::printf("in_place_overwrite_noticed %d\n", noticed ? 1 : 0);
        failures += noticed ? 0 : 1;
    }

    // ---- 3. two threads, own tracker objects, shared const pyramids ----
    {
        std::vector<Vec2> serial_a, serial_b;
        std::vector<uint8_t> st;
        OpticalFlowBasicKlt a0;
        a0.options().kMaxTrackPointsNumber = 100000;
        a0.TrackFeatures(ref_pyramid, cur_pyramid, ref_uv, serial_a, st);
        OpticalFlowLssdKlt b0;
        b0.options().kMaxTrackPointsNumber = 100000;
        st.clear();
        b0.TrackFeatures(cur_pyramid, ref_pyramid, ref_uv, serial_b, st);
        bool good[2] = {true, true};
        auto work = [&](int which) {
            for (int rep = 0; rep < 20; ++rep) {
                std::vector<Vec2> out;
                std::vector<uint8_t> s;
                if (which == 0) {
                    OpticalFlowBasicKlt t;
                    t.options().kMaxTrackPointsNumber = 100000;
                    good[0] = t.TrackFeatures(ref_pyramid, cur_pyramid, ref_uv, out, s) && SameBits(out, serial_a) && good[0];
                } else {
                    OpticalFlowLssdKlt t;
                    t.options().kMaxTrackPointsNumber = 100000;
                    good[1] = t.TrackFeatures(cur_pyramid, ref_pyramid, ref_uv, out, s) && SameBits(out, serial_b) && good[1];
                }
            }
        };
        std::thread t0(work, 0), t1(work, 1);
        t0.join();
        t1.join();
        std::printf("two_threads_equal_serial %d\n", (good[0] && good[1]) ? 1 : 0);
        failures += (good[0] && good[1]) ? 0 : 1;
    }
    std::printf("%s\n", failures == 0 ? "PASS" : "FAIL");
    return failures == 0 ? 0 : 1;
}
