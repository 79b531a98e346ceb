// track_cli — drives the host-side C++ API (the reference's class names) from the command line so
// that tests can compare it with the oracle bit for bit.
//
//   track_cli <model basic|affine|lssd> <method 0..4> <levels (0 = single-image overload)> <half_rows> <half_cols>
//             <ref.pgm|png> <cur.pgm|png> <features.txt> [max_points] [prior a00 a01 a10 a11] [luminance 0|1]
//
// features.txt: one "ref_u ref_v [cur_u cur_v status]" per line (hex floats accepted).
// Output: one "cur_u_bits cur_v_bits status iterations" line per feature (float bit patterns in hex).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "optical_flow_affine_klt.h"
#include "optical_flow_basic_klt.h"
#include "optical_flow_lssd_klt.h"
#include "slam_memory.h"
#include "visualizor_2d.h"

using namespace feature_tracker;

int main(int argc, char **argv) {
    if (argc < 9) {
        std::fprintf(stderr, "usage: see the header of track_cli.cpp\n");
        return 2;
    }
    const std::string model = argv[1];
    const int method = std::atoi(argv[2]);
    const int levels = std::atoi(argv[3]);
    const int half_rows = std::atoi(argv[4]), half_cols = std::atoi(argv[5]);
    GrayImage ref_image, cur_image;
    if (!slam_visualizor::Visualizor2D::LoadImage(argv[6], ref_image) || !slam_visualizor::Visualizor2D::LoadImage(argv[7], cur_image)) {
        std::fprintf(stderr, "cannot load images\n");
        return 2;
    }
    std::vector<Vec2> ref_uv, cur_uv;
    std::vector<uint8_t> status;
    {
        std::ifstream in(argv[8]);
        std::string line;
        bool have_prediction = false;
        while (std::getline(in, line)) {
            if (line.empty()) continue;
            float v[4];
            int st = 0;
            const int n = std::sscanf(line.c_str(), "%a %a %a %a %d", &v[0], &v[1], &v[2], &v[3], &st);
            if (n < 2) continue;
            ref_uv.emplace_back(v[0], v[1]);
            if (n >= 5) {
                have_prediction = true;
                cur_uv.emplace_back(v[2], v[3]);
                status.push_back(static_cast<uint8_t>(st));
            }
        }
        if (!have_prediction) {
            cur_uv.clear();
            status.clear();
        }
    }

    std::unique_ptr<OpticalFlow> klt;
    float prior[4] = {1, 0, 0, 1};
    if (argc >= 14) {
        for (int i = 0; i < 4; ++i) prior[i] = std::strtof(argv[10 + i], nullptr);
    }
    if (model == "basic") {
        klt.reset(new OpticalFlowBasicKlt());
    } else if (model == "affine") {
        auto *p = new OpticalFlowAffineKlt();
        p->predict_affine()(0, 0) = prior[0];
        p->predict_affine()(0, 1) = prior[1];
        p->predict_affine()(1, 0) = prior[2];
        p->predict_affine()(1, 1) = prior[3];
        klt.reset(p);
    } else {
        auto *p = new OpticalFlowLssdKlt();
        p->predict_R_cr()(0, 0) = prior[0];
        p->predict_R_cr()(0, 1) = prior[1];
        p->predict_R_cr()(1, 0) = prior[2];
        p->predict_R_cr()(1, 1) = prior[3];
        p->consider_patch_luminance() = argc >= 15 && std::atoi(argv[14]) != 0;
        klt.reset(p);
    }
    klt->options().kMethod = static_cast<OpticalFlowMethod>(method);
    klt->options().kPatchRowHalfSize = half_rows;
    klt->options().kPatchColHalfSize = half_cols;
    if (argc >= 10) {
        klt->options().kMaxTrackPointsNumber = static_cast<uint32_t>(std::atoi(argv[9]));
    }

    bool ok;
    if (levels <= 0) {
        ok = klt->TrackFeatures(ref_image, cur_image, ref_uv, cur_uv, status);
    } else {
        ImagePyramid ref_pyramid, cur_pyramid;
        ref_pyramid.SetPyramidBuff((uint8_t *)SlamMemory::Malloc(sizeof(uint8_t) * ref_image.rows() * ref_image.cols()), true);
        cur_pyramid.SetPyramidBuff((uint8_t *)SlamMemory::Malloc(sizeof(uint8_t) * cur_image.rows() * cur_image.cols()), true);
        ref_pyramid.SetRawImage(ref_image.data(), ref_image.rows(), ref_image.cols());
        cur_pyramid.SetRawImage(cur_image.data(), cur_image.rows(), cur_image.cols());
        ref_pyramid.CreateImagePyramid(levels);
        cur_pyramid.CreateImagePyramid(levels);
        ok = klt->TrackFeatures(ref_pyramid, cur_pyramid, ref_uv, cur_uv, status);
        // a second call on the unchanged pyramids must reuse the device twins and give the same answer
        std::vector<Vec2> again;
        std::vector<uint8_t> again_status;
        if (ok && !klt->TrackFeatures(ref_pyramid, cur_pyramid, ref_uv, again, again_status)) {
            ok = false;
        }
    }
    std::printf("ok %d name %s\n", ok ? 1 : 0, klt->OpticalFlowMethodName().c_str());
    if (!ok) {
        std::printf("error %s\n", klt->last_error().c_str());
        return ref_uv.empty() ? 0 : 1;
    }
    for (size_t i = 0; i < cur_uv.size(); ++i) {
        uint32_t ub, vb;
        std::memcpy(&ub, &cur_uv[i].x(), 4);
        std::memcpy(&vb, &cur_uv[i].y(), 4);
        std::printf("%08x %08x %d %u\n", ub, vb, int(status[i]), klt->last_iterations().empty() ? 0u : klt->last_iterations()[i]);
    }
    return 0;
}
