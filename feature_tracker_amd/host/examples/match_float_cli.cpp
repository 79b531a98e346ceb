// match_float_cli — drives DescriptorMatcher<SuperpointDescriptorType / DiskDescriptorType> through
// subclasses written exactly like the reference's SuperpointMatcher / DiskMatcher
// (test/test_descriptor_matcher_superpoint.cpp:26-35, test_descriptor_matcher_disk.cpp:26-35).
//
//   match_float_cli <force|nearby> <256|128> <max_distance> <max_col> <max_row> <descriptors.bin>
//
// descriptors.bin (little endian): int32 n_ref, n_cur; float ref[n_ref][dim], cur[n_cur][dim],
// ref_uv[n_ref][2], cur_uv[n_cur][2].  Output: "ok <0|1>", "device <0|1>" (1 = the call went to the
// MI355X, i.e. the distance probe recognised the cosine formula), then per ref descriptor
// "index status u_bits v_bits" of the index- and pixel-returning overloads.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "descriptor_matcher.h"
#include "nn_feature_point_detector.h"

using namespace feature_detector;

namespace {
int g_distance_calls = 0;
}

class SuperpointMatcher: public feature_tracker::DescriptorMatcher<SuperpointDescriptorType> {
public:
    virtual float ComputeDistance(const SuperpointDescriptorType &descriptor_ref, const SuperpointDescriptorType &descriptor_cur) override {
        ++g_distance_calls;
        return 0.5f - descriptor_ref.dot(descriptor_cur) / descriptor_ref.norm() / descriptor_cur.norm() * 0.5f;
    }
};

class DiskMatcher: public feature_tracker::DescriptorMatcher<DiskDescriptorType> {
public:
    virtual float ComputeDistance(const DiskDescriptorType &descriptor_ref, const DiskDescriptorType &descriptor_cur) override {
        ++g_distance_calls;
        return 0.5f - descriptor_ref.dot(descriptor_cur) / descriptor_ref.norm() / descriptor_cur.norm() * 0.5f;
    }
};

template <typename Matcher, typename Descriptor>
int Run(bool nearby, float max_distance, int max_col, int max_row, FILE *f) {
    int32_t n_ref = 0, n_cur = 0;
    if (fread(&n_ref, 4, 1, f) != 1 || fread(&n_cur, 4, 1, f) != 1) {
        return 2;
    }
    std::vector<Descriptor> ref(n_ref), cur(n_cur);
    std::vector<Vec2> ref_uv(n_ref), cur_uv(n_cur);
    bool ok_read = true;
    if (n_ref) ok_read &= fread(ref[0].data(), sizeof(Descriptor), n_ref, f) == static_cast<size_t>(n_ref);
    if (n_cur) ok_read &= fread(cur[0].data(), sizeof(Descriptor), n_cur, f) == static_cast<size_t>(n_cur);
    if (n_ref) ok_read &= fread(ref_uv[0].data(), sizeof(Vec2), n_ref, f) == static_cast<size_t>(n_ref);
    if (n_cur) ok_read &= fread(cur_uv[0].data(), sizeof(Vec2), n_cur, f) == static_cast<size_t>(n_cur);
    if (!ok_read) {
        return 2;
    }
    Matcher matcher;
    matcher.options().kMaxValidDescriptorDistance = max_distance;
    matcher.options().kMaxValidPredictColDistance = max_col;
    matcher.options().kMaxValidPredictRowDistance = max_row;
    std::vector<int32_t> index;
    std::vector<Vec2> matched;
    std::vector<uint8_t> status;
    bool ok, ok2;
    if (nearby) {
        ok = matcher.NearbyMatch(ref, cur, ref_uv, cur_uv, index);
        ok2 = matcher.NearbyMatch(ref, cur, ref_uv, cur_uv, matched, status);
    } else {
        ok = matcher.ForceMatch(ref, cur, index);
        ok2 = matcher.ForceMatch(ref, cur, cur_uv, matched, status);
    }
    // the device path evaluates ComputeDistance only for its probe (6 pairs) and its post-check (<= 64 pairs) per call, two calls here;
    // the host loop would evaluate it n_ref x n_cur times
    printf("ok %d\nok2 %d\ndevice %d\n", ok ? 1 : 0, ok2 ? 1 : 0, g_distance_calls <= 400 ? 1 : 0);
    for (size_t i = 0; i < index.size(); ++i) {
        uint32_t ub = 0, vb = 0;
        if (i < matched.size()) {
            std::memcpy(&ub, &matched[i].x(), 4);
            std::memcpy(&vb, &matched[i].y(), 4);
        }
        printf("%d %d %08x %08x\n", index[i], i < status.size() ? status[i] : -1, ub, vb);
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 7) {
        return 2;
    }
    const bool nearby = std::string(argv[1]) == "nearby";
    const int dim = std::atoi(argv[2]);
    FILE *f = fopen(argv[6], "rb");
    if (!f) {
        return 2;
    }
    int rc;
    if (dim == 256) {
        rc = Run<SuperpointMatcher, SuperpointDescriptorType>(nearby, std::strtof(argv[3], nullptr), std::atoi(argv[4]), std::atoi(argv[5]), f);
    } else {
        rc = Run<DiskMatcher, DiskDescriptorType>(nearby, std::strtof(argv[3], nullptr), std::atoi(argv[4]), std::atoi(argv[5]), f);
    }
    fclose(f);
    return rc;
}
