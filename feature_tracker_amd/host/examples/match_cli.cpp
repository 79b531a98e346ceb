// match_cli — drives DescriptorMatcher<BriefType> through a subclass written exactly like the
// reference's BriefMatcher (test/test_descriptor_matcher_brief.cpp:27-46).
//
//   match_cli <force|nearby> <max_distance> <max_col> <max_row> <descriptors.txt> [nearmiss|nooffload]
//
// nearmiss : the virtual distance is the Hamming distance EXCEPT that it answers 1000 below 25 differing bits — it agrees
//            with the built-in distance on random pairs (the call-time probe) and differs exactly on the pairs a matcher
//            returns; the library must notice (post-check) and honour the virtual on the host.
// nooffload: Options::kAllowDeviceOffload = false (the host loop over the virtual, whatever the distance is).
//
// descriptors.txt: first line "n_ref n_cur n_bits"; then n_ref lines "bits [u v]" and n_cur lines
// "bits [u v]" where bits is a 0/1 string.  Output: "ok <0|1>" then one index per ref descriptor and
// the pixel-returning overload's "status u_bits v_bits".
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "descriptor_brief.h"
#include "descriptor_matcher.h"

class BriefMatcher: public feature_tracker::DescriptorMatcher<feature_detector::BriefType> {
public:
    bool near_miss = false;
    virtual float ComputeDistance(const feature_detector::BriefType &descriptor_ref, const feature_detector::BriefType &descriptor_cur) override {
        if (descriptor_ref.empty() || descriptor_cur.empty()) {
            return kMaxInt32;
        }
        int32_t distance = 0;
        for (uint32_t i = 0; i < descriptor_ref.size(); ++i) {
            if (descriptor_ref[i] != descriptor_cur[i]) {
                ++distance;
            }
        }
        if (near_miss && distance < 25) {
            return 1000.0f;
        }
        return static_cast<float>(distance);
    }
};

int main(int argc, char **argv) {
    if (argc < 6) {
        return 2;
    }
    const bool nearby = std::string(argv[1]) == "nearby";
    BriefMatcher matcher;
    matcher.options().kMaxValidDescriptorDistance = std::strtof(argv[2], nullptr);
    matcher.options().kMaxValidPredictColDistance = std::atoi(argv[3]);
    matcher.options().kMaxValidPredictRowDistance = std::atoi(argv[4]);
    matcher.near_miss = argc >= 7 && std::string(argv[6]) == "nearmiss";
    matcher.options().kAllowDeviceOffload = !(argc >= 7 && std::string(argv[6]) == "nooffload");
    std::ifstream in(argv[5]);
    size_t n_ref = 0, n_cur = 0, n_bits = 0;
    in >> n_ref >> n_cur >> n_bits;
    std::vector<feature_detector::BriefType> ref(n_ref), cur(n_cur);
    std::vector<Vec2> ref_uv(n_ref), cur_uv(n_cur);
    auto read = [&](std::vector<feature_detector::BriefType> &d, std::vector<Vec2> &uv) {
        for (size_t i = 0; i < d.size(); ++i) {
            std::string bits, su, sv;
            in >> bits >> su >> sv;  // hex floats: parse with strtof (iostream does not)
            const float u = std::strtof(su.c_str(), nullptr), v = std::strtof(sv.c_str(), nullptr);
            if (bits == "-") bits.clear();
            d[i].resize(bits.size());
            for (size_t b = 0; b < bits.size(); ++b) d[i][b] = bits[b] == '1';
            uv[i] = Vec2(u, v);
        }
    };
    read(ref, ref_uv);
    read(cur, cur_uv);

    std::vector<int32_t> index;
    const bool ok = nearby ? matcher.NearbyMatch(ref, cur, ref_uv, cur_uv, index) : matcher.ForceMatch(ref, cur, index);
    std::printf("ok %d\n", ok ? 1 : 0);
    std::vector<Vec2> matched;
    std::vector<uint8_t> status;
    const bool ok2 = nearby ? matcher.NearbyMatch(ref, cur, ref_uv, cur_uv, matched, status) : matcher.ForceMatch(ref, cur, cur_uv, matched, status);
    std::printf("ok2 %d\n", ok2 ? 1 : 0);
    for (size_t i = 0; i < index.size(); ++i) {
        uint32_t ub = 0, vb = 0;
        int st = -1;
        if (ok2 && i < matched.size()) {
            std::memcpy(&ub, &matched[i].x(), 4);
            std::memcpy(&vb, &matched[i].y(), 4);
            st = status[i];
        }
        std::printf("%d %d %08x %08x\n", index[i], st, ub, vb);
    }
    return 0;
}
