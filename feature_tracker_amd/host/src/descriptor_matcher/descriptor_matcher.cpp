// descriptor_matcher.cpp — device side of DescriptorMatcher<T>.  Float descriptors go to
// ftk_cosine_match as they are (a std::vector of packed float vectors is one n x dim array).
// DescriptorMatcher<std::vector<bool>>: packs the per-bit
// BRIEF containers into 32-bit words and runs the all-pairs Hamming scan on the MI355X
// (replaces the double loops of descriptor_matcher.h:55-79 / :90-124 of the reference for the
// distance of test/test_descriptor_matcher_brief.cpp:33-45).
#include "descriptor_matcher.h"

#include <string>

#include "bit_words.h"
#include "device_runtime.h"
#include "ftk.h"
#include "slam_log_reporter.h"

namespace feature_tracker {
namespace device {

void WarmUpMatcher(bool hamming) { WarmUp(hamming ? FTK_WARM_HAMMING : FTK_WARM_COSINE); }

float HammingDistance(const std::vector<bool> &a, const std::vector<bool> &b) {
    if (a.empty() || b.empty()) {
        return static_cast<float>(kMaxInt32);
    }
    int32_t distance = 0;
#if FTK_BIT_WORDS_FAST
    if (a.size() == b.size()) {  // whole words of the two containers: xor + popcount (the padding bits of the last word masked out)
        const unsigned long *pa = a.begin()._M_p, *pb = b.begin()._M_p;
        constexpr size_t kWordBits = 8 * sizeof(unsigned long);
        const size_t full = a.size() / kWordBits, tail = a.size() % kWordBits;
        for (size_t w = 0; w < full; ++w) {
            distance += __builtin_popcountl(pa[w] ^ pb[w]);
        }
        if (tail) {
            distance += __builtin_popcountl((pa[full] ^ pb[full]) & ((1ul << tail) - 1ul));
        }
        return static_cast<float>(distance);
    }
#endif
    for (size_t i = 0; i < a.size(); ++i) {
        distance += (a[i] != b[i]) ? 1 : 0;
    }
    return static_cast<float>(distance);
}

namespace {
void PackBits(const std::vector<std::vector<bool>> &descriptors, int32_t n_words, std::vector<uint32_t> &words) {
    words.resize(descriptors.size() * static_cast<size_t>(n_words));
    for (size_t i = 0; i < descriptors.size(); ++i) {
        // a descriptor longer than the first one's n_words cannot occur here (the caller's length check), a shorter one is zero padded
        bit_words::Pack(descriptors[i], &words[i * n_words], static_cast<size_t>(n_words));
    }
}
}  // namespace

bool HammingMatch(const std::vector<std::vector<bool>> &descriptors_ref, const std::vector<std::vector<bool>> &descriptors_cur,
                  const std::vector<Vec2> *pixel_uv_pred_in_cur, const std::vector<Vec2> *pixel_uv_cur, float max_distance,
                  int32_t max_col_distance, int32_t max_row_distance, std::vector<int32_t> &index_pairs_in_cur) {
    if (descriptors_ref.empty()) {
        return true;  // nothing to match; the caller already handled an empty `cur`
    }
    std::string error;
    ftk_context *ctx = SharedContext(&error);
    if (ctx == nullptr) {
        ReportError("[DescriptorMatcher] " << error);
        return false;
    }
    const int32_t n_bits = descriptors_ref.empty() ? 0 : static_cast<int32_t>(descriptors_ref[0].size());
    const int32_t n_words = n_bits == 0 ? 1 : (n_bits + 31) / 32;
    std::vector<uint32_t> ref_words, cur_words;
    PackBits(descriptors_ref, n_words, ref_words);
    PackBits(descriptors_cur, n_words, cur_words);
    int ok = 0;
    const int rc = ftk_hamming_match(ctx, ref_words.data(), static_cast<int32_t>(descriptors_ref.size()), cur_words.data(),
                                     static_cast<int32_t>(descriptors_cur.size()), n_words, n_bits, max_distance,
                                     pixel_uv_pred_in_cur ? (*pixel_uv_pred_in_cur)[0].data() : nullptr,
                                     pixel_uv_cur ? (*pixel_uv_cur)[0].data() : nullptr, max_col_distance, max_row_distance,
                                     index_pairs_in_cur.data(), &ok);
    if (rc != FTK_OK) {
        ReportError("[DescriptorMatcher] " << ftk_last_error(ctx));
        return false;
    }
    return ok != 0;
}

bool CosineMatch(const float *descriptors_ref, int32_t n_ref, const float *descriptors_cur, int32_t n_cur, int32_t dim,
                 const std::vector<Vec2> *pixel_uv_pred_in_cur, const std::vector<Vec2> *pixel_uv_cur, float max_distance,
                 int32_t max_col_distance, int32_t max_row_distance, std::vector<int32_t> &index_pairs_in_cur) {
    std::string error;
    ftk_context *ctx = SharedContext(&error);
    if (ctx == nullptr) {
        ReportError("[DescriptorMatcher] " << error);
        return false;
    }
    int ok = 0;
    const int rc = ftk_cosine_match(ctx, descriptors_ref, n_ref, descriptors_cur, n_cur, dim, max_distance,
                                    pixel_uv_pred_in_cur ? (*pixel_uv_pred_in_cur)[0].data() : nullptr,
                                    pixel_uv_cur ? (*pixel_uv_cur)[0].data() : nullptr, max_col_distance, max_row_distance,
                                    index_pairs_in_cur.data(), &ok);
    if (rc != FTK_OK) {
        ReportError("[DescriptorMatcher] " << ftk_last_error(ctx));
        return false;
    }
    return ok != 0;
}

}  // namespace device
}  // namespace feature_tracker
