#include "feature_point_harris_detector.h"

#include <algorithm>
#include <cmath>

namespace feature_detector {

bool FeaturePointHarrisDetector::DetectGoodFeatures(const GrayImage &image, const uint32_t needed_feature_num, std::vector<Vec2> &features) {
    const int32_t rows = image.rows(), cols = image.cols();
    if (image.data() == nullptr || rows < 16 || cols < 16) {
        return false;
    }
    const int32_t h = options_.kHalfPatchSize;
    const size_t n = size_t(rows) * cols;
    std::vector<float> ixx(n, 0.0f), iyy(n, 0.0f), ixy(n, 0.0f);
    for (int32_t r = 1; r < rows - 1; ++r) {
        for (int32_t c = 1; c < cols - 1; ++c) {
            auto p = [&](int32_t dr, int32_t dc) { return float(image.GetPixelValueNoCheck(r + dr, c + dc)); };
            const float gx = (p(-1, 1) + 2.0f * p(0, 1) + p(1, 1)) - (p(-1, -1) + 2.0f * p(0, -1) + p(1, -1));
            const float gy = (p(1, -1) + 2.0f * p(1, 0) + p(1, 1)) - (p(-1, -1) + 2.0f * p(-1, 0) + p(-1, 1));
            const size_t i = size_t(r) * cols + c;
            ixx[i] = gx * gx;
            iyy[i] = gy * gy;
            ixy[i] = gx * gy;
        }
    }
    struct Candidate {
        float response;
        int32_t row, col;
    };
    std::vector<Candidate> candidates;
    const int32_t border = h + 1 + std::max(8, h);
    for (int32_t r = border; r < rows - border; ++r) {
        for (int32_t c = border; c < cols - border; ++c) {
            float a = 0.0f, b = 0.0f, d = 0.0f;
            for (int32_t dr = -h; dr <= h; ++dr) {
                const size_t base = size_t(r + dr) * cols + c;
                for (int32_t dc = -h; dc <= h; ++dc) {
                    a += ixx[base + dc];
                    b += ixy[base + dc];
                    d += iyy[base + dc];
                }
            }
            const float response = ((a * d - b * b) - options_.kAlpha * (a + d) * (a + d)) * 1e-6f;
            if (response > options_.kMinValidResponse) {
                candidates.push_back({response, r, c});
            }
        }
    }
    std::sort(candidates.begin(), candidates.end(), [](const Candidate &x, const Candidate &y) {
        return x.response != y.response ? x.response > y.response : (x.row != y.row ? x.row < y.row : x.col < y.col);
    });
    // greedy non-maximum suppression on a coarse occupancy grid
    const int32_t cell = std::max(1, options_.kMinFeatureDistance);
    const int32_t grid_rows = rows / cell + 1, grid_cols = cols / cell + 1;
    std::vector<std::vector<Vec2>> grid(size_t(grid_rows) * grid_cols);
    for (const Vec2 &f : features) {  // already present features block their neighbourhood too
        grid[size_t(int32_t(f.y()) / cell) * grid_cols + int32_t(f.x()) / cell].push_back(f);
    }
    const float min_dist2 = float(cell) * float(cell);
    for (const Candidate &cand : candidates) {
        if (features.size() >= needed_feature_num) {
            break;
        }
        const int32_t gr = cand.row / cell, gc = cand.col / cell;
        bool free_spot = true;
        for (int32_t dr = -1; dr <= 1 && free_spot; ++dr) {
            for (int32_t dc = -1; dc <= 1 && free_spot; ++dc) {
                const int32_t rr = gr + dr, cc = gc + dc;
                if (rr < 0 || cc < 0 || rr >= grid_rows || cc >= grid_cols) {
                    continue;
                }
                for (const Vec2 &f : grid[size_t(rr) * grid_cols + cc]) {
                    const float du = f.x() - float(cand.col), dv = f.y() - float(cand.row);
                    if (du * du + dv * dv < min_dist2) {
                        free_spot = false;
                        break;
                    }
                }
            }
        }
        if (free_spot) {
            const Vec2 f(float(cand.col), float(cand.row));
            features.push_back(f);
            grid[size_t(gr) * grid_cols + gc].push_back(f);
        }
    }
    return true;
}

}  // namespace feature_detector
