// basic_type.h — minimal stand-in for Slam_Utility's basic_type.h (un-vendored; Eigen is not in this
// image).  Provides the fixed-size float vector / matrix types the feature_tracker API exposes:
// Vec2 is laid out as {float x, float y} (8 bytes) exactly like Eigen::Vector2f, so
// std::vector<Vec2> can be handed to the C ABI as a flat (u, v) array.
#ifndef _SLAM_UTILITY_BASIC_TYPE_H_
#define _SLAM_UTILITY_BASIC_TYPE_H_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <limits>
#include <string>
#include <vector>

template <int R, int C>
class FixedMat {
public:
    static constexpr int kRows = R;
    static constexpr int kCols = C;

    FixedMat() { std::memset(d_, 0, sizeof(d_)); }
    template <int RR = R, int CC = C, typename = typename std::enable_if<RR * CC == 2>::type>
    FixedMat(float a, float b) {
        d_[0] = a;
        d_[1] = b;
    }
    template <int RR = R, int CC = C, typename = typename std::enable_if<RR * CC == 3>::type>
    FixedMat(float a, float b, float c) {
        d_[0] = a;
        d_[1] = b;
        d_[2] = c;
    }

    static FixedMat Zero() { return FixedMat(); }
    static FixedMat Identity() {
        FixedMat m;
        for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0f;
        return m;
    }
    void setZero() { std::memset(d_, 0, sizeof(d_)); }
    void setIdentity() { *this = Identity(); }

    // column-major storage, as Eigen's default
    float &operator()(int r, int c) { return d_[c * R + r]; }
    const float &operator()(int r, int c) const { return d_[c * R + r]; }
    float &operator()(int i) { return d_[i]; }
    const float &operator()(int i) const { return d_[i]; }
    float &operator[](int i) { return d_[i]; }
    const float &operator[](int i) const { return d_[i]; }
    float &x() { return d_[0]; }
    float &y() { return d_[1]; }
    float &z() { return d_[2]; }
    const float &x() const { return d_[0]; }
    const float &y() const { return d_[1]; }
    const float &z() const { return d_[2]; }
    float *data() { return d_; }
    const float *data() const { return d_; }
    static constexpr int rows() { return R; }
    static constexpr int cols() { return C; }
    static constexpr int size() { return R * C; }

    FixedMat operator+(const FixedMat &o) const {
        FixedMat m;
        for (int i = 0; i < R * C; ++i) m.d_[i] = d_[i] + o.d_[i];
        return m;
    }
    FixedMat operator-(const FixedMat &o) const {
        FixedMat m;
        for (int i = 0; i < R * C; ++i) m.d_[i] = d_[i] - o.d_[i];
        return m;
    }
    FixedMat operator-() const {
        FixedMat m;
        for (int i = 0; i < R * C; ++i) m.d_[i] = -d_[i];
        return m;
    }
    FixedMat operator*(float s) const {
        FixedMat m;
        for (int i = 0; i < R * C; ++i) m.d_[i] = d_[i] * s;
        return m;
    }
    FixedMat operator/(float s) const {
        FixedMat m;
        for (int i = 0; i < R * C; ++i) m.d_[i] = d_[i] / s;
        return m;
    }
    FixedMat &operator+=(const FixedMat &o) { return *this = *this + o; }
    FixedMat &operator-=(const FixedMat &o) { return *this = *this - o; }
    FixedMat &operator*=(float s) { return *this = *this * s; }
    FixedMat &operator/=(float s) { return *this = *this / s; }
    bool operator==(const FixedMat &o) const { return std::memcmp(d_, o.d_, sizeof(d_)) == 0; }

    template <int K>
    FixedMat<R, K> operator*(const FixedMat<C, K> &o) const {
        FixedMat<R, K> m;
        for (int r = 0; r < R; ++r)
            for (int k = 0; k < K; ++k) {
                float s = 0.0f;
                for (int c = 0; c < C; ++c) s += (*this)(r, c) * o(c, k);
                m(r, k) = s;
            }
        return m;
    }
    FixedMat<C, R> transpose() const {
        FixedMat<C, R> m;
        for (int r = 0; r < R; ++r)
            for (int c = 0; c < C; ++c) m(c, r) = (*this)(r, c);
        return m;
    }
    // Small matrices: left-to-right.  Descriptor-sized vectors (SuperPoint 256, DISK 128): Eigen's
    // reduction for the reference's SSE2 build once the expression cost passes EIGEN_UNROLLING_LIMIT
    // (redux_impl<LinearVectorizedTraversal, NoUnrolling>, Core/Redux.h): two packet accumulators,
    // their sum, a trailing packet, predux (a0 + a2) + (a1 + a3), scalar tail.  Same definition as
    // oracle/oracle_float_matcher.c and the device's exact pass (csrc/float_matcher_kernels.hip).
    float dot(const FixedMat &o) const {
        constexpr int n = R * C;
        if constexpr (n * 3 + (n - 1) > 400) {
            constexpr int aligned_size = (n / 4) * 4, aligned_end2 = (n / 8) * 8;
            float p0[4], p1[4];
            for (int q = 0; q < 4; ++q) p0[q] = d_[q] * o.d_[q];
            for (int q = 0; q < 4; ++q) p1[q] = d_[4 + q] * o.d_[4 + q];
            for (int index = 8; index < aligned_end2; index += 8) {
                for (int q = 0; q < 4; ++q) {
                    p0[q] = p0[q] + d_[index + q] * o.d_[index + q];
                    p1[q] = p1[q] + d_[index + 4 + q] * o.d_[index + 4 + q];
                }
            }
            for (int q = 0; q < 4; ++q) p0[q] = p0[q] + p1[q];
            if (aligned_size > aligned_end2) {
                for (int q = 0; q < 4; ++q) p0[q] = p0[q] + d_[aligned_end2 + q] * o.d_[aligned_end2 + q];
            }
            float res = (p0[0] + p0[2]) + (p0[1] + p0[3]);
            for (int index = aligned_size; index < n; ++index) res = res + d_[index] * o.d_[index];
            return res;
        } else {
            float s = 0.0f;
            for (int i = 0; i < n; ++i) s += d_[i] * o.d_[i];
            return s;
        }
    }
    float squaredNorm() const { return dot(*this); }
    float norm() const { return std::sqrt(squaredNorm()); }

private:
    float d_[R * C];
};

template <int R, int C>
inline FixedMat<R, C> operator*(float s, const FixedMat<R, C> &m) {
    return m * s;
}

using Vec1 = FixedMat<1, 1>;
using Vec2 = FixedMat<2, 1>;
using Vec3 = FixedMat<3, 1>;
using Vec6 = FixedMat<6, 1>;
using Mat2 = FixedMat<2, 2>;
using Mat3 = FixedMat<3, 3>;
using Mat6 = FixedMat<6, 6>;
using Mat1x2 = FixedMat<1, 2>;
using Mat1x3 = FixedMat<1, 3>;
using Mat2x3 = FixedMat<2, 3>;

using Mat2x6 = FixedMat<2, 6>;

static_assert(sizeof(Vec2) == 2 * sizeof(float), "Vec2 must be a packed (u, v) pair");
static_assert(sizeof(Vec3) == 3 * sizeof(float), "Vec3 must be a packed (x, y, z) triple");

// Quat — stand-in for Eigen::Quaternionf as the direct-method callers use it: constructor order
// (w, x, y, z), Identity(), inverse(), normalized() / normalize(), q * q, q * v.  Arithmetic follows
// Eigen 3.3.7 for the reference's SSE2 build (oracle/oracle_direct_method.c states it operation by
// operation); the device kernel and the oracle use the same definitions.
class Quat {
public:
    Quat() : x_(0.0f), y_(0.0f), z_(0.0f), w_(1.0f) {}
    Quat(float w, float x, float y, float z) : x_(x), y_(y), z_(z), w_(w) {}
    static Quat Identity() { return Quat(1.0f, 0.0f, 0.0f, 0.0f); }
    float w() const { return w_; }
    float x() const { return x_; }
    float y() const { return y_; }
    float z() const { return z_; }
    float &w() { return w_; }
    float &x() { return x_; }
    float &y() { return y_; }
    float &z() { return z_; }
    float squaredNorm() const { return (x_ * x_ + z_ * z_) + (y_ * y_ + w_ * w_); }
    float norm() const { return std::sqrt(squaredNorm()); }
    Quat inverse() const {
        const float n2 = squaredNorm();
        if (n2 > 0.0f) {
            return Quat(w_ / n2, -x_ / n2, -y_ / n2, -z_ / n2);
        }
        return Quat(0.0f, 0.0f, 0.0f, 0.0f);
    }
    Quat normalized() const {
        const float z = squaredNorm();
        if (z > 0.0f) {
            const float n = std::sqrt(z);
            return Quat(w_ / n, x_ / n, y_ / n, z_ / n);
        }
        return *this;
    }
    void normalize() { *this = normalized(); }
    Quat operator*(const Quat &b) const {
        const Quat &a = *this;
        return Quat((a.w_ * b.w_ - a.x_ * b.x_) - (a.z_ * b.z_ + a.y_ * b.y_), (a.x_ * b.w_ - a.z_ * b.y_) + (a.y_ * b.z_ + a.w_ * b.x_),
                    (a.y_ * b.w_ - a.x_ * b.z_) + (a.z_ * b.x_ + a.w_ * b.y_), (a.z_ * b.w_ - a.y_ * b.x_) + (a.x_ * b.y_ + a.w_ * b.z_));
    }
    Vec3 operator*(const Vec3 &v) const {
        float ux = y_ * v.z() - z_ * v.y(), uy = z_ * v.x() - x_ * v.z(), uz = x_ * v.y() - y_ * v.x();
        ux += ux;
        uy += uy;
        uz += uz;
        const float cx = y_ * uz - z_ * uy, cy = z_ * ux - x_ * uz, cz = x_ * uy - y_ * ux;
        return Vec3((v.x() + w_ * ux) + cx, (v.y() + w_ * uy) + cy, (v.z() + w_ * uz) + cz);
    }

private:
    float x_, y_, z_, w_;  // Eigen's coefficient order
};

constexpr int32_t kMaxInt32 = std::numeric_limits<int32_t>::max();
constexpr float kPai = 3.14159265358979323846f;
constexpr float kZeroFloat = 1e-6f;  // Slam_Utility slam_basic_math.h (un-vendored): this repo's normative value

#endif  // _SLAM_UTILITY_BASIC_TYPE_H_
