#include "visualizor_2d.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace slam_visualizor {

namespace {

bool ReadFile(const std::string &file, std::vector<uint8_t> &bytes) {
    std::ifstream in(file, std::ios::binary);
    if (!in) {
        return false;
    }
    bytes.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
    return true;
}

uint32_t Be32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

int Paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool DecodePng(const std::vector<uint8_t> &f, GrayImage &image) {
    static const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (f.size() < 33 || std::memcmp(f.data(), kSig, 8) != 0) {
        return false;
    }
    uint32_t width = 0, height = 0;
    int bit_depth = 0, color_type = 0, interlace = 0;
    std::vector<uint8_t> idat;
    size_t pos = 8;
    while (pos + 12 <= f.size()) {
        const uint32_t len = Be32(&f[pos]);
        const uint8_t *type = &f[pos + 4];
        const uint8_t *data = &f[pos + 8];
        if (pos + 12 + len > f.size()) {
            return false;
        }
        if (std::memcmp(type, "IHDR", 4) == 0 && len >= 13) {
            width = Be32(data);
            height = Be32(data + 4);
            bit_depth = data[8];
            color_type = data[9];
            interlace = data[12];
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), data, data + len);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            break;
        }
        pos += 12 + len;
    }
    int channels = 0;
    switch (color_type) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: return false;  // palette images are not supported
    }
    if (width == 0 || height == 0 || bit_depth != 8 || interlace != 0) {
        return false;
    }
    const size_t stride = size_t(width) * channels;
    std::vector<uint8_t> raw((stride + 1) * height);
    uLongf raw_len = raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), idat.size()) != Z_OK || raw_len != raw.size()) {
        return false;
    }
    // undo the per-scanline filters in place
    std::vector<uint8_t> pixels(stride * height);
    for (uint32_t y = 0; y < height; ++y) {
        const uint8_t filter = raw[y * (stride + 1)];
        const uint8_t *src = &raw[y * (stride + 1) + 1];
        uint8_t *dst = &pixels[y * stride];
        const uint8_t *up = y > 0 ? &pixels[(y - 1) * stride] : nullptr;
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= size_t(channels) ? dst[x - channels] : 0;
            const int b = up ? up[x] : 0;
            const int c = (up && x >= size_t(channels)) ? up[x - channels] : 0;
            int v = src[x];
            switch (filter) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += Paeth(a, b, c); break;
                default: return false;
            }
            dst[x] = static_cast<uint8_t>(v);
        }
    }
    uint8_t *gray = static_cast<uint8_t *>(std::malloc(size_t(width) * height));
    if (gray == nullptr) {
        return false;
    }
    for (size_t i = 0; i < size_t(width) * height; ++i) {
        const uint8_t *p = &pixels[i * channels];
        gray[i] = channels <= 2 ? p[0] : static_cast<uint8_t>((299u * p[0] + 587u * p[1] + 114u * p[2] + 500u) / 1000u);
    }
    image.SetImage(gray, static_cast<int32_t>(height), static_cast<int32_t>(width), true);
    return true;
}

bool DecodePgm(const std::vector<uint8_t> &f, GrayImage &image) {
    if (f.size() < 7 || f[0] != 'P' || f[1] != '5') {
        return false;
    }
    size_t pos = 2;
    int vals[3];
    for (int k = 0; k < 3; ++k) {
        while (pos < f.size() && (f[pos] == ' ' || f[pos] == '\n' || f[pos] == '\r' || f[pos] == '\t' || f[pos] == '#')) {
            if (f[pos] == '#') {
                while (pos < f.size() && f[pos] != '\n') ++pos;
            } else {
                ++pos;
            }
        }
        int v = 0;
        while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9') {
            v = v * 10 + (f[pos++] - '0');
        }
        vals[k] = v;
    }
    ++pos;  // single whitespace after maxval
    const size_t n = size_t(vals[0]) * vals[1];
    if (vals[2] != 255 || n == 0 || pos + n > f.size()) {
        return false;
    }
    uint8_t *gray = static_cast<uint8_t *>(std::malloc(n));
    std::memcpy(gray, &f[pos], n);
    image.SetImage(gray, vals[1], vals[0], true);
    return true;
}

void DumpTracks(const std::string &title, const std::vector<Vec2> &ref_pixel_uv, const std::vector<Vec2> &cur_pixel_uv,
                const std::vector<uint8_t> &status) {
    const char *dir = std::getenv("FTK_VIS_DIR");
    if (dir == nullptr) {
        return;
    }
    std::string name = title;
    for (char &ch : name) {
        if (!((ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || (ch >= '0' && ch <= '9'))) {
            ch = '_';
        }
    }
    std::ofstream out(std::string(dir) + "/" + name + ".csv");
    out << "ref_u,ref_v,cur_u,cur_v,status\n";
    out.precision(9);
    for (size_t i = 0; i < ref_pixel_uv.size() && i < cur_pixel_uv.size(); ++i) {
        out << ref_pixel_uv[i].x() << ',' << ref_pixel_uv[i].y() << ',' << cur_pixel_uv[i].x() << ',' << cur_pixel_uv[i].y() << ','
            << (i < status.size() ? int(status[i]) : -1) << '\n';
    }
}

}  // namespace

bool Visualizor2D::LoadImage(const std::string &file, GrayImage &image) {
    std::vector<uint8_t> bytes;
    if (!ReadFile(file, bytes)) {
        return false;
    }
    return DecodePng(bytes, image) || DecodePgm(bytes, image);
}

bool Visualizor2D::SaveImage(const std::string &file, const GrayImage &image) {
    std::ofstream out(file, std::ios::binary);
    if (!out) {
        return false;
    }
    out << "P5\n" << image.cols() << ' ' << image.rows() << "\n255\n";
    out.write(reinterpret_cast<const char *>(image.data()), std::streamsize(image.rows()) * image.cols());
    return bool(out);
}

void Visualizor2D::ShowImageWithDetectedFeatures(const std::string &title, const GrayImage &image, const std::vector<Vec2> &pixel_uv) {
    (void)image;
    DumpTracks(title, pixel_uv, pixel_uv, std::vector<uint8_t>());
}

void Visualizor2D::ShowImageWithTrackedFeatures(const std::string &title, const GrayImage &cur_image, const std::vector<Vec2> &ref_pixel_uv,
                                                const std::vector<Vec2> &cur_pixel_uv, const std::vector<uint8_t> &status,
                                                uint8_t min_valid_status_value) {
    (void)cur_image;
    (void)min_valid_status_value;
    DumpTracks(title, ref_pixel_uv, cur_pixel_uv, status);
}

void Visualizor2D::ShowImageWithTrackedFeatures(const std::string &title, const GrayImage &ref_image, const GrayImage &cur_image,
                                                const std::vector<Vec2> &ref_pixel_uv, const std::vector<Vec2> &cur_pixel_uv,
                                                const std::vector<uint8_t> &status, uint8_t min_valid_status_value) {
    (void)ref_image;
    (void)cur_image;
    (void)min_valid_status_value;
    DumpTracks(title, ref_pixel_uv, cur_pixel_uv, status);
}

}  // namespace slam_visualizor
