// cv_shim.h — the OpenCV types MultiH's public interface uses (cv::Point2d, cv::Mat, and for
// DrawClusters cv::Scalar / cv::circle), provided only when OpenCV headers are absent.  With OpenCV
// installed, define MULTIH_USE_OPENCV and the real types are used, so the class drops into the
// reference harness (M/main.cpp:262-296) unchanged.
//
// The stand-in keeps OpenCV's OWNERSHIP rules, because the host code must be correct under the real
// library: Mat(rows, cols, type) allocates shared storage (a copy of the header shares it),
// Mat(rows, cols, type, void* data) is a NON-owning header over the caller's memory exactly like
// cv::Mat's (and, like it, does not accept a pointer to const), clone() makes a deep copy.  Code
// that wants a matrix of its own from a raw array allocates and copies explicitly.
#pragma once

#if defined(MULTIH_USE_OPENCV)
#include <opencv2/core.hpp>
#include <opencv2/imgproc.hpp>
#else
#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

#ifndef CV_64F
#define CV_8UC3 16
#define CV_64F 6
#endif

namespace cv {

struct Point2d {
    double x = 0.0, y = 0.0;
    Point2d() = default;
    Point2d(double x_, double y_) : x(x_), y(y_) {}
};

struct Scalar {
    double val[4] = { 0, 0, 0, 0 };
    Scalar() = default;
    Scalar(double a, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
};

// Dense row-major matrix, CV_64F (doubles) or CV_8UC3 (three bytes per pixel).
class Mat {
public:
    int rows = 0, cols = 0;
    unsigned char* data = nullptr;

    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    // NON-owning header over `external`: the caller keeps the memory alive (cv::Mat's rule).
    Mat(int r, int c, int type, void* external) : rows(r), cols(c), data(static_cast<unsigned char*>(external)), type_(type) {}

    void create(int r, int c, int type = CV_64F)
    {
        rows = r; cols = c; type_ = type;
        const size_t bytes = (size_t)r * c * elemSize();
        store_ = std::shared_ptr<unsigned char>(new unsigned char[bytes ? bytes : 1](), std::default_delete<unsigned char[]>());
        data = store_.get();
    }
    int type() const { return type_; }
    size_t elemSize() const { return type_ == CV_8UC3 ? 3 : sizeof(double); }
    bool empty() const { return data == nullptr || rows * cols == 0; }
    Mat clone() const
    {
        Mat m;
        if (!empty()) { m.create(rows, cols, type_); std::memcpy(m.data, data, (size_t)rows * cols * elemSize()); }
        return m;
    }

    template <typename T> T& at(int r, int c) { return reinterpret_cast<T*>(data)[(size_t)r * cols + c]; }
    template <typename T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(data)[(size_t)r * cols + c]; }
    template <typename T> T& at(int i) { return reinterpret_cast<T*>(data)[i]; }
    template <typename T> const T& at(int i) const { return reinterpret_cast<const T*>(data)[i]; }

private:
    std::shared_ptr<unsigned char> store_;
    int type_ = CV_64F;
};

// Filled disc (thickness < 0) or one-pixel ring of the given radius on a CV_8UC3 image; enough for DrawClusters.
inline void circle(Mat& img, const Point2d& centre, int radius, const Scalar& colour, int thickness = 1)
{
    if (img.empty() || img.type() != CV_8UC3) return;
    const int cx = (int)(centre.x + (centre.x >= 0 ? 0.5 : -0.5)), cy = (int)(centre.y + (centre.y >= 0 ? 0.5 : -0.5));
    for (int y = cy - radius; y <= cy + radius; ++y)
        for (int x = cx - radius; x <= cx + radius; ++x) {
            if (x < 0 || y < 0 || x >= img.cols || y >= img.rows) continue;
            const int d2 = (x - cx) * (x - cx) + (y - cy) * (y - cy);
            const bool on = thickness < 0 ? d2 <= radius * radius : (d2 <= radius * radius && d2 > (radius - 1) * (radius - 1));
            if (!on) continue;
            unsigned char* p = img.data + 3 * ((size_t)y * img.cols + x);
            for (int ch = 0; ch < 3; ++ch) p[ch] = (unsigned char)colour.val[ch];
        }
}

} // namespace cv
#endif
