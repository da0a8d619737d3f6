// cv_shim.h — the two OpenCV types MultiH's public interface uses
// (cv::Point2d, cv::Mat as a small dense CV_64F matrix), provided only when
// OpenCV headers are absent.  With OpenCV installed, define MULTIH_USE_OPENCV and
// the real types are used, so the class drops into the reference harness
// (M/main.cpp:262-296) unchanged.
#pragma once

#if defined(MULTIH_USE_OPENCV)
#include <opencv2/core.hpp>
#else
#include <cstddef>
#include <memory>
#include <vector>

#ifndef CV_64F
#define CV_64F 6
#endif

namespace cv {

struct Point2d {
    double x = 0.0, y = 0.0;
    Point2d() = default;
    Point2d(double x_, double y_) : x(x_), y(y_) {}
};

// Dense row-major double matrix with shared storage (copy = header copy, like cv::Mat).
class Mat {
public:
    int rows = 0, cols = 0;
    unsigned char* data = nullptr;

    Mat() = default;
    Mat(int r, int c, int /*type*/) { create(r, c); }
    Mat(int r, int c, int /*type*/, const double* src) { create(r, c); for (int i = 0; i < r * c; ++i) ptr()[i] = src[i]; }

    void create(int r, int c)
    {
        rows = r; cols = c;
        store_ = std::shared_ptr<double>(new double[(size_t)r * c](), std::default_delete<double[]>());
        data = reinterpret_cast<unsigned char*>(store_.get());
    }
    bool empty() const { return data == nullptr || rows * cols == 0; }
    Mat clone() const { Mat m; if (!empty()) { m.create(rows, cols); for (int i = 0; i < rows * cols; ++i) m.ptr()[i] = ptr()[i]; } return m; }

    template <typename T> T& at(int r, int c) { return reinterpret_cast<T*>(data)[(size_t)r * cols + c]; }
    template <typename T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(data)[(size_t)r * cols + c]; }
    template <typename T> T& at(int i) { return reinterpret_cast<T*>(data)[i]; }
    template <typename T> const T& at(int i) const { return reinterpret_cast<const T*>(data)[i]; }

    double* ptr() { return reinterpret_cast<double*>(data); }
    const double* ptr() const { return reinterpret_cast<const double*>(data); }

private:
    std::shared_ptr<double> store_;
};

} // namespace cv
#endif
