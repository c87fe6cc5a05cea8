// Minimal stand-ins for the OpenCV types the reference's NMS2() touches (swarm_loop/src/superpoint_tensorrt.cpp:237-310), so that the
// reference's own function text compiles VERBATIM into oracle/_ref/libref_nms2.so (OpenCV 3.4 itself is not available here).  Test
// infrastructure only.  Also what getKeyPoints / computeDescriptors (:164-230) need on top (sp_post_wrap.cpp): cv::Point, `mat > float`,
// findNonZero (row-major, as OpenCV's), a const at<T>() and the data pointer.  cv::Mat here is what NMS2 needs of it: a zero-initialisable contiguous row-major plane with at<T>(row, col) and
// setTo().  Like the real cv::Mat it does NOT bounds-check: a column index outside [0, cols) lands in the adjacent row (contiguous memory);
// a row index outside [0, rows) would be heap UB in the reference -- the shim keeps GUARD zero rows above and below so that such reads see
// zeros instead of crashing.  The pin test therefore only uses maps whose candidates keep 4 pixels away from the frame.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#define CV_8UC1 0
#define CV_16UC1 2
#define CV_32FC1 5
#define CV_32F 5

namespace cv {
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float a, float b) : x(a), y(b) {} };
struct Point { int x, y; Point() : x(0), y(0) {} Point(int a, int b) : x(a), y(b) {} };
struct Size { int width, height; Size(int w, int h) : width(w), height(h) {} };
class Mat {
public:
    enum { GUARD = 8 };
    Mat() : rows(0), cols(0), esz(0) {}
    Mat(Size s, int type) { create(s.height, s.width, type); }
    Mat(int r, int c, int type) { create(r, c, type); }
    void create(int r, int c, int type) {
        rows = r; cols = c; esz = type == CV_8UC1 ? 1 : (type == CV_16UC1 ? 2 : 4);
        buf.assign((size_t)(rows + 2 * GUARD) * cols * esz, 0xCD);          // uninitialised, as cv::Mat(Size, type) is
        std::memset(buf.data(), 0, (size_t)GUARD * cols * esz);
        std::memset(buf.data() + (size_t)(GUARD + rows) * cols * esz, 0, (size_t)GUARD * cols * esz);
        data = buf.data() + (size_t)GUARD * cols * esz;
    }
    Mat(const Mat& o) : rows(o.rows), cols(o.cols), esz(o.esz), buf(o.buf) { data = buf.data() + (size_t)GUARD * cols * esz; }
    Mat& operator=(const Mat& o) { rows = o.rows; cols = o.cols; esz = o.esz; buf = o.buf; data = buf.data() + (size_t)GUARD * cols * esz; return *this; }
    void setTo(int v) { std::memset(buf.data() + (size_t)GUARD * cols * esz, v, (size_t)rows * cols * esz); }
    template <typename T> T& at(int r, int c) { return *reinterpret_cast<T*>(buf.data() + ((size_t)(r + GUARD) * cols + c) * esz); }
    template <typename T> const T& at(int r, int c) const { return *reinterpret_cast<const T*>(buf.data() + ((size_t)(r + GUARD) * cols + c) * esz); }
    int rows, cols, esz;
    std::vector<unsigned char> buf;
    unsigned char* data = nullptr;           // element (0, 0)
};
// `prob > threshold` on a CV_32F plane: an 8-bit mask, 255 where true
inline Mat operator>(const Mat& m, float t) {
    Mat o(m.rows, m.cols, CV_8UC1);
    for (int r = 0; r < m.rows; ++r)
        for (int c = 0; c < m.cols; ++c) o.at<unsigned char>(r, c) = m.at<float>(r, c) > t ? 255 : 0;
    return o;
}
// cv::findNonZero: the non-zero pixels in row-major order
inline void findNonZero(const Mat& m, std::vector<Point>& out) {
    out.clear();
    for (int r = 0; r < m.rows; ++r)
        for (int c = 0; c < m.cols; ++c) if (m.at<unsigned char>(r, c)) out.push_back(Point(c, r));
}
}  // namespace cv
