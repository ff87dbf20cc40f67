/*
 * TEST INFRASTRUCTURE (oracle/_ref), NOT product code.
 *
 * A minimal stand-in for the parts of OpenCV 3 that the reference's MATCH half uses only as a
 * buffer type (cv::Mat, Size, Point, Rect, Ptr, CV_Assert, checkHardwareSupport ...), so that
 * /root/reference/linemodLevelup/linemodLevelup.h and the match functions of linemodLevelup.cpp
 * (LL.cpp:1022-1658 spread / computeResponseMaps / linearize / similarity*, :1692-1941
 * Detector::match / matchClass) compile HERE, unmodified, from where they lie.  See oracle/Makefile
 * (target _ref) and oracle/ref_harness.cpp.  Nothing in this header restates reference logic: it
 * only provides storage and the handful of element-wise operations those lines call on a Mat
 * (zeros, create, convertTo 8U/16U -> 16U, operator+= on 16U, add).
 *
 * Differences from OpenCV that matter here:
 *   - every allocation is zero-filled and followed by a zero tail (rows*cols + 4096 bytes): the
 *     reference reads a little past a linear-memory Mat for features at x==width / y==height
 *     (SURVEY A7); inside one label that is the next phase row (same as OpenCV), past the last
 *     phase it is heap garbage in OpenCV and zeros here.  Comparisons stay inside the former.
 *   - CV_SSE2 / CV_SSE3 / CV_SSSE3 follow the compiler's __SSE2__ / __SSE3__ / __SSSE3__ exactly
 *     like OpenCV's cvdef.h, so `-O3 -Wall` (the reference's flags) selects the SSE2 paths and
 *     `-mssse3` the pshufb path.
 */
#ifndef REF_SHIM_OPENCV_CORE_HPP
#define REF_SHIM_OPENCV_CORE_HPP
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#if defined __SSE2__
#include <emmintrin.h>
#define CV_SSE2 1
#endif
#if defined __SSE3__
#include <pmmintrin.h>
#define CV_SSE3 1
#endif
#if defined __SSSE3__
#include <tmmintrin.h>
#define CV_SSSE3 1
#endif

#define CV_DECL_ALIGNED(x) __attribute__((aligned(x)))

typedef unsigned char uchar;
typedef unsigned short ushort;

#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_MAT_DEPTH(t) ((t) & 7)
#define CV_MAT_CN(t) ((((t) >> 3) & 511) + 1)
#define CV_MAKETYPE(d, cn) (CV_MAT_DEPTH(d) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)

enum { CV_CPU_SSE2 = 3, CV_CPU_SSE3 = 4, CV_CPU_SSSE3 = 5 };

namespace cv {
using ::uchar;
using ::ushort;
typedef std::string String;

enum { CPU_SSE2 = 3, CPU_SSE3 = 4, CPU_SSSE3 = 5 };
inline bool checkHardwareSupport(int) { return true; }

class Exception : public std::runtime_error {
public:
    explicit Exception(const std::string& m) : std::runtime_error(m) {}
};
#define CV_Assert(expr) do { if (!(expr)) throw cv::Exception(std::string("CV_Assert failed: ") + #expr); } while (0)
#define CV_DbgAssert(expr) ((void)0)
#define CV_Error(code, msg) throw cv::Exception(std::string(msg))
#define CV_StsBadArg (-5)

template <typename T> struct Size_ {
    T width, height;
    Size_() : width(0), height(0) {}
    Size_(T w, T h) : width(w), height(h) {}
    bool operator==(const Size_& o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size_& o) const { return !(*this == o); }
    T area() const { return width * height; }
};
typedef Size_<int> Size;
template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T _x, T _y) : x(_x), y(_y) {}
};
typedef Point_<int> Point;
template <typename T> struct Rect_ {
    T x, y, width, height;
    Rect_() : x(0), y(0), width(0), height(0) {}
    Rect_(T _x, T _y, T w, T h) : x(_x), y(_y), width(w), height(h) {}
};
typedef Rect_<int> Rect;

template <typename T> using Ptr = std::shared_ptr<T>;
template <typename T, typename... A> Ptr<T> makePtr(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }

class FileNode;
class FileStorage;

inline size_t shim_elem_size(int type)
{
    static const size_t depth_size[8] = {1, 1, 2, 2, 4, 4, 8, 0};
    return depth_size[CV_MAT_DEPTH(type)] * CV_MAT_CN(type);
}

class Mat {
public:
    int rows, cols;
    uchar* data;
    size_t step;    // bytes per row

    Mat() : rows(0), cols(0), data(nullptr), step(0), type_(0) {}
    Mat(int r, int c, int type) : Mat() { create(r, c, type); }
    Mat(Size s, int type) : Mat() { create(s.height, s.width, type); }
    // wrap caller memory without taking ownership (same as cv::Mat(rows, cols, type, data, step))
    Mat(int r, int c, int type, void* d, size_t st = 0)
        : rows(r), cols(c), data(static_cast<uchar*>(d)), step(st ? st : c * shim_elem_size(type)), type_(type) {}

    void create(int r, int c, int type)
    {
        if (data && r == rows && c == cols && type == type_ && buf_) return;   // cv::Mat::create keeps a matching buffer
        rows = r; cols = c; type_ = type; step = size_t(c) * shim_elem_size(type);
        size_t bytes = step * size_t(r);
        size_t alloc = bytes * 2 + 4096;
        void* p = nullptr;
        if (posix_memalign(&p, 64, alloc) != 0) throw std::bad_alloc();
        std::memset(p, 0, alloc);
        buf_.reset(static_cast<uchar*>(p), std::free);
        data = buf_.get();
    }
    void create(Size s, int type) { create(s.height, s.width, type); }

    static Mat zeros(int r, int c, int type)
    {
        Mat m;
        m.rows = m.cols = 0; m.data = nullptr;
        m.buf_.reset();
        m.create(r, c, type);     // fresh allocation is zero-filled
        return m;
    }
    static Mat zeros(Size s, int type) { return zeros(s.height, s.width, type); }

    int type() const { return type_; }
    int depth() const { return CV_MAT_DEPTH(type_); }
    int channels() const { return CV_MAT_CN(type_); }
    size_t elemSize() const { return shim_elem_size(type_); }
    size_t elemSize1() const { return shim_elem_size(CV_MAT_DEPTH(type_)); }
    size_t step1() const { return step / elemSize1(); }
    size_t total() const { return size_t(rows) * size_t(cols); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    Size size() const { return Size(cols, rows); }
    bool isContinuous() const { return step == size_t(cols) * elemSize(); }

    uchar* ptr(int r = 0) { return data + step * size_t(r); }
    const uchar* ptr(int r = 0) const { return data + step * size_t(r); }
    template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + step * size_t(r)); }
    template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + step * size_t(r)); }
    template <typename T> T& at(int r, int c) { return reinterpret_cast<T*>(data + step * size_t(r))[c]; }
    template <typename T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(data + step * size_t(r))[c]; }

    Mat clone() const
    {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; ++r) std::memcpy(m.ptr(r), ptr(r), size_t(cols) * elemSize());
        return m;
    }

    // the two conversions the match path performs: 8U -> 16U and 16U -> 16U (addSimilarities*)
    void convertTo(Mat& dst, int rtype) const
    {
        CV_Assert(CV_MAT_DEPTH(rtype) == CV_16U && channels() == 1 && (depth() == CV_8U || depth() == CV_16U));
        Mat out(rows, cols, CV_16U);
        for (int r = 0; r < rows; ++r) {
            ushort* d = out.ptr<ushort>(r);
            if (depth() == CV_8U) { const uchar* s = ptr(r); for (int c = 0; c < cols; ++c) d[c] = s[c]; }
            else { const ushort* s = ptr<ushort>(r); for (int c = 0; c < cols; ++c) d[c] = s[c]; }
        }
        dst = out;
    }

    // dst += src on 16U with cv::add's saturation (addSimilarities, LL.cpp:1442-1446)
    Mat& operator+=(const Mat& o)
    {
        CV_Assert(type_ == CV_16U && o.type_ == CV_16U && rows == o.rows && cols == o.cols);
        for (int r = 0; r < rows; ++r) {
            ushort* d = ptr<ushort>(r);
            const ushort* s = o.ptr<ushort>(r);
            for (int c = 0; c < cols; ++c) { unsigned v = unsigned(d[c]) + s[c]; d[c] = ushort(v > 65535u ? 65535u : v); }
        }
        return *this;
    }

private:
    int type_;
    std::shared_ptr<uchar> buf_;
};

struct NoArrayTag {};
inline NoArrayTag noArray() { return NoArrayTag(); }
// add(16U, 8U or 16U) -> 16U, only reached with more than two modalities (LL.cpp:1654-1655)
inline void add(const Mat& a, const Mat& b, Mat& dst, NoArrayTag, int dtype)
{
    CV_Assert(dtype == CV_16U && a.type() == CV_16U && a.size() == b.size());
    Mat out(a.rows, a.cols, CV_16U);
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < a.cols; ++c) {
            unsigned v = unsigned(a.at<ushort>(r, c)) + (b.depth() == CV_8U ? unsigned(b.at<uchar>(r, c)) : unsigned(b.at<ushort>(r, c)));
            out.at<ushort>(r, c) = ushort(v > 65535u ? 65535u : v);
        }
    dst = out;
}
}  // namespace cv
#endif
