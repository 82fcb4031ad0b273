// Minimal stand-in for <opencv2/core/core.hpp>, used ONLY when real OpenCV is absent
// (it is on this image) so that the reference's mainSift.cpp / geomFuncs.cpp can be
// compiled unchanged against libcudasift.so.  It implements exactly the surface those
// two files touch (mainSift.cpp:34-40,86; geomFuncs.cpp:17-55): a dense 2-D single-channel
// cv::Mat of 8U / 32F / 64F, convertTo, at<T>, Scalar fill, += , Mat*scalar, and
// cv::solve(..., DECOMP_CHOLESKY) for small symmetric positive-definite systems.
// Not OpenCV code; written from the public API description.
#ifndef MISIFT_COMPAT_OPENCV_CORE_HPP
#define MISIFT_COMPAT_OPENCV_CORE_HPP
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_8UC1 CV_8U
#define CV_32FC1 CV_32F
#define CV_64FC1 CV_64F

namespace cv {

typedef unsigned char uchar;

struct Scalar {
  double val[4];
  Scalar(double v0 = 0, double v1 = 0, double v2 = 0, double v3 = 0) { val[0] = v0; val[1] = v1; val[2] = v2; val[3] = v3; }
};

enum { DECOMP_LU = 0, DECOMP_SVD = 1, DECOMP_EIG = 2, DECOMP_CHOLESKY = 3, DECOMP_QR = 4 };

class Mat {
public:
  int rows, cols;
  uchar *data;

  Mat() : rows(0), cols(0), data(NULL), type_(CV_8U) {}
  Mat(int r, int c, int type) : rows(0), cols(0), data(NULL), type_(type) { create(r, c, type); }
  Mat(int r, int c, int type, void *ext) : rows(r), cols(c), data((uchar *)ext), type_(type) {}   // borrows

  void create(int r, int c, int type)
  {
    rows = r; cols = c; type_ = type;
    store_.reset(new std::vector<uchar>((size_t)r * c * elemSize(), 0));
    data = store_->data();
  }
  int type() const { return type_; }
  size_t elemSize() const { return type_ == CV_8U ? 1 : (type_ == CV_32F ? 4 : 8); }
  bool empty() const { return data == NULL || rows * cols == 0; }

  template <typename T> T &at(int i, int j) { return ((T *)data)[(size_t)i * cols + j]; }
  template <typename T> const T &at(int i, int j) const { return ((const T *)data)[(size_t)i * cols + j]; }
  template <typename T> T &at(int i) { return ((T *)data)[i]; }
  template <typename T> const T &at(int i) const { return ((const T *)data)[i]; }

  double get(size_t k) const
  {
    return type_ == CV_8U ? (double)data[k] : (type_ == CV_32F ? (double)((const float *)data)[k] : ((const double *)data)[k]);
  }
  void set(size_t k, double v)
  {
    if (type_ == CV_8U) {
      double r = std::nearbyint(v);
      data[k] = (uchar)(r < 0 ? 0 : (r > 255 ? 255 : r));     // saturate_cast<uchar>
    } else if (type_ == CV_32F) {
      ((float *)data)[k] = (float)v;
    } else {
      ((double *)data)[k] = v;
    }
  }

  void convertTo(Mat &dst, int rtype, double alpha = 1.0, double beta = 0.0) const
  {
    Mat out(rows, cols, rtype);
    for (size_t k = 0; k < (size_t)rows * cols; k++) out.set(k, get(k) * alpha + beta);
    dst = out;
  }
  Mat &operator=(const Scalar &s)
  {
    for (size_t k = 0; k < (size_t)rows * cols; k++) set(k, s.val[0]);
    return *this;
  }
  Mat &operator+=(const Mat &o)
  {
    for (size_t k = 0; k < (size_t)rows * cols; k++) set(k, get(k) + o.get(k));
    return *this;
  }
  Mat clone() const
  {
    Mat out(rows, cols, type_);
    if (data) memcpy(out.data, data, (size_t)rows * cols * elemSize());
    return out;
  }

private:
  int type_;
  std::shared_ptr<std::vector<uchar> > store_;
};

inline Mat operator*(const Mat &a, double s)
{
  Mat out(a.rows, a.cols, a.type());
  for (size_t k = 0; k < (size_t)a.rows * a.cols; k++) out.set(k, a.get(k) * s);
  return out;
}
inline Mat operator*(double s, const Mat &a) { return a * s; }

// Solve src1 * dst = src2.  DECOMP_CHOLESKY: symmetric positive definite src1; when the matrix is not positive
// definite it returns false and ZEROES dst, like OpenCV (cv::solve: `if (!result) dst = Scalar(0);`).
inline bool solve(const Mat &src1, const Mat &src2, Mat &dst, int flags = DECOMP_LU)
{
  const int n = src1.rows, m = src2.cols;
  std::vector<double> A((size_t)n * n), B((size_t)n * m);
  for (int i = 0; i < n * n; i++) A[i] = src1.get(i);
  for (int i = 0; i < n * m; i++) B[i] = src2.get(i);
  if (flags == DECOMP_CHOLESKY) {
    for (int i = 0; i < n; i++) {
      for (int j = 0; j <= i; j++) {
        double s = A[(size_t)i * n + j];
        for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
        if (i == j) {
          if (!(s > 0)) {
            Mat zero(n, m, src2.type() == CV_32F ? CV_32F : CV_64F);     // create() zero-fills
            dst = zero;
            return false;
          }
          A[(size_t)i * n + i] = std::sqrt(s);
        } else {
          A[(size_t)i * n + j] = s / A[(size_t)j * n + j];
        }
      }
    }
    for (int c = 0; c < m; c++) {
      for (int i = 0; i < n; i++) {
        double s = B[(size_t)i * m + c];
        for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * B[(size_t)k * m + c];
        B[(size_t)i * m + c] = s / A[(size_t)i * n + i];
      }
      for (int i = n - 1; i >= 0; i--) {
        double s = B[(size_t)i * m + c];
        for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * B[(size_t)k * m + c];
        B[(size_t)i * m + c] = s / A[(size_t)i * n + i];
      }
    }
  } else {
    for (int c = 0; c < n; c++) {            // Gaussian elimination with partial pivoting
      int piv = c;
      for (int r = c + 1; r < n; r++)
        if (std::fabs(A[(size_t)r * n + c]) > std::fabs(A[(size_t)piv * n + c])) piv = r;
      if (A[(size_t)piv * n + c] == 0.0) return false;
      if (piv != c) {
        for (int k = 0; k < n; k++) std::swap(A[(size_t)c * n + k], A[(size_t)piv * n + k]);
        for (int k = 0; k < m; k++) std::swap(B[(size_t)c * m + k], B[(size_t)piv * m + k]);
      }
      for (int r = c + 1; r < n; r++) {
        const double f = A[(size_t)r * n + c] / A[(size_t)c * n + c];
        for (int k = c; k < n; k++) A[(size_t)r * n + k] -= f * A[(size_t)c * n + k];
        for (int k = 0; k < m; k++) B[(size_t)r * m + k] -= f * B[(size_t)c * m + k];
      }
    }
    for (int r = n - 1; r >= 0; r--)
      for (int k = 0; k < m; k++) {
        double s = B[(size_t)r * m + k];
        for (int j = r + 1; j < n; j++) s -= A[(size_t)r * n + j] * B[(size_t)j * m + k];
        B[(size_t)r * m + k] = s / A[(size_t)r * n + r];
      }
  }
  Mat out(n, m, src2.type() == CV_32F ? CV_32F : CV_64F);
  for (int i = 0; i < n * m; i++) out.set(i, B[i]);
  dst = out;
  return true;
}

}  // namespace cv
#endif
