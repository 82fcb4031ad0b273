// Stand-in for <opencv2/imgproc/imgproc.hpp>: mainSift.cpp includes it but uses nothing from it.
#ifndef MISIFT_COMPAT_OPENCV_IMGPROC_HPP
#define MISIFT_COMPAT_OPENCV_IMGPROC_HPP
#include "../core/core.hpp"
#endif
