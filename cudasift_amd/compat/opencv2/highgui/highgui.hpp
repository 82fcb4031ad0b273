// Minimal stand-in for <opencv2/highgui/highgui.hpp>: cv::imread / cv::imwrite for binary
// PGM (P5, maxval 255) only — the format of the reference's sample images
// (mainSift.cpp:36-37, :86).  See core/core.hpp for scope.
#ifndef MISIFT_COMPAT_OPENCV_HIGHGUI_HPP
#define MISIFT_COMPAT_OPENCV_HIGHGUI_HPP
#include <cstdio>
#include <string>
#include "../core/core.hpp"

namespace cv {

inline Mat imread(const std::string &path, int /*flags*/ = 1)
{
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return Mat();
  auto token = [&](char *buf, int cap) -> bool {
    int c = fgetc(f);
    while (c != EOF) {
      if (c == '#') { while (c != EOF && c != '\n') c = fgetc(f); }
      else if (c == ' ' || c == '\n' || c == '\r' || c == '\t') c = fgetc(f);
      else break;
    }
    int n = 0;
    while (c != EOF && c != ' ' && c != '\n' && c != '\r' && c != '\t' && n < cap - 1) { buf[n++] = (char)c; c = fgetc(f); }
    buf[n] = 0;
    return n > 0;
  };
  char t[64];
  if (!token(t, 64) || std::string(t) != "P5") { fclose(f); return Mat(); }
  int w = 0, h = 0, maxv = 0;
  if (token(t, 64)) w = atoi(t);
  if (token(t, 64)) h = atoi(t);
  if (token(t, 64)) maxv = atoi(t);
  if (w <= 0 || h <= 0 || maxv != 255) { fclose(f); return Mat(); }
  Mat img(h, w, CV_8UC1);
  const size_t got = fread(img.data, 1, (size_t)w * h, f);
  fclose(f);
  if (got != (size_t)w * h) return Mat();
  return img;
}

inline bool imwrite(const std::string &path, const Mat &img)
{
  Mat u8;
  if (img.type() == CV_8U) u8 = img; else img.convertTo(u8, CV_8UC1);
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) return false;
  fprintf(f, "P5\n%d %d\n255\n", u8.cols, u8.rows);
  fwrite(u8.data, 1, (size_t)u8.cols * u8.rows, f);
  fclose(f);
  return true;
}

}  // namespace cv
#endif
