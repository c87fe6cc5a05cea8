// C entry point around the reference's NMS2() compiled from its own text (see oracle/Makefile target _ref/libref_nms2.so).
#include "nms2_shim.h"
// the reference's pt_conf_comp + NMS2, extracted at build time from /root/reference/swarm_loop/src/superpoint_tensorrt.cpp:232-310
#include REF_NMS2_SNIPPET

extern "C" int ref_nms2(const float* xy, const float* conf, int n, int img_width, int img_height, int dist_thresh, int max_num, float* out_xy) {
    std::vector<cv::Point2f> det(n), pts;
    cv::Mat c(n > 0 ? n : 1, 1, CV_32F);
    for (int i = 0; i < n; ++i) { det[i] = cv::Point2f(xy[2 * i], xy[2 * i + 1]); c.at<float>(i, 0) = conf[i]; }
    NMS2(det, c, pts, /*border=*/0, dist_thresh, img_width, img_height, max_num);
    for (size_t i = 0; i < pts.size(); ++i) { out_xy[2 * i] = pts[i].x; out_xy[2 * i + 1] = pts[i].y; }
    return (int)pts.size();
}
