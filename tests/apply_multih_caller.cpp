// apply_multih_caller.cpp — TEST: the call sequence of the reference's ApplyMultiH (M/main.cpp:262-296) against the
// host class, built on CPU with cv_shim.h (OpenCV's ownership rules) and tests/fake_engine.cpp under ASan.
// Everything the reference's caller touches afterwards — labels, points, AFFINITIES (saved to the result file,
// M/main.cpp:429-446), homographies, the images DrawClusters paints — must still be valid memory.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "MultiH.h"

extern "C" void fake_engine_set_mode(int mode);

static int run(int mode, bool own_front_half)
{
    fake_engine_set_mode(mode);
    const int n = 60;
    std::vector<cv::Point2d> srcPointsOrig, dstPointsOrig;
    std::vector<cv::Mat> origAffines;
    for (int i = 0; i < n; ++i) {
        // even points follow the fake engine's model 0 (identity), odd ones its model 1 (x, y) -> ((x + 300) / 1.3, (y + 600) / 1.3)
        const double x = 10.0 + 7 * i + 3 * (i % 5), y = 20.0 + 37 * (i % 11);
        srcPointsOrig.push_back(cv::Point2d(x, y));
        dstPointsOrig.push_back((i % 2) ? cv::Point2d((x + 300.0) / 1.3, (y + 600.0) / 1.3) : cv::Point2d(x, y));
        cv::Mat A(2, 2, CV_64F);                                   // M/main.cpp:394: (cv::Mat_<double>(2,2) << a1, a2, a3, a4)
        A.at<double>(0, 0) = 1.0 + i; A.at<double>(0, 1) = 0.1; A.at<double>(1, 0) = -0.1; A.at<double>(1, 1) = 2.0 + i;
        origAffines.push_back(A);
    }
    cv::Mat img1(480, 640, CV_8UC3), img2(480, 640, CV_8UC3);

    // ---- M/main.cpp:262-296 ----
    MultiH* multiH = new MultiH(2.6, 2.2, 0.005, 0.5, 20);
    if (!own_front_half) {
        const double F[9] = { 0, -1, 2000, 1, 0, -1000, -2000, 1000, 0 }, e2[2] = { 1000, 2000 };
        multiH->SetEpipolarGeometry(F, e2);
    }
    if (!multiH->Process(srcPointsOrig, dstPointsOrig, origAffines)) return 10;
    std::vector<int> labeling;
    multiH->GetLabels(labeling);
    const int iterationNum = multiH->GetIterationNumber();
    if (multiH->GetClusterNumber() < 1) { multiH->Release(); return 11; }
    multiH->DrawClusters(img1, img2, 2);
    std::vector<cv::Point2d> src_points, dst_points;
    std::vector<cv::Mat> affinities;
    std::vector<int> labels;
    multiH->GetLabels(labels);
    double checksum = 0.0;
    if (labels.size() == srcPointsOrig.size()) {
        for (size_t i = 0; i < labels.size(); ++i) checksum += origAffines[i].at<double>(0, 0) + labels[i];
    } else {
        multiH->GetSourcePoints(src_points);
        multiH->GetDestinationPoints(dst_points);
        multiH->GetAffinities(affinities);
        if (affinities.size() != labels.size() || src_points.size() != labels.size()) return 12;
        for (size_t i = 0; i < labels.size(); ++i) {
            const double a00 = affinities[i].at<double>(0, 0), a11 = affinities[i].at<double>(1, 1);
            // the fake engine's refinement adds 1000 to every affinity entry: a dangling or aliased header cannot show that
            if (!(a00 >= 1001.0 && a00 <= 1001.0 + n && a11 >= 1002.0 && a11 <= 1002.0 + n)) return 13;
            checksum += a00 + a11 + labels[i] + src_points[i].x + dst_points[i].y;
        }
    }
    for (int k = 1; k <= multiH->GetClusterNumber(); ++k) {
        cv::Mat H = multiH->GetHomography(k);                     // 1-based, M/MultiH.h:69
        for (int q = 0; q < 9; ++q) checksum += H.at<double>(q / 3, q % 3);
    }
    long painted = 0;
    for (size_t p = 0; p < (size_t)img1.rows * img1.cols * 3; ++p) painted += img1.data[p] != 0;
    std::printf("mode %d front-half %d: clusters %d iterations %d labels %zu painted %ld checksum %.3f\n", mode,
                (int)own_front_half, multiH->GetClusterNumber(), iterationNum, labels.size(), painted, checksum);
    if (painted == 0) return 14;
    delete multiH;
    return 0;
}

int main()
{
    int rc = 0;
    for (int mode = 0; mode < 2; ++mode)
        for (int own = 0; own < 2; ++own)
            if ((rc = run(mode, own != 0)) != 0) { std::printf("FAILED mode %d own %d rc %d\n", mode, own, rc); return rc; }
    std::printf("apply_multih_caller ok\n");
    return 0;
}
