// main.cpp — command-line harness around class MultiH, following the call order of the
// reference's ApplyMultiH (M/main.cpp:232-311): load correspondences (8 numbers per line,
// M/main.cpp:380-412) -> MultiH ctor -> Process -> GetLabels -> GetIterationNumber ->
// GetClusterNumber -> getters -> SavePointsToFile (9 numbers per line, :429-446).
// No imread / DrawClusters / imshow / waitKey (SURVEY A-11: the reference blocks on a key press).
//
//   multih_harness <in_corr.txt> <out_result.txt> [--epipolar <file with F(9) e2x e2y>]
//                  [--thrF 2.6] [--thrH 2.2] [--locality 0.005] [--lambda 0.5] [--min-inliers 20]
//                  [--hypotheses 10000] [--max-models 32] [--seed 1234] [--iterations 0]
//                  [--neighbourhood knn|radius]   (knn, the default: the 16 nearest hits within 1/locality pixels; radius: every hit within it)
// Defaults are the harness defaults of the reference (M/main.cpp:55-59).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "MultiH.h"

static bool LoadPointsFromFile(std::vector<cv::Point2d>& srcPoints, std::vector<cv::Point2d>& dstPoints,
                               std::vector<cv::Mat>& affines, const char* file)
{
    std::ifstream infile(file);
    if (!infile.is_open()) return false;
    double x1, y1, x2, y2, a1, a2, a3, a4;
    while (infile >> x1 >> y1 >> x2 >> y2 >> a1 >> a2 >> a3 >> a4) {
        srcPoints.push_back(cv::Point2d(x1, y1));
        dstPoints.push_back(cv::Point2d(x2, y2));
        const double a[4] = { a1, a2, a3, a4 };
        cv::Mat A(2, 2, CV_64F);                                     // owns its storage (M/main.cpp:394 builds a Mat_<double>)
        for (int q = 0; q < 4; ++q) A.at<double>(q / 2, q % 2) = a[q];
        affines.push_back(A);
    }
    // The reference also drops F-RANSAC outliers here (findFundamentalMat, :399-409): OpenCV
    // front end, out of scope (§8(f) row 4).
    return true;
}

static bool SavePointsToFile(std::vector<cv::Point2d>& srcPoints, std::vector<cv::Point2d>& dstPoints,
                             std::vector<cv::Mat>& affines, std::vector<int>& labels, const char* file)
{
    std::ofstream outfile(file, std::ios::out);
    if (!outfile.is_open()) return false;
    for (size_t i = 0; i < srcPoints.size(); ++i)
        outfile << srcPoints[i].x << " " << srcPoints[i].y << " " << dstPoints[i].x << " " << dstPoints[i].y << " "
                << affines[i].at<double>(0, 0) << " " << affines[i].at<double>(0, 1) << " "
                << affines[i].at<double>(1, 0) << " " << affines[i].at<double>(1, 1) << " " << labels[i] << std::endl;
    return true;
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::cerr << "usage: multih_harness <in_corr.txt> <out_result.txt> [--epipolar file] [--thrF v] [--thrH v] "
                     "[--locality v] [--lambda v] [--min-inliers n] [--hypotheses n] [--max-models n] [--seed n] "
                     "[--iterations n] [--neighbourhood knn|radius]\n";
        return 2;
    }
    double thrF = 2.6, thrH = 2.2, locality = 0.005, lambda = 0.5;     // M/main.cpp:55-59
    int min_inliers = 20, hypotheses = 10000, max_models = 32, iterations = 0;
    unsigned long long seed = 1234;
    std::string epi, neighbourhood = "knn";
    for (int i = 3; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        const char* v = argv[i + 1];
        if (k == "--epipolar") epi = v;
        else if (k == "--thrF") thrF = atof(v);
        else if (k == "--thrH") thrH = atof(v);
        else if (k == "--locality") locality = atof(v);
        else if (k == "--lambda") lambda = atof(v);
        else if (k == "--min-inliers") min_inliers = atoi(v);
        else if (k == "--hypotheses") hypotheses = atoi(v);
        else if (k == "--max-models") max_models = atoi(v);
        else if (k == "--seed") seed = strtoull(v, nullptr, 10);
        else if (k == "--iterations") iterations = atoi(v);
        else if (k == "--neighbourhood") neighbourhood = v;
        else { std::cerr << "unknown option " << k << "\n"; return 2; }
    }

    std::vector<cv::Point2d> srcPointsOrig, dstPointsOrig;
    std::vector<cv::Mat> origAffines;
    if (!LoadPointsFromFile(srcPointsOrig, dstPointsOrig, origAffines, argv[1])) {
        std::cerr << "cannot read " << argv[1] << "\n";
        return 1;
    }
    printf("Found %d matches.\n", (int)srcPointsOrig.size());

    MultiH* multiH = new MultiH(thrF, thrH, locality, lambda, min_inliers);
    if (!epi.empty()) {
        std::ifstream f(epi);
        double F[9], e2[2];
        bool ok = true;
        for (double& x : F) ok = ok && static_cast<bool>(f >> x);
        for (double& x : e2) ok = ok && static_cast<bool>(f >> x);
        if (!ok) { std::cerr << "cannot read epipolar geometry from " << epi << "\n"; return 1; }
        multiH->SetEpipolarGeometry(F, e2);
    }
    multiH->SetProposal(seed, hypotheses, max_models);
    multiH->SetFixedIterations(iterations);
    if (neighbourhood == "radius") multiH->SetNeighbourRadius(1.0 / locality);        // the complete list of M/MultiH.cpp:252-253 (see MultiH.h)
    if (!multiH->Process(srcPointsOrig, dstPointsOrig, origAffines)) { delete multiH; return 1; }

    std::vector<int> labeling;
    multiH->GetLabels(labeling);
    int iterationNum = multiH->GetIterationNumber();
    if (multiH->GetClusterNumber() < 1) {
        multiH->Release();
        std::cerr << "No homographies were found!\n";
        delete multiH;
        return 1;
    }
    printf("[Multi-H] %d clusters, %d iterations, energy %.0f\n", multiH->GetClusterNumber(), iterationNum,
           multiH->GetEnergy());

    std::vector<cv::Point2d> src_points, dst_points;
    std::vector<cv::Mat> affinities;
    std::vector<int> labels;
    multiH->GetLabels(labels);
    bool saved;
    if (labels.size() == srcPointsOrig.size()) {                        // M/main.cpp:287-297
        saved = SavePointsToFile(srcPointsOrig, dstPointsOrig, origAffines, labels, argv[2]);
    } else {
        multiH->GetSourcePoints(src_points);
        multiH->GetDestinationPoints(dst_points);
        multiH->GetAffinities(affinities);
        saved = SavePointsToFile(src_points, dst_points, affinities, labels, argv[2]);
    }
    multiH->Release();
    delete multiH;
    return saved ? 0 : 1;
}
