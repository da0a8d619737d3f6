// main.cpp — command-line harness around class MultiH, following the call order of the
// reference's ApplyMultiH (M/main.cpp:232-311): load correspondences (8 numbers per line,
// M/main.cpp:380-412) -> MultiH ctor -> Process -> GetLabels -> GetIterationNumber ->
// GetClusterNumber -> getters -> SavePointsToFile (9 numbers per line, :429-446).
// No imread / DrawClusters / imshow / waitKey (SURVEY A-11: the reference blocks on a key press).
//
//   multih_harness <in_corr.txt> <out_result.txt> [--epipolar <file with F(9) e2x e2y>]
//                  [--thrF 2.6] [--thrH 2.2] [--locality 0.005] [--lambda 0.5] [--min-inliers 20]
//                  [--hypotheses 10000] [--max-models 32] [--seed 1234] [--iterations 0]
//                  [--neighbourhood knn|radius|approx]   (knn, the default: the 16 nearest hits within 1/locality pixels; radius: every hit within
//                  it; approx: what FLANN's default search — 4 randomised KD-trees, 32 checks — finds of them, MultiH::SetNeighbourApprox)
//                  [--load-filter 2.0]   r06: the filter of the reference's LoadPointsFromFile (M/main.cpp:399-409:
//                                findFundamentalMat(CV_FM_RANSAC, 2.0, 0.99) and the erase loop) through the engine's own estimator
//                                (multih::FilterCorrespondencesByEpipolarGeometry); the value is the threshold in pixels, 0 switches
//                                the filter off (what the harness did until r05)
//                  [--f-metric opencv|sampson]   what the two F estimations compare with their thresholds (MultiH::SetFundamentalMetric)
//                  [--stages <file>]   write the stage table as one JSON object: rows loaded, after the load filter, in
//                                Process()'s RANSAC mask, after OptimalTriangulation, after distanceError <= 1 (M/MultiH.cpp:807-838)
//                  [--ranks N]   one process per GPU (rank r on device r), the hypothesis batches sharded over the ranks and
//                                exchanged by RCCL (host/rccl_transport.cpp: ncclAllGather on the engine's stream); this
//                                process becomes rank 0 and starts the others before anything touches the GPU.  Every rank
//                                computes the same result; rank 0 writes <out_result.txt>, rank r > 0 <out_result.txt>.rank<r>.
//                                N = 1 runs the same protocol on a one-rank communicator.
// Defaults are the harness defaults of the reference (M/main.cpp:55-59).
#include <algorithm>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <sys/wait.h>
#include <poll.h>
#include <unistd.h>

#include "MultiH.h"
#include "multih_rccl.h"

// M/main.cpp:380-412.  filter_threshold > 0: the load-time F-RANSAC filter of :399-409 (see MultiH.h,
// multih::FilterCorrespondencesByEpipolarGeometry); *loaded = rows read from the file.
static bool LoadPointsFromFile(std::vector<cv::Point2d>& srcPoints, std::vector<cv::Point2d>& dstPoints,
                               std::vector<cv::Mat>& affines, const char* file, double filter_threshold,
                               unsigned long long seed, int metric, int device, int* loaded)
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
    if (loaded) *loaded = (int)srcPoints.size();
    if (filter_threshold > 0.0 && srcPoints.size() >= 8 &&
        !multih::FilterCorrespondencesByEpipolarGeometry(srcPoints, dstPoints, affines, filter_threshold,
                                                         seed ^ 0x10adf117e4ull, 4000, metric, device))
        return false;
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
                     "[--iterations n] [--neighbourhood knn|radius|approx] [--load-filter px] [--f-metric opencv|sampson] "
                     "[--stages file] [--ranks n]\n";
        return 2;
    }
    double thrF = 2.6, thrH = 2.2, locality = 0.005, lambda = 0.5;     // M/main.cpp:55-59
    int min_inliers = 20, hypotheses = 10000, max_models = 32, iterations = 0;
    unsigned long long seed = 1234;
    int ranks = 0;
    double load_filter = 2.0;                                          // M/main.cpp:400
    int f_metric = MultiH::FUND_EPIPOLAR_MAX;
    std::string epi, neighbourhood = "knn", stages_path;
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
        else if (k == "--ranks") ranks = atoi(v);
        else if (k == "--load-filter") load_filter = atof(v);
        else if (k == "--f-metric") {
            if (std::string(v) == "sampson") f_metric = MultiH::FUND_SAMPSON;
            else if (std::string(v) == "opencv") f_metric = MultiH::FUND_EPIPOLAR_MAX;
            else { std::cerr << "--f-metric: opencv or sampson\n"; return 2; }
        }
        else if (k == "--stages") stages_path = v;
        else { std::cerr << "unknown option " << k << "\n"; return 2; }
    }

    // --ranks N: fork ranks 1..N-1 now — no HIP call has been made yet — then every rank joins the communicator.  RCCL's
    // 128-byte bootstrap id travels from rank 0 to each child through a pipe made before the fork: no file, so no
    // predictable path to plant a link at and no stale id of a crashed run to pick up (r03 advisor finding).
    int rank = 0;
    std::vector<pid_t> kids;
    std::vector<int> id_writers;                 // rank 0: write ends, one per child
    int id_reader = -1;                          // child: read end
    if (ranks > 1) {
        for (int r = 1; r < ranks; ++r) {
            int fds[2];
            if (pipe(fds) != 0) { perror("pipe"); return 1; }
            const pid_t pid = fork();
            if (pid < 0) { perror("fork"); return 1; }
            if (pid == 0) {
                rank = r;
                kids.clear();
                close(fds[1]);
                for (int w : id_writers) close(w);
                id_writers.clear();
                id_reader = fds[0];
                break;
            }
            close(fds[0]);
            id_writers.push_back(fds[1]);
            kids.push_back(pid);
        }
    }
    auto finish = [&](int rc) {
        for (int w : id_writers) close(w);       // (a child still waiting for the id sees end-of-file and gives up)
        id_writers.clear();
        for (pid_t pid : kids) {
            int st = 0;
            if (waitpid(pid, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) {
                std::cerr << "[Multi-H] a rank failed\n";
                if (rc == 0) rc = 1;
            }
        }
        return rc;
    };
    mhr_comm* comm = nullptr;
    int (*allgather)(void*, const void*, void*, unsigned long long, void*) = nullptr;
    void (*comm_destroy)(mhr_comm*) = nullptr;
    if (ranks >= 1) {
        void* lib = dlopen("libmultih_rccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) { std::cerr << "cannot load libmultih_rccl.so: " << dlerror() << "\n"; return finish(1); }
        auto init_id = (int (*)(mhr_comm**, int, int, const unsigned char*, int))dlsym(lib, "mhr_init");
        auto make_id = (int (*)(unsigned char*))dlsym(lib, "mhr_unique_id");
        auto last_error = (const char* (*)())dlsym(lib, "mhr_last_error");
        allgather = (int (*)(void*, const void*, void*, unsigned long long, void*))dlsym(lib, "mhr_allgather");
        comm_destroy = (void (*)(mhr_comm*))dlsym(lib, "mhr_destroy");
        if (!init_id || !make_id || !allgather || !comm_destroy || !last_error) { std::cerr << "libmultih_rccl.so lacks a symbol\n"; return finish(1); }
        unsigned char id[MHR_ID_BYTES];
        int rc = 0;
        if (rank == 0) {
            rc = make_id(id);
            for (int w : id_writers) {
                if (rc == 0 && write(w, id, MHR_ID_BYTES) != (ssize_t)MHR_ID_BYTES) { std::cerr << "[Multi-H] cannot hand the RCCL id to a rank\n"; rc = 1; }
                close(w);
            }
            id_writers.clear();
        } else {
            // the id arrives within two minutes or not at all (end-of-file: rank 0 gave up)
            size_t got = 0;
            struct pollfd pf = { id_reader, POLLIN, 0 };
            while (got < MHR_ID_BYTES) {
                if (poll(&pf, 1, 120000) <= 0) break;
                const ssize_t k = read(id_reader, id + got, MHR_ID_BYTES - got);
                if (k <= 0) break;
                got += (size_t)k;
            }
            close(id_reader);
            if (got != MHR_ID_BYTES) { std::cerr << "[Multi-H] rank " << rank << ": no RCCL id from rank 0\n"; return finish(1); }
        }
        if (rc == 0) {
            signal(SIGALRM, [](int) {
                static const char msg[] = "[Multi-H] the RCCL communicator was not complete after 300 s (a rank died?): giving up\n";
                (void)!write(2, msg, sizeof(msg) - 1);
                _exit(1);
            });
            alarm(300);                          // ncclCommInitRank waits for every rank: a peer that died must not hang the others
            rc = init_id(&comm, rank, std::max(ranks, 1), id, rank);
            alarm(0);
        }
        if (rc != 0) { std::cerr << "[Multi-H] rank " << rank << ": RCCL communicator: " << last_error() << "\n"; return finish(1); }
        printf("[Multi-H] rank %d of %d joined the RCCL communicator (device %d)\n", rank, ranks, rank);
    }

    std::vector<cv::Point2d> srcPointsOrig, dstPointsOrig;
    std::vector<cv::Mat> origAffines;
    int rows_loaded = 0;
    if (!LoadPointsFromFile(srcPointsOrig, dstPointsOrig, origAffines, argv[1], epi.empty() ? load_filter : 0.0, seed, f_metric,
                            comm ? rank : 0, &rows_loaded)) {
        std::cerr << "cannot read " << argv[1] << " (or the load filter failed)\n";
        return finish(1);
    }
    printf("Found %d matches.\n", (int)srcPointsOrig.size());
    const int rows_after_load_filter = (int)srcPointsOrig.size();

    MultiH* multiH = new MultiH(thrF, thrH, locality, lambda, min_inliers);
    if (!epi.empty()) {
        std::ifstream f(epi);
        double F[9], e2[2];
        bool ok = true;
        for (double& x : F) ok = ok && static_cast<bool>(f >> x);
        for (double& x : e2) ok = ok && static_cast<bool>(f >> x);
        if (!ok) { std::cerr << "cannot read epipolar geometry from " << epi << "\n"; return finish(1); }
        multiH->SetEpipolarGeometry(F, e2);
    }
    multiH->SetFundamentalMetric(f_metric);
    multiH->SetProposal(seed, hypotheses, max_models);
    multiH->SetFixedIterations(iterations);
    if (comm) {
        multiH->SetDevice(rank);
        multiH->SetShardingStream(rank, ranks, allgather, comm);
    }
    if (neighbourhood == "radius") multiH->SetNeighbourRadius(1.0 / locality);        // the complete list of M/MultiH.cpp:252-253 (see MultiH.h)
    else if (neighbourhood == "approx") multiH->SetNeighbourApprox(4, 32, 0x464c414e4eull + seed);   // ... as FLANN's default search answers it
    if (!multiH->Process(srcPointsOrig, dstPointsOrig, origAffines)) { delete multiH; return finish(1); }
    const std::string out_path = rank == 0 ? std::string(argv[2]) : std::string(argv[2]) + ".rank" + std::to_string(rank);
    {
        // the stage table: where the rows of the input file go before the loop sees them
        const MultiH::FrontStages st = multiH->GetFrontStages();
        printf("[Multi-H] stages: %d loaded, %d after the load filter (%.2f px), %d in Process()'s RANSAC mask (%.2f px), "
               "%d after OptimalTriangulation, %d after distanceError <= 1\n",
               rows_loaded, rows_after_load_filter, epi.empty() ? load_filter : 0.0, st.in_ransac_mask, thrF, st.triangulated,
               st.affine_consistent);
        if (!stages_path.empty() && rank == 0) {
            std::ofstream sf(stages_path);
            sf << "{\"loaded\": " << rows_loaded << ", \"after_load_filter\": " << rows_after_load_filter
               << ", \"load_filter_px\": " << (epi.empty() ? load_filter : 0.0) << ", \"in_ransac_mask\": " << st.in_ransac_mask
               << ", \"ransac_px\": " << thrF << ", \"after_optimal_triangulation\": " << st.triangulated
               << ", \"after_distance_error\": " << st.affine_consistent << ", \"f_metric\": \""
               << (f_metric == MultiH::FUND_SAMPSON ? "sampson" : "opencv") << "\"}\n";
        }
    }

    std::vector<int> labeling;
    multiH->GetLabels(labeling);
    int iterationNum = multiH->GetIterationNumber();
    if (multiH->GetClusterNumber() < 1) {
        multiH->Release();
        std::cerr << "No homographies were found!\n";
        delete multiH;
        return finish(1);
    }
    printf("[Multi-H] %d clusters, %d iterations, energy %.0f\n", multiH->GetClusterNumber(), iterationNum,
           multiH->GetEnergy());

    std::vector<cv::Point2d> src_points, dst_points;
    std::vector<cv::Mat> affinities;
    std::vector<int> labels;
    multiH->GetLabels(labels);
    bool saved;
    if (labels.size() == srcPointsOrig.size()) {                        // M/main.cpp:287-297
        saved = SavePointsToFile(srcPointsOrig, dstPointsOrig, origAffines, labels, out_path.c_str());
    } else {
        multiH->GetSourcePoints(src_points);
        multiH->GetDestinationPoints(dst_points);
        multiH->GetAffinities(affinities);
        saved = SavePointsToFile(src_points, dst_points, affinities, labels, out_path.c_str());
    }
    multiH->Release();
    delete multiH;
    if (comm) comm_destroy(comm);
    return finish(saved ? 0 : 1);
}
