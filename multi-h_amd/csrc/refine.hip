// refine.hip — per-correspondence refinement of GetFundamentalMatrixAndRefineData
// (M/MultiH.cpp:807-838), one thread per point, gfx950.  SURVEY §8(f) row 4.
//
//   1. OptimalTriangulation (:1116-1188): Hartley-Sturm correction of (p1, p2) onto the epipolar
//      geometry: translate both points to the origin, rotate the epipoles onto the x axes
//      (R1, R2 of :801-802, built from the epipoles normalised by their third coordinate, so
//      f1 = f2 = 1), form the degree-6 polynomial (:1136-1142), take the real root of smallest cost
//      (:1152-1167), compare with the cost at infinity (:1169-1175: the point is DROPPED when the
//      optimum is at infinity), map the optimal pair back (:1177-1186).
//      The reference calls cv::solvePoly (OpenCV 3.1.0, not under /root/reference); roots are found
//      here by Durand-Kerner iteration in complex FP64 from the same starting points on every run
//      ("parity unpinned"; restated bit for bit in the oracle).
//   2. GetAffineConsistency (:1057-1090) with GetBetaScale (:1092-1114): distanceError = |A^-T n1 -
//      beta n2|; the point is dropped when it exceeds 1 (:826).
//   3. GetOptimalAffineTransformation (:1190-1223): the affinity closest to A that maps the normal
//      n1 onto beta*n2 — the reference inverts the 6x6 KKT matrix (:1211-1219); its closed form is
//      used here (lambda_k = (n1_k - p.A_col_k) / (p.p), A' = A + p lambda^T, p = beta n2).
// Outputs per point: keep flag, corrected x1 y1 x2 y2, optimal affinity a11 a12 a21 a22.
#include "mh_kernels.hpp"

namespace mh {

struct Cplx { double re, im; };
__device__ __forceinline__ Cplx cmul(Cplx a, Cplx b) { return { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; }
__device__ __forceinline__ Cplx csub(Cplx a, Cplx b) { return { a.re - b.re, a.im - b.im }; }
__device__ __forceinline__ Cplx cadd(Cplx a, Cplx b) { return { a.re + b.re, a.im + b.im }; }
__device__ __forceinline__ Cplx cdiv(Cplx a, Cplx b)
{
    const double den = b.re * b.re + b.im * b.im;
    return { (a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den };
}

// Roots of c[0] + c[1] z + ... + c[n] z^n (n <= 6, c[n] != 0): Durand-Kerner, start values
// (0.4 + 0.9 i)^k, at most 200 sweeps, stop when no root moved by more than 1e-14 (relative + absolute).
__device__ inline void poly_roots(const double* c, int n, Cplx* z)
{
    double m[7];
    for (int k = 0; k <= n; ++k) m[k] = c[k] / c[n];           // monic
    Cplx seed = { 1.0, 0.0 };
    const Cplx base = { 0.4, 0.9 };
    for (int k = 0; k < n; ++k) { z[k] = seed; seed = cmul(seed, base); }
    for (int it = 0; it < 200; ++it) {
        double moved = 0.0;
        for (int k = 0; k < n; ++k) {
            Cplx p = { 1.0, 0.0 };                             // Horner, monic
            for (int j = n - 1; j >= 0; --j) { p = cmul(p, z[k]); p.re = p.re + m[j]; }
            Cplx q = { 1.0, 0.0 };
            for (int j = 0; j < n; ++j) if (j != k) q = cmul(q, csub(z[k], z[j]));
            const Cplx d = cdiv(p, q);
            z[k] = csub(z[k], d);
            const double step = fabs(d.re) + fabs(d.im);
            const double mag = fabs(z[k].re) + fabs(z[k].im);
            if (step > 1e-14 * mag + 1e-300 && step > moved) moved = step;
        }
        if (moved == 0.0) break;
    }
}

struct RefineGeom {           // host-prepared, passed by value
    double F[9];
    double e1x, e1y, e2x, e2y;
};

__global__ void __launch_bounds__(256)
k_refine_points(const double* __restrict__ x1, const double* __restrict__ y1,
                const double* __restrict__ x2, const double* __restrict__ y2,
                const double* __restrict__ a11p, const double* __restrict__ a12p,
                const double* __restrict__ a21p, const double* __restrict__ a22p, int N, RefineGeom g,
                const unsigned char* __restrict__ in_mask, unsigned char* __restrict__ keep,
                double* __restrict__ out /* N x 8: x1 y1 x2 y2 a11 a12 a21 a22 */,
                unsigned char* __restrict__ reason /* N: MH_REFINE_* — which stage of :807-838 a row left at */)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    keep[n] = 0;
    reason[n] = 1;                                             // MH_REFINE_NOT_IN_MASK
    if (in_mask && !in_mask[n]) return;                        // :809 (F-RANSAC mask)
    reason[n] = 2;                                             // MH_REFINE_TRIANGULATION until it succeeds
    const double* F = g.F;
    const double px = x1[n], py = y1[n], qx = x2[n], qy = y2[n];

    // ---- 1. OptimalTriangulation ----
    // F2 = T2^-T F T1^-1 with T^-1 = [1 0 x; 0 1 y; 0 0 1]
    double G[9];                                               // F * T1^-1
    for (int r = 0; r < 3; ++r) {
        G[3 * r] = F[3 * r];
        G[3 * r + 1] = F[3 * r + 1];
        G[3 * r + 2] = (F[3 * r] * px + F[3 * r + 1] * py) + F[3 * r + 2];
    }
    double F2[9];                                              // T2^-T * G, T2^-T = [1 0 0; 0 1 0; x2 y2 1]
    for (int c = 0; c < 3; ++c) {
        F2[c] = G[c];
        F2[3 + c] = G[3 + c];
        F2[6 + c] = (qx * G[c] + qy * G[3 + c]) + G[6 + c];
    }
    // F3 = R2 F2 R1^T,  R1 = [e1x e1y 0; -e1y e1x 0; 0 0 1],  R2 = [-e2x -e2y 0; e2y -e2x 0; 0 0 1]
    double M2[9];                                              // R2 * F2
    for (int c = 0; c < 3; ++c) {
        M2[c] = (-g.e2x) * F2[c] + (-g.e2y) * F2[3 + c];
        M2[3 + c] = g.e2y * F2[c] + (-g.e2x) * F2[3 + c];
        M2[6 + c] = F2[6 + c];
    }
    double F3[9];                                              // M2 * R1^T, R1^T = [e1x -e1y 0; e1y e1x 0; 0 0 1]
    for (int r = 0; r < 3; ++r) {
        F3[3 * r] = M2[3 * r] * g.e1x + M2[3 * r + 1] * g.e1y;
        F3[3 * r + 1] = M2[3 * r] * (-g.e1y) + M2[3 * r + 1] * g.e1x;
        F3[3 * r + 2] = M2[3 * r + 2];
    }
    const double f1 = 1.0, f2 = 1.0;                           // epipoles are (x, y, 1) (:793,:799)
    const double a = F3[4], b = F3[5], c = F3[7], d = F3[8];
    const double f14 = f1 * f1 * f1 * f1, f22 = f2 * f2, f12 = f1 * f1;
    const double adbc = a * d - b * c;
    double t[7];
    t[6] = -a * c * f14 * adbc;
    t[5] = (a * a + f22 * c * c) * (a * a + f22 * c * c) - (a * d + b * c) * f14 * adbc;
    t[4] = 2 * (a * a + f22 * c * c) * (2 * a * b + 2 * c * d * f22) - d * b * f14 * adbc - 2 * a * c * f12 * adbc;
    t[3] = (2 * a * b + 2 * c * d * f22) * (2 * a * b + 2 * c * d * f22) + 2 * (a * a + f22 * c * c) * (b * b + f22 * d * d) -
           2 * f12 * adbc * (a * d + b * c);
    t[2] = 2 * (2 * a * b + 2 * c * d * f22) * (b * b + f22 * d * d) - 2 * (f12 * a * d - f12 * b * c) * b * d - a * c * adbc;
    t[1] = (b * b + f22 * d * d) * (b * b + f22 * d * d) - (a * d + b * c) * adbc;
    t[0] = -adbc * b * d;
    int deg = 6;
    while (deg > 0 && t[deg] == 0.0) --deg;
    double bestS = 2147483647.0, bestT = 0.0;                  // static_cast<double>(INT_MAX), :1147
    if (deg > 0) {
        Cplx z[6];
        poly_roots(t, deg, z);
        for (int i = 0; i < deg; ++i) {
            if (fabs(z[i].im) <= 1e-10) {                      // :1156
                const double tt = z[i].re;
                const double ct = c * tt + d, at = a * tt + b;
                const double val = tt * tt / (1 + f12 * tt * tt) + (ct * ct) / (at * at + f22 * (ct * ct));
                if (val < bestS) { bestS = val; bestT = tt; }
            }
        }
    }
    const double valInf = 1 / f12 + (c * c) / (a * a + f22 * c * c);
    if (valInf < bestS) return;                                // :1170-1175 -> dropped at :816-817
    reason[n] = 3;                                             // MH_REFINE_AFFINE_TEST until it passes
    // point1 = (0, bestT, 1); line2 = F3 point1; point2 = (-l0 l2, -l1 l2, l0^2 + l1^2) / (l0^2 + l1^2)
    const double l0 = F3[1] * bestT + F3[2], l1 = F3[4] * bestT + F3[5], l2 = F3[7] * bestT + F3[8];
    const double w2 = l0 * l0 + l1 * l1;
    const double iw = 1.0 / w2;
    const double p2x = (-l0 * l2) * iw, p2y = (-l1 * l2) * iw;
    // u = (R1 T1)^-1 point1 = T1^-1 R1^-1 point1 ; R1^-1 = (1/s1) [e1x -e1y; e1y e1x]
    const double s1 = g.e1x * g.e1x + g.e1y * g.e1y, s2 = g.e2x * g.e2x + g.e2y * g.e2y;
    const double ux = (g.e1x * 0.0 - g.e1y * bestT) / s1 + px;
    const double uy = (g.e1y * 0.0 + g.e1x * bestT) / s1 + py;
    // R2^-1 = (1/s2) [-e2x e2y; -e2y -e2x]
    const double vx = (-g.e2x * p2x + g.e2y * p2y) / s2 + qx;
    const double vy = (-g.e2y * p2x - g.e2x * p2y) / s2 + qy;

    // ---- 2. affine consistency ----
    const double A11 = a11p[n], A12 = a12p[n], A21 = a21p[n], A22 = a22p[n];
    // l1 = F^T pt2, l2 = F pt1 with the corrected points
    const double L1[3] = { (F[0] * vx + F[3] * vy) + F[6], (F[1] * vx + F[4] * vy) + F[7], (F[2] * vx + F[5] * vy) + F[8] };
    const double L2[3] = { (F[0] * ux + F[1] * uy) + F[2], (F[3] * ux + F[4] * uy) + F[5], (F[6] * ux + F[7] * uy) + F[8] };
    // GetBetaScale (:1092-1114)
    const double xn1 = ux + 1.0;
    const double yn1 = -(L1[0] * xn1 + L1[2]) / L1[1];
    double d1x = xn1 - ux, d1y = yn1 - uy;                     // third component 1 - 1 = 0
    const double nd1 = sqrt(d1x * d1x + d1y * d1y);
    d1x = d1x / nd1; d1y = d1y / nd1;
    const double beta = fabs(sqrt(L2[0] * L2[0] + L2[1] * L2[1]) /
                             ((-F[0] * d1y + F[1] * d1x) * vx + (-F[3] * d1y + F[4] * d1x) * vy - F[6] * d1y + F[7] * d1x));
    // unit normals of the epipolar lines (normalising by the third component first, :1063-1070)
    double n1x = L1[0] / L1[2], n1y = L1[1] / L1[2], n2x = L2[0] / L2[2], n2y = L2[1] / L2[2];
    const double nn1 = sqrt(n1x * n1x + n1y * n1y), nn2 = sqrt(n2x * n2x + n2y * n2y);
    n1x = n1x / nn1; n1y = n1y / nn1; n2x = n2x / nn2; n2y = n2y / nn2;
    // r1 = A^-T n1 ; A^-T = (1/det) [A22 -A21; -A12 A11]
    const double det = A11 * A22 - A12 * A21;
    const double r1x = (A22 * n1x - A21 * n1y) / det, r1y = (-A12 * n1x + A11 * n1y) / det;
    const double ex_ = r1x - beta * n2x, ey_ = r1y - beta * n2y;
    const double distanceError = sqrt(ex_ * ex_ + ey_ * ey_);
    if (distanceError > 1.0) return;                           // :826
    if (!(distanceError <= 1.0)) return;                       // NaN: the reference's `>` keeps it; we drop (documented)

    // ---- 3. optimal affinity ----
    if (n1x * n2x + n1y * n2y < 0) { n2x = -n2x; n2y = -n2y; } // :1207
    const double ppx = beta * n2x, ppy = beta * n2y, pp = ppx * ppx + ppy * ppy;
    const double lam1 = (n1x - (ppx * A11 + ppy * A21)) / pp;
    const double lam2 = (n1y - (ppx * A12 + ppy * A22)) / pp;
    double* o = out + 8 * (size_t)n;
    o[0] = ux; o[1] = uy; o[2] = vx; o[3] = vy;
    o[4] = A11 + ppx * lam1; o[5] = A12 + ppx * lam2; o[6] = A21 + ppy * lam1; o[7] = A22 + ppy * lam2;
    keep[n] = 1;
    reason[n] = 0;                                             // MH_REFINE_KEPT
}

hipError_t launch_refine_points(const Points& p, const Affines& a, const double F[9], const double e1[2],
                                const double e2[2], const unsigned char* in_mask, unsigned char* keep,
                                double* out, unsigned char* reason, hipStream_t s)
{
    if (p.n <= 0) return hipSuccess;
    RefineGeom g;
    for (int i = 0; i < 9; ++i) g.F[i] = F[i];
    g.e1x = e1[0]; g.e1y = e1[1]; g.e2x = e2[0]; g.e2y = e2[1];
    hipLaunchKernelGGL(k_refine_points, dim3((p.n + 255) / 256), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2,
                       a.a11, a.a12, a.a21, a.a22, p.n, g, in_mask, keep, out, reason);
    return hipGetLastError();
}

} // namespace mh
