// ref_gco_driver.cpp — thin extern "C" driver around the REFERENCE's own
// alpha-expansion sources, compiled where they lie under /root/reference
// (oracle/Makefile target `ref`; output oracle/_ref/libmh_ref_gco.so, which is
// git-ignored and never contains copied source).  TEST INFRASTRUCTURE ONLY.
//
// It drives GCoptimizationGeneralGraph exactly the way MultiH::LabelingStep
// does (M/MultiH.cpp:520-555): callback data cost + callback Potts smooth
// cost, optional warm start via setLabel, setNeighbors once per DIRECTED hit
// with j != i, expansion(it, 1000), whatLabel.  The data-cost callback here is
// the oracle's restatement of dataEnergy (the original needs OpenCV types);
// everything below the callbacks — graph construction, BK max-flow, cut
// read-out, acceptance test, cycle loop — is the reference's compiled code.
//
// No file from /root/reference is copied: this TU only #includes the header.

#include "GCoptimization.h"
#include <cmath>
#include <cstddef>

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {

struct CostCtx {
    // mode A: dense precomputed cost table
    const int* cost; int L;
    // mode B: dataEnergy restatement (M/MultiH.cpp:473-504)
    const double *x1, *y1, *x2, *y2, *H;
    double lam, T;
    int potts;
};

int dc_table(int p, int l, void* d)
{
    const CostCtx* c = (const CostCtx*)d;
    return c->cost[(size_t)p * c->L + l];
}

int dc_formula(int p, int l, void* d)
{
    const CostCtx* c = (const CostCtx*)d;
    if (l == 0) return round(c->lam * c->T);
    const double* h = c->H + 9 * (size_t)(l - 1);
    const double ox1 = c->x1[p], oy1 = c->y1[p], x2 = c->x2[p], y2 = c->y2[p];
    const double s1 = h[6] * ox1 + h[7] * oy1 + h[8];
    const double x1 = (h[0] * ox1 + h[1] * oy1 + h[2]) / s1;
    const double y1 = (h[3] * ox1 + h[4] * oy1 + h[5]) / s1;
    const double dx = x1 - x2, dy = y1 - y2;
    const double distance = dx * dx + dy * dy;
    if (distance < c->T) return round(c->lam * (1.0f - (distance / c->T)));
    return 2 * round(c->lam * c->T);
}

int sc_potts(int, int, int l1, int l2, void* d)
{
    const CostCtx* c = (const CostCtx*)d;
    return l1 != l2 ? c->potts : 0;
}

int run(int N, int L, CostCtx& ctx, GCoptimization::DataCostFnExtra dc, const int* hit_rowptr,
        const int* hit_col, const int* init_labels, int* labels_out)
{
    try {
        GCoptimizationGeneralGraph* gc = new GCoptimizationGeneralGraph(N, L);
        gc->setDataCost(dc, &ctx);
        gc->setSmoothCost(&sc_potts, &ctx);
        if (init_labels)
            for (int i = 0; i < N; ++i) gc->setLabel(i, init_labels[i]);
        for (int i = 0; i < N; ++i)
            for (int k = hit_rowptr[i]; k < hit_rowptr[i + 1]; ++k)
                if (hit_col[k] != i) gc->setNeighbors(i, hit_col[k]);
        int it = 0;
        const int energy = gc->expansion(it, 1000);
        for (int i = 0; i < N; ++i) labels_out[i] = gc->whatLabel(i);
        delete gc;
        return energy;
    } catch (GCException e) {
        return -2147483647 - 1;
    }
}

} // namespace

// Dense-table variant: labels in GCO numbering (0..L-1).
REF_API int ref_gco_expand_table(int N, int L, const int* cost, const int* hit_rowptr,
                                 const int* hit_col, int potts, const int* init_labels,
                                 int* labels_out)
{
    CostCtx ctx = {};
    ctx.cost = cost; ctx.L = L; ctx.potts = potts;
    return run(N, L, ctx, &dc_table, hit_rowptr, hit_col, init_labels, labels_out);
}

// Formula variant: the LabelingStep optimisation exactly as the reference
// sets it up (callback evaluated lazily per (site,label)).
REF_API int ref_gco_expand_formula(const double* x1, const double* y1, const double* x2,
                                   const double* y2, int N, const double* H, int Nh,
                                   double lambda, double thr2, const int* hit_rowptr,
                                   const int* hit_col, const int* init_labels, int* labels_out)
{
    CostCtx ctx = {};
    ctx.x1 = x1; ctx.y1 = y1; ctx.x2 = x2; ctx.y2 = y2; ctx.H = H;
    ctx.lam = 100.0 / lambda;               // MultiH.h:42
    ctx.T = thr2 * 81.0 / 16.0;             // MultiH.h:44
    ctx.potts = (int)round(100 * lambda);   // MultiH.h:41 + MultiH.cpp:510
    return run(N, Nh + 1, ctx, &dc_formula, hit_rowptr, hit_col, init_labels, labels_out);
}

REF_API int ref_gco_abi_version(void) { return 1; }
