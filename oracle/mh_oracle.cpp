// mh_oracle.cpp — CPU ORACLE for the Multi-H propose-score-label hot path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
// `cpu_baseline` leg and __graft_entry__.smoke() may load it, and only as
// the checker.  The shipped engine (multi-h_amd/csrc) never links or calls
// anything in this directory.
//
// It restates, in plain scalar FP64 / int32 C++, the arithmetic of the
// reference (danini/multi-h) for the path BASELINE.json names.  "M/" below is
// /root/reference/MultiH/MultiH/.  Every function cites the lines it follows.
//
// Parity status of each piece:
//   residual / score / data cost / Potts / alpha-expansion labels : PINNED
//       - formulas restated line by line from M/MultiH.cpp,
//       - known-answer constants from the harness defaults (4901/9802/200/50),
//       - alpha-expansion labels+energies checked against the reference's own
//         GCoptimization sources compiled unmodified (oracle/_ref, see Makefile)
//         and against fixtures generated from them (tests/golden).
//   HAF non-minimal re-estimation, collinearity test : formulas pinned to
//       M/MultiH.cpp:913-989 / :446-463; the symmetric eigen-solver they call
//       (cv::eigen, OpenCV 3.1.0 core, not under /root/reference) is restated
//       as a cyclic Jacobi solver -> "parity unpinned at the cv::eigen
//       boundary"; cross-checked with numpy.linalg.eigh in tests.
//   4-point DLT, hypothesis sampling : NO reference source exists (only
//       cv::findHomography call sites) -> "parity unpinned"; this file is the
//       definition (Hartley-normalised DLT, one-sided Jacobi null space,
//       splitmix64 counter RNG); cross-checked with numpy.linalg.svd in tests.
//
//   the merge <-> label alternation (section 11), the post-filter HomographyCompatibilityCheck, stable point
//       sets, the sequential greedy selection and Process() end to end (section 12) : restated from the reference
//       text; every alpha-expansion inside them can run through the reference's own GCoptimization (oracle/_ref);
//       "parity unpinned" at rand() (splitmix64 counters instead) and at the OpenCV primitives under the 3-point
//       solver, the mean shift's summation order (section 8b's engine-order variant) and cv::findHomography.
//
// Build:  g++ -O2 -std=c++14 -ffp-contract=off -fPIC -shared (oracle/Makefile)
// -ffp-contract=off matters: the GPU side is compiled the same way so that
// every FP64 operation rounds once, in the reference's association order.

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <omp.h>

#define MHO_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------
// 1. Forward transfer residual, inlier score            
// ---------------------------------------------------------------------------

// d^2 of one (point, model) pair.  Operation order of M/MultiH.cpp:434-441:
//   s  = h6*x + h7*y + h8          (left-to-right:  (h6*x + h7*y) + h8)
//   u  = (h0*x + h1*y + h2) / s ;  v = (h3*x + h4*y + h5) / s
//   dx = x2 - u ; dy = y2 - v ;    d2 = dx*dx + dy*dy
// (M/MultiH.cpp:495-496,759-760 compute u - x2 instead; the squares are
// bit-identical, so one function serves all four copies of the formula.)
static inline double fwd_d2(const double* h, double x, double y, double x2, double y2)
{
    const double s = h[6] * x + h[7] * y + h[8];
    const double u = (h[0] * x + h[1] * y + h[2]) / s;
    const double v = (h[3] * x + h[4] * y + h[5]) / s;
    const double dx = x2 - u;
    const double dy = y2 - v;
    return dx * dx + dy * dy;
}

// Residual matrix R[m*N + n] (model-major), M/MultiH.cpp:432-443 without the
// threshold.  The reference never stores it; north_star materialises it.
MHO_API void mho_residual_matrix(const double* x1, const double* y1,
                                 const double* x2, const double* y2, int N,
                                 const double* H, int M, double* R)
{
    for (int m = 0; m < M; ++m) {
        const double* h = H + 9 * (size_t)m;
        double* r = R + (size_t)m * N;
        for (int n = 0; n < N; ++n) r[n] = fwd_d2(h, x1[n], y1[n], x2[n], y2[n]);
    }
}

// Symmetric transfer error (north_star wording; NO reference counterpart -> "parity unpinned"):
//   d2 = |H p1 - p2|^2 + |adj(H) p2 - p1|^2, adj(H) = adjugate (H^-1 up to scale), every entry
// (mul, mul, sub); both halves use fwd_d2's operation order.  The engine's optional mode.
static inline void adjugate(const double* h, double* a)
{
    a[0] = h[4] * h[8] - h[5] * h[7]; a[1] = h[2] * h[7] - h[1] * h[8]; a[2] = h[1] * h[5] - h[2] * h[4];
    a[3] = h[5] * h[6] - h[3] * h[8]; a[4] = h[0] * h[8] - h[2] * h[6]; a[5] = h[2] * h[3] - h[0] * h[5];
    a[6] = h[3] * h[7] - h[4] * h[6]; a[7] = h[1] * h[6] - h[0] * h[7]; a[8] = h[0] * h[4] - h[1] * h[3];
}

MHO_API void mho_residual_matrix_sym(const double* x1, const double* y1, const double* x2,
                                     const double* y2, int N, const double* H, int M, double* R)
{
    for (int m = 0; m < M; ++m) {
        const double* h = H + 9 * (size_t)m;
        double a[9];
        adjugate(h, a);
        double* r = R + (size_t)m * N;
        for (int n = 0; n < N; ++n)
            r[n] = fwd_d2(h, x1[n], y1[n], x2[n], y2[n]) + fwd_d2(a, x2[n], y2[n], x1[n], y1[n]);
    }
}

// Inlier counts, strict '<' (M/MultiH.cpp:441; MultipleHomographies.h:166).
// mask (optional, may be NULL): only points with mask[n] != 0 are counted.
MHO_API void mho_score(const double* x1, const double* y1, const double* x2,
                       const double* y2, int N, const double* H, int M,
                       double thr2, const unsigned char* mask, int* counts)
{
    for (int m = 0; m < M; ++m) {
        const double* h = H + 9 * (size_t)m;
        int c = 0;
        for (int n = 0; n < N; ++n) {
            if (mask && !mask[n]) continue;
            if (fwd_d2(h, x1[n], y1[n], x2[n], y2[n]) < thr2) ++c;
        }
        counts[m] = c;
    }
}

// The same count on the SYMMETRIC transfer error (north_star's wording; no reference counterpart, so this restatement of
// the definition  d2 = ||H p1 - p2||^2 + ||adj(H) p2 - p1||^2  is the checker; tests/test_gpu_symmetric.py also checks the
// definition itself in exact rationals).
MHO_API void mho_score_sym(const double* x1, const double* y1, const double* x2, const double* y2, int N, const double* H,
                           int M, double thr2, const unsigned char* mask, int* counts)
{
    for (int m = 0; m < M; ++m) {
        const double* h = H + 9 * (size_t)m;
        double a[9];
        adjugate(h, a);
        int c = 0;
        for (int n = 0; n < N; ++n) {
            if (mask && !mask[n]) continue;
            if (fwd_d2(h, x1[n], y1[n], x2[n], y2[n]) + fwd_d2(a, x2[n], y2[n], x1[n], y1[n]) < thr2) ++c;
        }
        counts[m] = c;
    }
}

// Same loop, OpenMP over models: the "best-effort CPU" baseline B2 of BASELINE.md
// (the reference's own loop is serial, M/MultiH.cpp:415-466).  Returns threads used.
MHO_API int mho_score_mt(const double* x1, const double* y1, const double* x2,
                         const double* y2, int N, const double* H, int M,
                         double thr2, int* counts)
{
    int threads = 1;
#pragma omp parallel
    {
#pragma omp single
        threads = omp_get_num_threads();
#pragma omp for schedule(static)
        for (int m = 0; m < M; ++m) {
            const double* h = H + 9 * (size_t)m;
            int c = 0;
            for (int n = 0; n < N; ++n)
                if (fwd_d2(h, x1[n], y1[n], x2[n], y2[n]) < thr2) ++c;
            counts[m] = c;
        }
    }
    return threads;
}

// ComputeInliersOfHomography, M/MultiH.cpp:743-768: label[i] = idx where the
// point is an inlier, untouched otherwise.
MHO_API void mho_inliers_of_homography(const double* x1, const double* y1,
                                       const double* x2, const double* y2, int N,
                                       const double* h, double thr2, int idx,
                                       int* labeling)
{
    for (int n = 0; n < N; ++n)
        if (fwd_d2(h, x1[n], y1[n], x2[n], y2[n]) < thr2) labeling[n] = idx;
}

// ---------------------------------------------------------------------------
// 2. PEARL data cost and Potts smoothness               
// ---------------------------------------------------------------------------

// dataEnergy, M/MultiH.cpp:473-504 with EnergyDataStruct M/MultiH.h:33-46:
//   lam = 100/lambda ; T = thr2*81/16
//   l == 0           -> round(lam*T)
//   d2 < T           -> round(lam*(1 - d2/T))     (A-4: decreasing in d2)
//   otherwise        -> 2*round(lam*T)
// round() is C round (half away from zero); the double->int conversion is the
// implicit one of the reference's `return round(...)` from an int function.
static inline int data_energy(const double* H, int l, double x, double y,
                              double x2, double y2, double lam, double T)
{
    if (l == 0) return (int)round(lam * T);
    const double d2 = fwd_d2(H + 9 * (size_t)(l - 1), x, y, x2, y2);
    if (d2 < T) return (int)round(lam * (1.0 - (d2 / T)));
    return 2 * (int)round(lam * T);
}

// cost[n*(Nh+1) + l]  (site-major, the layout GCO's array data cost uses).
MHO_API void mho_data_cost(const double* x1, const double* y1, const double* x2,
                           const double* y2, int N, const double* H, int Nh,
                           double lambda, double thr2, int* cost)
{
    const double lam = 100.0 / lambda;        // one_per_energy_lambda, MultiH.h:42
    const double T = thr2 * 81.0 / 16.0;      // truncated_sqr_threshold, MultiH.h:44
    const int L = Nh + 1;
    for (int n = 0; n < N; ++n)
        for (int l = 0; l < L; ++l)
            cost[(size_t)n * L + l] = data_energy(H, l, x1[n], y1[n], x2[n], y2[n], lam, T);
}

// smoothnessEnergy, M/MultiH.cpp:506-511: l1 != l2 ? round(100*lambda) : 0.
MHO_API int mho_potts(double lambda) { return (int)round(100.0 * lambda); }

// ---------------------------------------------------------------------------
// 3. Neighbourhood semantics                             
// ---------------------------------------------------------------------------

// LabelingStep calls setNeighbors(i, j) for EVERY directed hit j != i
// (M/MultiH.cpp:532-540); each call appends j to i's list and i to j's list
// (GCoptimization.cpp:1672-1679).  The pair weight seen by the energy is thus
// mult(i,j) = #[i->j] + #[j->i].  This builds that symmetric weighted CSR.
struct SymGraph {
    std::vector<int> rowptr, col, w;
};

static void build_sym(int N, const int* hit_rowptr, const int* hit_col, SymGraph& g)
{
    std::vector<std::vector<int>> adj(N);
    for (int i = 0; i < N; ++i)
        for (int k = hit_rowptr[i]; k < hit_rowptr[i + 1]; ++k) {
            const int j = hit_col[k];
            if (j == i) continue;             // M/MultiH.cpp:537
            adj[i].push_back(j);
            adj[j].push_back(i);
        }
    g.rowptr.assign(N + 1, 0);
    g.col.clear();
    g.w.clear();
    for (int i = 0; i < N; ++i) {
        std::sort(adj[i].begin(), adj[i].end());
        for (size_t k = 0; k < adj[i].size();) {
            size_t e = k;
            while (e < adj[i].size() && adj[i][e] == adj[i][k]) ++e;
            g.col.push_back(adj[i][k]);
            g.w.push_back((int)(e - k));
            k = e;
        }
        g.rowptr[i + 1] = (int)g.col.size();
    }
}

// Exposed so tests can check the engine's own graph builder against it.
// Returns nnz; if out arrays are NULL only counts.
MHO_API int mho_build_sym_graph(int N, const int* hit_rowptr, const int* hit_col,
                                int* rowptr, int* col, int* w)
{
    SymGraph g;
    build_sym(N, hit_rowptr, hit_col, g);
    if (rowptr) std::copy(g.rowptr.begin(), g.rowptr.end(), rowptr);
    if (col) std::copy(g.col.begin(), g.col.end(), col);
    if (w) std::copy(g.w.begin(), g.w.end(), w);
    return (int)g.col.size();
}

// ---------------------------------------------------------------------------
// 4. alpha-expansion                                     
// ---------------------------------------------------------------------------
// Restates GCoptimization::expansion's standard-cycle branch
// (GCoptimization.cpp:1032-1049), oneExpansionIteration (:1278-1289),
// alpha_expansion (:1212-1274), the graph construction of
// setupDataCostsExpansion (:336-342) / setupSmoothCostsExpansion (:346-373)
// through Energy::add_term1/add_term2 (energy.h:204-253), and the cut read-out
// get_var == what_segment (graph.h:478-488): a node is SINK (var 1, keeps its
// label) iff it can reach t in the final residual graph, SOURCE (var 0, takes
// alpha) otherwise (SURVEY A-1).  The max-flow itself is Dinic's algorithm —
// any exact max-flow yields the same canonical cut, which is what the
// reference comparison in tests/test_oracle_vs_ref.py demonstrates.
// All energies are int32 as in the reference (GCoptimization.h:166-170).

namespace {

struct Dinic {
    int n;
    std::vector<int> head, nxt, to, cap, level, it;
    explicit Dinic(int n_) : n(n_), head(n_, -1) {}
    void add_edge(int u, int v, int c, int rc)
    {
        to.push_back(v); cap.push_back(c); nxt.push_back(head[u]); head[u] = (int)to.size() - 1;
        to.push_back(u); cap.push_back(rc); nxt.push_back(head[v]); head[v] = (int)to.size() - 1;
    }
    bool bfs(int s, int t)
    {
        level.assign(n, -1);
        std::vector<int> q; q.reserve(n);
        q.push_back(s); level[s] = 0;
        for (size_t qi = 0; qi < q.size(); ++qi) {
            int u = q[qi];
            for (int e = head[u]; e >= 0; e = nxt[e])
                if (cap[e] > 0 && level[to[e]] < 0) { level[to[e]] = level[u] + 1; q.push_back(to[e]); }
        }
        return level[t] >= 0;
    }
    // iterative DFS augment
    long long maxflow(int s, int t)
    {
        long long flow = 0;
        std::vector<int> path_e;
        while (bfs(s, t)) {
            it = head;
            for (;;) {
                // find one augmenting path in the level graph
                path_e.clear();
                int u = s;
                bool found = false;
                for (;;) {
                    if (u == t) { found = true; break; }
                    int& e = it[u];
                    while (e >= 0 && !(cap[e] > 0 && level[to[e]] == level[u] + 1)) e = nxt[e];
                    if (e >= 0) { path_e.push_back(e); u = to[e]; }
                    else {
                        if (path_e.empty()) break;
                        level[u] = -1;          // dead end
                        int pe = path_e.back(); path_e.pop_back();
                        u = to[pe ^ 1];
                        it[u] = nxt[it[u]];
                    }
                }
                if (!found) break;
                int f = INT32_MAX;
                for (int e : path_e) f = std::min(f, cap[e]);
                for (int e : path_e) { cap[e] -= f; cap[e ^ 1] += f; }
                flow += f;
            }
        }
        return flow;
    }
};

struct Expander {
    int N, L;
    const int* cost;             // N*L
    const SymGraph* g;
    int potts;
    std::vector<int> label, curCost, lookup;

    int smooth_energy() const     // giveSmoothEnergyInternal, GCoptimization.cpp:267-286
    {
        int e = 0;
        for (int i = 0; i < N; ++i)
            for (int k = g->rowptr[i]; k < g->rowptr[i + 1]; ++k) {
                int j = g->col[k];
                if (j < i && label[i] != label[j]) e += g->w[k] * potts;
            }
        return e;
    }
    int data_energy_sum() const
    {
        int e = 0;
        for (int i = 0; i < N; ++i) e += curCost[i];
        return e;
    }
    int compute_energy() const { return data_energy_sum() + smooth_energy(); } // :953-956

    bool alpha_expansion(int alpha)          // GCoptimization.cpp:1212-1274
    {
        std::vector<int> active;
        for (int i = 0; i < N; ++i) if (label[i] != alpha) active.push_back(i); // :324-332
        const int size = (int)active.size();
        if (size == 0) return false;
        for (int v = 0; v < size; ++v) lookup[active[v]] = v;

        // t-link accumulators: add_tweights(x, cap_source, cap_sink)
        std::vector<long long> srcCap(size, 0), snkCap(size, 0);
        long long before = 0;                 // m_beforeExpansionEnergy
        Dinic G(size + 2);
        const int S = size, T = size + 1;
        // data terms: add_term1(i, E0 = cost(site,alpha), E1 = current cost)  (:336-342)
        for (int v = 0; v < size; ++v) {
            const int site = active[v];
            const int e0 = cost[(size_t)site * L + alpha], e1 = curCost[site];
            before += e1;
            srcCap[v] += e1; snkCap[v] += e0;   // add_tweights(x, B=E1, A=E0), energy.h:207
        }
        // smooth terms (:346-373), Potts
        for (int v = size - 1; v >= 0; --v) {
            const int site = active[v];
            for (int k = g->rowptr[site]; k < g->rowptr[site + 1]; ++k) {
                const int nSite = g->col[k], w = g->w[k];
                if (lookup[nSite] == -1) {
                    // neighbour already has alpha: term1(E0 = V(alpha,alpha)=0, E1 = V(l_site, alpha))
                    const int e1 = potts * w;   // l_site != alpha by construction
                    before += e1;
                    srcCap[v] += e1;
                } else if (nSite < site) {
                    const int u = lookup[nSite];
                    const int e00 = 0, e01 = potts * w, e10 = potts * w;
                    const int e11 = (label[site] != label[nSite]) ? potts * w : 0;
                    before += e11;
                    // add_term2(x=v, y=u, A=e00, B=e01, C=e10, D=e11), energy.h:211-253
                    srcCap[v] += e11; snkCap[v] += e00;
                    const int B = e01 - e00, C = e10 - e11;   // both >= 0 for Potts
                    G.add_edge(v, u, B, C);
                }
            }
        }
        for (int v = 0; v < size; ++v) {
            // Graph::add_tweights keeps only the difference and books min() as flow
            const long long m = std::min(srcCap[v], snkCap[v]);
            const long long s = srcCap[v] - m, t = snkCap[v] - m;
            if (s > 0) G.add_edge(S, v, (int)s, 0);
            if (t > 0) G.add_edge(v, T, (int)t, 0);
        }
        long long constFlow = 0;
        for (int v = 0; v < size; ++v) constFlow += std::min(srcCap[v], snkCap[v]);
        const long long after = constFlow + G.maxflow(S, T);      // e.minimize()

        const bool accept = after < before;                       // strict, :1259
        if (accept) {
            // canonical cut: reverse BFS from T over arcs with residual capacity
            std::vector<char> reachT(size + 2, 0);
            std::vector<int> q; q.push_back(T); reachT[T] = 1;
            for (size_t qi = 0; qi < q.size(); ++qi) {
                int x = q[qi];
                // arcs y->x with residual cap > 0: for edge e out of x, reverse e^1 goes to[e]->x
                for (int e = G.head[x]; e >= 0; e = G.nxt[e]) {
                    int y = G.to[e];
                    if (!reachT[y] && G.cap[e ^ 1] > 0) { reachT[y] = 1; q.push_back(y); }
                }
            }
            // applyNewLabeling (:423-441): var == 0 (SOURCE) takes alpha
            for (int v = 0; v < size; ++v)
                if (!reachT[v]) {
                    const int site = active[v];
                    label[site] = alpha;
                    curCost[site] = cost[(size_t)site * L + alpha];
                }
        }
        for (int v = 0; v < size; ++v) lookup[active[v]] = -1;
        return accept;
    }
};

} // namespace

// labels: in = initial labeling (GCO default is all 0; LabelingStep warm-starts
// with previous+1 when !changed, M/MultiH.cpp:525-529), out = result.
// hits CSR: directed radius/kNN hits (self hits allowed, skipped).
// Returns final int32 energy; *cycles_out = number of cycles executed.
MHO_API int mho_expand(int N, int L, const int* cost, const int* hit_rowptr,
                       const int* hit_col, int potts, int* labels, int max_cycles,
                       int* cycles_out, int* energies_out /* optional, per cycle */)
{
    SymGraph g;
    build_sym(N, hit_rowptr, hit_col, g);
    if (cycles_out) *cycles_out = 0;

    // solveSpecialCases (GCoptimization.cpp:455-491): no neighbours at all ->
    // independent per-site argmin, first minimum wins.
    if (g.col.empty()) {
        int energy = 0;
        for (int i = 0; i < N; ++i) {
            int best = 0, bc = cost[(size_t)i * L];
            for (int l = 1; l < L; ++l) {
                int c = cost[(size_t)i * L + l];
                if (c < bc) { bc = c; best = l; }
            }
            labels[i] = best;
            energy += bc;
        }
        return energy;
    }

    Expander ex;
    ex.N = N; ex.L = L; ex.cost = cost; ex.g = &g; ex.potts = potts;
    ex.label.assign(labels, labels + N);
    ex.curCost.resize(N);
    ex.lookup.assign(N, -1);
    for (int i = 0; i < N; ++i) ex.curCost[i] = cost[(size_t)i * L + ex.label[i]]; // :445-451

    int new_energy = ex.compute_energy();     // :1036
    int old_energy;
    int cyc = 0;
    for (int cycle = 1; cycle <= max_cycles; ++cycle) {
        old_energy = new_energy;
        for (int a = 0; a < L; ++a) ex.alpha_expansion(a);   // fixed order, :1285-1286
        new_energy = ex.compute_energy();
        if (energies_out) energies_out[cyc] = new_energy;
        ++cyc;
        if (new_energy == old_energy) break;                 // :1045
    }
    if (cycles_out) *cycles_out = cyc;
    std::copy(ex.label.begin(), ex.label.end(), labels);
    return new_energy;
}

// Total energy of a given labeling (data + smooth), for property tests.
MHO_API long long mho_labeling_energy(int N, int L, const int* cost, const int* hit_rowptr,
                                      const int* hit_col, int potts, const int* labels)
{
    SymGraph g;
    build_sym(N, hit_rowptr, hit_col, g);
    long long e = 0;
    for (int i = 0; i < N; ++i) {
        e += cost[(size_t)i * L + labels[i]];
        for (int k = g.rowptr[i]; k < g.rowptr[i + 1]; ++k)
            if (g.col[k] < i && labels[i] != labels[g.col[k]]) e += (long long)g.w[k] * potts;
    }
    return e;
}

// ---------------------------------------------------------------------------
// 5. Symmetric eigen-solver (stands in for cv::eigen, OpenCV 3.1.0)
// ---------------------------------------------------------------------------
// Cyclic Jacobi, fixed (p,q) order, rotation skipped when the off-diagonal is
// exactly zero; stops after a sweep whose off-diagonal sum of squares is
// <= 1e-30 * (sum of squares of the diagonal), max 30 sweeps.  a is n*n row
// major (destroyed), v gets eigenvectors as COLUMNS, d the eigenvalues
// (unsorted).  The GPU re-estimation kernel uses the same recurrence so that
// both sides round identically.
static void jacobi_sym(int n, double* a, double* v, double* d)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) v[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag = diag + a[i * n + i] * a[i * n + i];
            for (int j = i + 1; j < n; ++j) off = off + a[i * n + j] * a[i * n + j];
        }
        if (off <= 1e-30 * diag) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                for (int k = 0; k < n; ++k) {            // columns p,q of A
                    const double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - s * akq;
                    a[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {            // rows p,q of A
                    const double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - s * aqk;
                    a[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {            // accumulate V
                    const double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = c * vkp - s * vkq;
                    v[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) d[i] = a[i * n + i];
}

MHO_API void mho_jacobi_sym(int n, const double* a_in, double* evecs_cols, double* evals)
{
    std::vector<double> a(a_in, a_in + n * n);
    jacobi_sym(n, a.data(), evecs_cols, evals);
}

// ---------------------------------------------------------------------------
// 6. Deterministic strided-tree reduction
// ---------------------------------------------------------------------------
// FP sums that feed thresholds or homographies must round identically on CPU
// and GPU.  The reference's own summation order is inside OpenCV (gemm) and
// unknowable, so the order is DEFINED here: W = 256 strided partial sums
// (lane t adds items t, t+W, t+2W, ... in increasing order, starting from
// 0.0), then a binary tree  v[t] += v[t+s]  for s = 128, 64, ..., 1.
enum { MHO_W = 256 };

struct TreeAcc {
    int K;
    std::vector<double> part;                 // W*K
    explicit TreeAcc(int k) : K(k), part((size_t)MHO_W * k, 0.0) {}
    void add(int item_index, const double* contrib)
    {
        double* p = &part[(size_t)(item_index % MHO_W) * K];
        for (int k = 0; k < K; ++k) p[k] = p[k] + contrib[k];
    }
    void finish(double* out)
    {
        for (int s = MHO_W / 2; s >= 1; s >>= 1)
            for (int t = 0; t < s; ++t)
                for (int k = 0; k < K; ++k)
                    part[(size_t)t * K + k] = part[(size_t)t * K + k] + part[(size_t)(t + s) * K + k];
        for (int k = 0; k < K; ++k) out[k] = part[k];
    }
};

// ---------------------------------------------------------------------------
// 7. Collinearity (straightness) test                    
// ---------------------------------------------------------------------------
// M/MultiH.cpp:446-463: S = sum over inliers of [x y 1]^T [x y 1]; reject the
// model if the smallest eigenvalue of S < straightness_threshold (0.005) or
// inliers < 3.  moments out: {n, Sx, Sy, Sxx, Sxy, Syy} per model (n as double).
// Items are indexed by POINT index n (not by inlier rank) in the tree sum.
MHO_API void mho_inlier_moments(const double* x1, const double* y1, const double* x2,
                                const double* y2, int N, const double* H, int M,
                                double thr2, double* moments /* M*6 */, double* min_eig /* M */)
{
    for (int m = 0; m < M; ++m) {
        const double* h = H + 9 * (size_t)m;
        TreeAcc acc(5);
        int cnt = 0;
        for (int n = 0; n < N; ++n) {
            if (fwd_d2(h, x1[n], y1[n], x2[n], y2[n]) < thr2) {
                const double x = x1[n], y = y1[n];
                const double c[5] = { x, y, x * x, x * y, y * y };
                acc.add(n, c);
                ++cnt;
            }
        }
        double s[5];
        acc.finish(s);
        double* mo = moments + 6 * (size_t)m;
        mo[0] = (double)cnt; mo[1] = s[0]; mo[2] = s[1]; mo[3] = s[2]; mo[4] = s[3]; mo[5] = s[4];
        if (min_eig) {
            double a[9] = { s[2], s[3], s[0],  s[3], s[4], s[1],  s[0], s[1], (double)cnt };
            double v[9], d[3];
            jacobi_sym(3, a, v, d);
            min_eig[m] = std::min(d[0], std::min(d[1], d[2]));
        }
    }
}

// ---------------------------------------------------------------------------
// 8. HAF non-minimal re-estimation                       
// ---------------------------------------------------------------------------
// GetHomographyHAFNonminimal, M/MultiH.cpp:913-989, for every label at once:
// six rows per point (:938-966), A^T A (4x4), eigenvector of the smallest
// eigenvalue (:970-982), rows 1-2 of H from e2, F, lambda (:984-989), then the
// in-place rescale of RefineHomographyHAF (Homography_RefineHAFCallback.h:33-34).
// The LM loop that follows has no effect on the output (SURVEY A-3) and reads
// out of bounds; it is deliberately not restated.
// A^T A is accumulated with the strided tree (items indexed by site index).
// Labels: -1 = outlier, 0..Nh-1.  A label without points keeps its H
// (M/MultiH.cpp:592-593).
static void haf_rows(double a11, double a12, double a21, double a22, double x1, double y1,
                     double x2, double y2, const double* F, double ex, double ey,
                     double r[6][4])
{
    r[0][0] = a11 * x1 + x2 - ex; r[0][1] = a11 * y1;           r[0][2] = a11; r[0][3] = -F[3];
    r[1][0] = a12 * x1;           r[1][1] = a12 * y1 + x2 - ex; r[1][2] = a12; r[1][3] = -F[4];
    r[2][0] = a21 * x1 + y2 - ey; r[2][1] = a21 * y1;           r[2][2] = a21; r[2][3] = F[0];
    r[3][0] = a22 * x1;           r[3][1] = a22 * y1 + y2 - ey; r[3][2] = a22; r[3][3] = F[1];
    r[4][0] = ex * x1 - x2 * x1;  r[4][1] = ex * y1 - x2 * y1;  r[4][2] = ex - x2;
    r[4][3] = x1 * F[3] + y1 * F[4] + F[5];
    r[5][0] = ey * x1 - y2 * x1;  r[5][1] = ey * y1 - y2 * y1;  r[5][2] = ey - y2;
    r[5][3] = -(x1 * F[0] + y1 * F[1] + F[2]);
}

MHO_API void mho_haf_reestimate(const double* x1, const double* y1, const double* x2,
                                const double* y2, const double* aff /* N*4: a11 a12 a21 a22 */,
                                int N, const int* labels, int Nh, const double* F,
                                const double* e2 /* ex, ey */, double* H /* Nh*9 in/out */,
                                int* counts_out /* optional Nh */)
{
    const double ex = e2[0], ey = e2[1];
    for (int l = 0; l < Nh; ++l) {
        TreeAcc acc(10);
        int cnt = 0;
        for (int n = 0; n < N; ++n) {
            if (labels[n] != l) continue;
            double r[6][4];
            haf_rows(aff[4 * n], aff[4 * n + 1], aff[4 * n + 2], aff[4 * n + 3],
                     x1[n], y1[n], x2[n], y2[n], F, ex, ey, r);
            double c[10];
            int k = 0;
            for (int i = 0; i < 4; ++i)
                for (int j = i; j < 4; ++j) {
                    double s = r[0][i] * r[0][j];
                    for (int q = 1; q < 6; ++q) s = s + r[q][i] * r[q][j];
                    c[k++] = s;
                }
            acc.add(n, c);
            ++cnt;
        }
        if (counts_out) counts_out[l] = cnt;
        if (cnt == 0) continue;
        double u[10];
        acc.finish(u);
        double a[16], v[16], d[4];
        int k = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = i; j < 4; ++j) { a[i * 4 + j] = u[k]; a[j * 4 + i] = u[k]; ++k; }
        jacobi_sym(4, a, v, d);
        int jm = 0;
        for (int j = 1; j < 4; ++j) if (d[j] < d[jm]) jm = j;
        const double h6 = v[0 * 4 + jm], h7 = v[1 * 4 + jm], h8 = v[2 * 4 + jm], lam = v[3 * 4 + jm];
        double* h = H + 9 * (size_t)l;
        h[6] = h6; h[7] = h7; h[8] = h8;
        h[3] = ey * h6 - lam * F[0];
        h[4] = ey * h7 - lam * F[1];
        h[5] = ey * h8 - lam * F[2];
        h[0] = ex * h6 + lam * F[3];
        h[1] = ex * h7 + lam * F[4];
        h[2] = ex * h8 + lam * F[5];
        // RefineHomographyHAF rescale, Homography_RefineHAFCallback.h:33-34
        const double lam2 = (h[0] - ex * h[6]) / F[3];
        const double inv = 1.0 / lam2;
        for (int q = 0; q < 9; ++q) h[q] = h[q] * inv;
    }
}

// ---------------------------------------------------------------------------
// 9. Hypothesis sampling + 4-point DLT      (north_star; no reference source)
// ---------------------------------------------------------------------------
static inline uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// K distinct indices (K = 4: homography samples, K = 8: fundamental-matrix samples), same draw
// scheme; at most `maxdraw` draws.
static void sample_tuple(uint64_t seed, uint64_t m, uint32_t N, int K, int maxdraw, int* out)
{
    int got = 0;
    for (uint32_t c = 0; c < (uint32_t)maxdraw && got < K; ++c) {
        const uint64_t r = splitmix64(seed + (m << 8) + c);
        const int idx = (int)(((r >> 32) * (uint64_t)N) >> 32);
        bool dup = false;
        for (int k = 0; k < got; ++k) dup = dup || (out[k] == idx);
        if (!dup) out[got++] = idx;
    }
    for (; got < K; ++got) out[got] = out[0];
}

// Four distinct point indices for hypothesis m: draw c = 0,1,2,... gives
// r = splitmix64(seed + (m << 8) + c), idx = ((r >> 32) * N) >> 32; a draw equal
// to an earlier member of the tuple is rejected (sampling without replacement
// inside a tuple, as MultipleHomographies.h:118-125 does).  At most 64 draws.
static void sample4(uint64_t seed, uint64_t m, uint32_t N, int out[4])
{
    int got = 0;
    for (uint32_t c = 0; c < 64 && got < 4; ++c) {
        const uint64_t r = splitmix64(seed + (m << 8) + c);
        const int idx = (int)(((r >> 32) * (uint64_t)N) >> 32);
        bool dup = false;
        for (int k = 0; k < got; ++k) dup = dup || (out[k] == idx);
        if (!dup) out[got++] = idx;
    }
    for (; got < 4; ++got) out[got] = out[0];   // unreachable for N >= 4
}

MHO_API void mho_sample4(unsigned long long seed, long long m0, int M, int N, int* idx /* M*4 */)
{
    for (int m = 0; m < M; ++m) sample4(seed, (uint64_t)(m0 + m), (uint32_t)N, idx + 4 * (size_t)m);
}

// Round-robin (circle method) schedule for 9 columns + 1 bye: 9 rounds of 4
// disjoint pairs.  Disjoint pairs commute exactly, so a GPU wave rotating the 4
// pairs of a round in parallel and this serial loop produce the same bits.
static void rr_schedule(int sched[9][4][2])
{
    for (int r = 0; r < 9; ++r) {
        int cnt = 0;
        // players 0..8 on a circle, player 9 (bye) fixed
        for (int k = 1; k <= 4; ++k) {
            int a = (r + k) % 9, b = (r + 9 - k) % 9;
            sched[r][cnt][0] = std::min(a, b);
            sched[r][cnt][1] = std::max(a, b);
            ++cnt;
        }
    }
}

MHO_API void mho_rr_schedule(int* out /* 9*4*2 */)
{
    int s[9][4][2];
    rr_schedule(s);
    memcpy(out, s, sizeof(s));
}

// One normalised DLT solve.  src/dst: 4 points each.  Returns H (row-major 9,
// Frobenius norm 1, h[8] >= 0) and the ratio second-smallest/largest column
// norm as a conditioning witness (tests mask degenerate samples with it).
static void dlt4(const double sx[4], const double sy[4], const double dx[4], const double dy[4],
                 double H[9], double* witness, int* sweeps_out)
{
    // Hartley normalisation (same recipe as NormalizePoints,
    // Homography_Refine3PTCallback.h:236-271: centroid, mean distance -> sqrt 2)
    double cx1 = ((sx[0] + sx[1]) + sx[2]) + sx[3], cy1 = ((sy[0] + sy[1]) + sy[2]) + sy[3];
    double cx2 = ((dx[0] + dx[1]) + dx[2]) + dx[3], cy2 = ((dy[0] + dy[1]) + dy[2]) + dy[3];
    cx1 = cx1 * 0.25; cy1 = cy1 * 0.25; cx2 = cx2 * 0.25; cy2 = cy2 * 0.25;
    double d1 = 0.0, d2 = 0.0;
    for (int i = 0; i < 4; ++i) {
        const double ax = sx[i] - cx1, ay = sy[i] - cy1, bx = dx[i] - cx2, by = dy[i] - cy2;
        d1 = d1 + sqrt(ax * ax + ay * ay);
        d2 = d2 + sqrt(bx * bx + by * by);
    }
    const double s1 = sqrt(2.0) / (d1 * 0.25), s2 = sqrt(2.0) / (d2 * 0.25);

    // W = [A (8x9); V (9x9)] column pairs are rotated together.
    double W[17][9];
    for (int i = 0; i < 4; ++i) {
        const double x = (sx[i] - cx1) * s1, y = (sy[i] - cy1) * s1;
        const double u = (dx[i] - cx2) * s2, v = (dy[i] - cy2) * s2;
        double* r0 = W[2 * i];
        double* r1 = W[2 * i + 1];
        r0[0] = -x; r0[1] = -y; r0[2] = -1.0; r0[3] = 0.0; r0[4] = 0.0; r0[5] = 0.0;
        r0[6] = u * x; r0[7] = u * y; r0[8] = u;
        r1[0] = 0.0; r1[1] = 0.0; r1[2] = 0.0; r1[3] = -x; r1[4] = -y; r1[5] = -1.0;
        r1[6] = v * x; r1[7] = v * y; r1[8] = v;
    }
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) W[8 + i][j] = (i == j) ? 1.0 : 0.0;

    int sched[9][4][2];
    rr_schedule(sched);
    int sweeps = 0;
    for (; sweeps < 30; ++sweeps) {
        int rotated = 0;
        for (int r = 0; r < 9; ++r)
            for (int k = 0; k < 4; ++k) {
                const int p = sched[r][k][0], q = sched[r][k][1];
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int i = 0; i < 8; ++i) {
                    alpha = alpha + W[i][p] * W[i][p];
                    beta = beta + W[i][q] * W[i][q];
                    gamma = gamma + W[i][p] * W[i][q];
                }
                // rotate only if |gamma| > 1e-15*sqrt(alpha*beta) and neither column has
                // already vanished (norm < 1e-14; the data is Hartley-normalised so an
                // absolute floor is meaningful).  Written so that NaN never rotates.
                const bool rot = (gamma != 0.0) && (gamma * gamma > 1e-30 * (alpha * beta)) &&
                                 (alpha >= 1e-28) && (beta >= 1e-28);
                if (!rot) continue;
                ++rotated;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t);
                const double s = c * t;
                for (int i = 0; i < 17; ++i) {
                    const double wp = W[i][p], wq = W[i][q];
                    W[i][p] = c * wp - s * wq;
                    W[i][q] = s * wp + c * wq;
                }
            }
        if (!rotated) break;
    }
    if (sweeps_out) *sweeps_out = sweeps;
    // null vector = V column of the smallest A-column norm
    double nrm[9];
    int jm = 0;
    for (int j = 0; j < 9; ++j) {
        double a = 0.0;
        for (int i = 0; i < 8; ++i) a = a + W[i][j] * W[i][j];
        nrm[j] = a;
        if (a < nrm[jm]) jm = j;
    }
    if (witness) {
        double second = -1.0, largest = 0.0;
        for (int j = 0; j < 9; ++j) {
            if (nrm[j] > largest) largest = nrm[j];
            if (j != jm && (second < 0.0 || nrm[j] < second)) second = nrm[j];
        }
        *witness = (largest > 0.0) ? sqrt(second / largest) : 0.0;
    }
    double g[9];
    for (int j = 0; j < 9; ++j) g[j] = W[8 + j][jm];
    // H = T2^-1 * Hn * T1,  T1 = [s1 0 -s1*cx1; 0 s1 -s1*cy1; 0 0 1],
    //                       T2^-1 = [1/s2 0 cx2; 0 1/s2 cy2; 0 0 1]
    double A1[9];  // Hn * T1
    for (int r = 0; r < 3; ++r) {
        const double a = g[3 * r], b = g[3 * r + 1], c = g[3 * r + 2];
        A1[3 * r] = a * s1;
        A1[3 * r + 1] = b * s1;
        A1[3 * r + 2] = (c - (a * s1) * cx1) - (b * s1) * cy1;
    }
    const double is2 = 1.0 / s2;
    double Hh[9];
    for (int j = 0; j < 3; ++j) {
        Hh[j] = A1[j] * is2 + cx2 * A1[6 + j];
        Hh[3 + j] = A1[3 + j] * is2 + cy2 * A1[6 + j];
        Hh[6 + j] = A1[6 + j];
    }
    double fro = 0.0;
    for (int j = 0; j < 9; ++j) fro = fro + Hh[j] * Hh[j];
    double sc = 1.0 / sqrt(fro);
    if (Hh[8] < 0.0) sc = -sc;
    for (int j = 0; j < 9; ++j) H[j] = Hh[j] * sc;
}

MHO_API void mho_dlt4(const double* x1, const double* y1, const double* x2, const double* y2,
                      const int* idx /* M*4 */, int M, double* H /* M*9 */,
                      double* witness /* optional M */, int* sweeps /* optional M */)
{
    for (int m = 0; m < M; ++m) {
        double sx[4], sy[4], dx[4], dy[4];
        for (int k = 0; k < 4; ++k) {
            const int i = idx[4 * (size_t)m + k];
            sx[k] = x1[i]; sy[k] = y1[i]; dx[k] = x2[i]; dy[k] = y2[i];
        }
        dlt4(sx, sy, dx, dy, H + 9 * (size_t)m, witness ? witness + m : nullptr,
             sweeps ? sweeps + m : nullptr);
    }
}

// ---------------------------------------------------------------------------
// 8b. Per-point homographies and the mean shift over them (reference-style initialisation)
// ---------------------------------------------------------------------------
// ComputeLocalHomographies / GetHomographyHAF (M/MultiH.cpp:696-717, :850-911) + the 10-D feature of
// EstablishStablePointSets (:617-644).
MHO_API void mho_haf_point(const double* x1, const double* y1, const double* x2, const double* y2,
                           const double* aff, int N, const double* F, const double* e2, double locality,
                           double* H /* N*9 */, double* feat /* N*10 */)
{
    const double ex = e2[0], ey = e2[1];
    for (int n = 0; n < N; ++n) {
        double r[6][4];
        haf_rows(aff[4 * n], aff[4 * n + 1], aff[4 * n + 2], aff[4 * n + 3], x1[n], y1[n], x2[n], y2[n], F, ex, ey, r);
        double a[16], v[16], d[4];
        for (int i = 0; i < 4; ++i)
            for (int j = i; j < 4; ++j) {
                double s = r[0][i] * r[0][j];
                for (int q = 1; q < 6; ++q) s = s + r[q][i] * r[q][j];
                a[i * 4 + j] = s; a[j * 4 + i] = s;
            }
        jacobi_sym(4, a, v, d);
        int jm = 0;
        for (int j = 1; j < 4; ++j) if (d[j] < d[jm]) jm = j;
        const double h6 = v[0 * 4 + jm], h7 = v[1 * 4 + jm], h8 = v[2 * 4 + jm], lam = v[3 * 4 + jm];
        double h[9];
        h[6] = h6; h[7] = h7; h[8] = h8;
        h[3] = ey * h6 - lam * F[0]; h[4] = ey * h7 - lam * F[1]; h[5] = ey * h8 - lam * F[2];
        h[0] = ex * h6 + lam * F[3]; h[1] = ex * h7 + lam * F[4]; h[2] = ex * h8 + lam * F[5];
        const double inv = 1.0 / h[8];
        for (int q = 0; q < 9; ++q) h[q] = h[q] * inv;
        if (H) for (int q = 0; q < 9; ++q) H[9 * (size_t)n + q] = h[q];
        if (feat) {
            double* f = feat + 10 * (size_t)n;
            const double s1 = h[8], s2 = h[6] + h[8], s3 = h[7] + h[8];
            f[0] = h[2] / s1; f[1] = (h[0] + h[2]) / s2; f[2] = (h[1] + h[2]) / s3;
            f[3] = h[5] / s1; f[4] = (h[3] + h[5]) / s2; f[5] = (h[4] + h[5]) / s3;
            f[6] = x1[n] * locality; f[7] = y1[n] * locality; f[8] = x2[n] * locality; f[9] = y2[n] * locality;
        }
    }
}

// MeanShiftClustering<double>::Cluster (MeanShiftClustering.h:23-157) with the engine's counter RNG
// and the engine's summation order for the member sums (strided tree), so that the GPU climbs can be
// compared bit for bit.  Returns the number of modes; modes (k x d), assign (n).
static const int MHO_MS_BATCH = 256;   // climbs whose seeds are drawn together (the engine's MS_BATCH)
MHO_API int mho_mean_shift(const double* data, int n, int d, double bw, unsigned long long seed,
                           double* modes, int max_modes, int* assign)
{
    const double band_sq = bw * bw, stop = 1e-3 * bw;
    std::vector<int> init(n), visited(n, 0);
    for (int i = 0; i < n; ++i) init[i] = i;
    std::vector<std::vector<double>> cent;
    std::vector<std::vector<int>> votes;
    uint64_t counter = 0;
    auto l2 = [](const double* a, const double* b, int dd) {
        double s = 0.0;
        for (int j = 0; j < dd; ++j) { const double x = a[j] - b[j]; s = s + x * x; }
        return sqrt(s);
    };
    while (!init.empty()) {
        // the engine's seed rule: MHO_MS_BATCH seeds are drawn together from the rows unvisited now (:55-56 for each draw)
        // and climbed in draw order; a seed that an earlier climb of the batch has visited is dropped, as the reference
        // never starts from a visited row.  (Section 11's mean_shift_reference_order redraws after every climb.)
        const int climbs = (int)std::min<size_t>(MHO_MS_BATCH, init.size());
        std::vector<int> starts(climbs);
        for (int b = 0; b < climbs; ++b) {
            const double rnd = (double)(splitmix64(seed + counter++) >> 11) * (1.0 / 9007199254740992.0);
            starts[b] = init[(int)round(rnd * (double)(init.size() - 1))];
        }
        for (int b = 0; b < climbs; ++b) {
            const int st = starts[b];
            if (visited[st]) continue;
            std::vector<double> mean(data + (size_t)st * d, data + (size_t)(st + 1) * d);
            std::vector<int> my(n, 0);
            bool converged = false;
            for (int it = 0; it < 100000; ++it) {
                const std::vector<double> old = mean;
                // engine order: 64 groups x 256 strided lanes (row i -> lane i % 16384), binary tree
                // inside each group, groups added in sequence
                const int G = 64;
                std::vector<TreeAcc> acc(G, TreeAcc(d));
                int in = 0;
                for (int i = 0; i < n; ++i) {
                    double dist = 0.0;
                    for (int j = 0; j < d; ++j) { const double r = old[j] - data[(size_t)i * d + j]; dist += sqrt(r * r); }
                    if (dist < band_sq) {
                        const int lane = i % (G * MHO_W);
                        acc[lane / MHO_W].add(lane % MHO_W, data + (size_t)i * d);
                        ++in; ++my[i]; visited[i] = 1;
                    }
                }
                if (in == 0) break;
                std::vector<double> sum(d, 0.0), part(d);
                for (int g = 0; g < G; ++g) {
                    acc[g].finish(part.data());
                    for (int j = 0; j < d; ++j) sum[j] = sum[j] + part[j];
                }
                const double inv = 1.0 / (double)in;
                double move = 0.0;
                for (int j = 0; j < d; ++j) { mean[j] = sum[j] * inv; const double dd = mean[j] - old[j]; move = move + dd * dd; }
                if (sqrt(move) < stop) { converged = true; break; }
            }
            if (!converged) {
                visited[st] = 1;
            } else {
                int mw = -1;
                for (size_t cn = 0; cn < cent.size(); ++cn)
                    if (l2(mean.data(), cent[cn].data(), d) < bw / 2) { mw = (int)cn; break; }
                if (mw > -1) {
                    for (int j = 0; j < d; ++j) cent[mw][j] = 0.5 * (cent[mw][j] + mean[j]);
                    for (int i = 0; i < n; ++i) votes[mw][i] += my[i];
                } else {
                    cent.push_back(mean);
                    votes.push_back(my);
                }
            }
        }
        init.clear();
        for (int i = 0; i < n; ++i) if (!visited[i]) init.push_back(i);
    }
    std::vector<int> bv(n, 0);
    for (int i = 0; i < n; ++i) assign[i] = -1;
    for (size_t r = 0; r < votes.size(); ++r)
        for (int i = 0; i < n; ++i)
            if (bv[i] < votes[r][i]) { bv[i] = votes[r][i]; assign[i] = (int)r; }
    for (int c = 0; c < (int)cent.size() && c < max_modes; ++c)
        for (int j = 0; j < d; ++j) modes[(size_t)c * d + j] = cent[c][j];
    return (int)cent.size();
}

// ---------------------------------------------------------------------------
// 9b. Epipolar front half — own definition, "parity unpinned": the
//     reference calls cv::findFundamentalMat(RANSAC) (M/MultiH.cpp:775), OpenCV 3.1.0 calib3d.
// ---------------------------------------------------------------------------
// Two definitions of the distance to the epipolar geometry (the product's mh_set_fundamental_metric):
//   0  Sampson                       e^2 / (a^2 + b^2 + a'^2 + b'^2)
//   1  what cv::findFundamentalMat compares with its threshold (OpenCV 3.1.0 FMEstimatorCallback::computeError, published
//      source outside /root/reference, restated): the squared distance of each point to the epipolar line of the other,
//      the larger of the two.  The call sites: M/main.cpp:400 (2.0 px), M/MultiH.cpp:775 (threshold_fundamental_matrix).
static int g_fund_metric = 0;
MHO_API void mho_set_fundamental_metric(int metric) { g_fund_metric = metric; }
static inline double sampson_d(const double* f, double x, double y, double u, double v)
{
    const double a = f[0] * x + f[1] * y + f[2];
    const double b = f[3] * x + f[4] * y + f[5];
    const double c = f[6] * x + f[7] * y + f[8];
    const double a2 = f[0] * u + f[3] * v + f[6];
    const double b2 = f[1] * u + f[4] * v + f[7];
    const double e = u * a + v * b + c;
    if (g_fund_metric == 0) return (e * e) / (a * a + b * b + a2 * a2 + b2 * b2);
    const double d2 = (e * e) / (a * a + b * b);
    const double d1 = (e * e) / (a2 * a2 + b2 * b2);
    return d1 > d2 ? d1 : d2;
}

// null vector g (normalised frame) -> rank-2 F in pixel coordinates, unit Frobenius, F[8] >= 0
static void fund_finish(const double g[9], double cx1, double cy1, double s1, double cx2, double cy2,
                        double s2, double F[9])
{
    double Mm[9], V[9], D[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0.0;
            for (int k = 0; k < 3; ++k) a = a + g[3 * k + i] * g[3 * k + j];
            Mm[3 * i + j] = a;
        }
    jacobi_sym(3, Mm, V, D);
    int jm = 0;
    for (int j = 1; j < 3; ++j) if (D[j] < D[jm]) jm = j;
    const double v0 = V[0 * 3 + jm], v1 = V[1 * 3 + jm], v2 = V[2 * 3 + jm];
    double Fn[9];
    for (int r = 0; r < 3; ++r) {
        const double w = (g[3 * r] * v0 + g[3 * r + 1] * v1) + g[3 * r + 2] * v2;
        Fn[3 * r] = g[3 * r] - w * v0;
        Fn[3 * r + 1] = g[3 * r + 1] - w * v1;
        Fn[3 * r + 2] = g[3 * r + 2] - w * v2;
    }
    double B[9];
    for (int r = 0; r < 3; ++r) {
        const double a = Fn[3 * r] * s1, b = Fn[3 * r + 1] * s1;
        B[3 * r] = a; B[3 * r + 1] = b;
        B[3 * r + 2] = (Fn[3 * r + 2] - a * cx1) - b * cy1;
    }
    double Fh[9];
    const double tx = s2 * cx2, ty = s2 * cy2;
    for (int j = 0; j < 3; ++j) {
        Fh[j] = s2 * B[j];
        Fh[3 + j] = s2 * B[3 + j];
        Fh[6 + j] = (B[6 + j] - tx * B[j]) - ty * B[3 + j];
    }
    double fro = 0.0;
    for (int j = 0; j < 9; ++j) fro = fro + Fh[j] * Fh[j];
    double sc = 1.0 / sqrt(fro);
    if (Fh[8] < 0.0) sc = -sc;
    for (int j = 0; j < 9; ++j) F[j] = Fh[j] * sc;
}

// One-sided Jacobi null vector of W = [A (8x9); I9] in the round-robin order (shared with dlt4).
static int null9(double W[17][9], double g[9])
{
    int sched[9][4][2];
    rr_schedule(sched);
    int sweeps = 0;
    for (; sweeps < 30; ++sweeps) {
        int rotated = 0;
        for (int r = 0; r < 9; ++r)
            for (int k = 0; k < 4; ++k) {
                const int p = sched[r][k][0], q = sched[r][k][1];
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int i = 0; i < 8; ++i) {
                    alpha = alpha + W[i][p] * W[i][p];
                    beta = beta + W[i][q] * W[i][q];
                    gamma = gamma + W[i][p] * W[i][q];
                }
                const bool rot = (gamma != 0.0) && (gamma * gamma > 1e-30 * (alpha * beta)) &&
                                 (alpha >= 1e-28) && (beta >= 1e-28);
                if (!rot) continue;
                ++rotated;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t);
                const double s = c * t;
                for (int i = 0; i < 17; ++i) {
                    const double wp = W[i][p], wq = W[i][q];
                    W[i][p] = c * wp - s * wq;
                    W[i][q] = s * wp + c * wq;
                }
            }
        if (!rotated) break;
    }
    int jm = 0;
    double best = 0.0;
    for (int j = 0; j < 9; ++j) {
        double a = 0.0;
        for (int i = 0; i < 8; ++i) a = a + W[i][j] * W[i][j];
        if (j == 0 || a < best) { best = a; jm = j; }
    }
    for (int j = 0; j < 9; ++j) g[j] = W[8 + j][jm];
    return sweeps;
}

MHO_API void mho_sample8(unsigned long long seed, long long m0, int M, int N, int* idx /* M*8 */)
{
    for (int m = 0; m < M; ++m) sample_tuple(seed, (uint64_t)(m0 + m), (uint32_t)N, 8, 256, idx + 8 * (size_t)m);
}

MHO_API void mho_fund8(const double* x1, const double* y1, const double* x2, const double* y2,
                       const int* idx /* M*8 */, int M, double* F /* M*9 */)
{
    for (int m = 0; m < M; ++m) {
        double sx[8], sy[8], dx[8], dy[8];
        for (int k = 0; k < 8; ++k) {
            const int i = idx[8 * (size_t)m + k];
            sx[k] = x1[i]; sy[k] = y1[i]; dx[k] = x2[i]; dy[k] = y2[i];
        }
        double cx1 = sx[0], cy1 = sy[0], cx2 = dx[0], cy2 = dy[0];
        for (int k = 1; k < 8; ++k) { cx1 = cx1 + sx[k]; cy1 = cy1 + sy[k]; cx2 = cx2 + dx[k]; cy2 = cy2 + dy[k]; }
        cx1 = cx1 * 0.125; cy1 = cy1 * 0.125; cx2 = cx2 * 0.125; cy2 = cy2 * 0.125;
        double d1 = 0.0, d2 = 0.0;
        for (int k = 0; k < 8; ++k) {
            const double ax = sx[k] - cx1, ay = sy[k] - cy1, bx = dx[k] - cx2, by = dy[k] - cy2;
            d1 = d1 + sqrt(ax * ax + ay * ay);
            d2 = d2 + sqrt(bx * bx + by * by);
        }
        const double s1 = sqrt(2.0) / (d1 * 0.125), s2 = sqrt(2.0) / (d2 * 0.125);
        double W[17][9];
        for (int k = 0; k < 8; ++k) {
            const double x = (sx[k] - cx1) * s1, y = (sy[k] - cy1) * s1;
            const double u = (dx[k] - cx2) * s2, v = (dy[k] - cy2) * s2;
            W[k][0] = u * x; W[k][1] = u * y; W[k][2] = u; W[k][3] = v * x; W[k][4] = v * y; W[k][5] = v;
            W[k][6] = x; W[k][7] = y; W[k][8] = 1.0;
        }
        for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) W[8 + i][j] = (i == j) ? 1.0 : 0.0;
        double g[9];
        null9(W, g);
        fund_finish(g, cx1, cy1, s1, cx2, cy2, s2, F + 9 * (size_t)m);
    }
}

MHO_API void mho_sampson_score(const double* x1, const double* y1, const double* x2, const double* y2,
                               int N, const double* F, int M, double thr2, int* counts)
{
    for (int m = 0; m < M; ++m) {
        int c = 0;
        for (int n = 0; n < N; ++n) if (sampson_d(F + 9 * (size_t)m, x1[n], y1[n], x2[n], y2[n]) < thr2) ++c;
        counts[m] = c;
    }
}

MHO_API void mho_sampson(const double* x1, const double* y1, const double* x2, const double* y2, int N,
                         const double* F, double* d)
{
    for (int n = 0; n < N; ++n) d[n] = sampson_d(F, x1[n], y1[n], x2[n], y2[n]);
}

// LS 8-point refit on the Sampson inliers of F_in; sums in the strided-tree order.  Returns count.
MHO_API int mho_fund_refit(const double* x1, const double* y1, const double* x2, const double* y2, int N,
                           const double* F_in, double thr2, double* F_out, unsigned char* mask)
{
    TreeAcc c4(4);
    int count = 0;
    for (int n = 0; n < N; ++n) {
        const bool in = sampson_d(F_in, x1[n], y1[n], x2[n], y2[n]) < thr2;
        if (mask) mask[n] = in ? 1 : 0;
        if (in) { const double c[4] = { x1[n], y1[n], x2[n], y2[n] }; c4.add(n, c); ++count; }
    }
    if (count < 8) { for (int i = 0; i < 9; ++i) F_out[i] = F_in[i]; return count; }
    double s4[4];
    c4.finish(s4);
    const double inv = 1.0 / (double)count;
    const double cx1 = s4[0] * inv, cy1 = s4[1] * inv, cx2 = s4[2] * inv, cy2 = s4[3] * inv;
    TreeAcc c2(2);
    for (int n = 0; n < N; ++n)
        if (sampson_d(F_in, x1[n], y1[n], x2[n], y2[n]) < thr2) {
            const double ax = x1[n] - cx1, ay = y1[n] - cy1, bx = x2[n] - cx2, by = y2[n] - cy2;
            const double c[2] = { sqrt(ax * ax + ay * ay), sqrt(bx * bx + by * by) };
            c2.add(n, c);
        }
    double s2v[2];
    c2.finish(s2v);
    const double s1 = sqrt(2.0) / (s2v[0] / (double)count), s2 = sqrt(2.0) / (s2v[1] / (double)count);
    TreeAcc c45(45);
    for (int n = 0; n < N; ++n)
        if (sampson_d(F_in, x1[n], y1[n], x2[n], y2[n]) < thr2) {
            const double x = (x1[n] - cx1) * s1, y = (y1[n] - cy1) * s1, u = (x2[n] - cx2) * s2, v = (y2[n] - cy2) * s2;
            const double r[9] = { u * x, u * y, u, v * x, v * y, v, x, y, 1.0 };
            double c[45];
            int k = 0;
            for (int i = 0; i < 9; ++i) for (int j = i; j < 9; ++j) c[k++] = r[i] * r[j];
            c45.add(n, c);
        }
    double u45[45];
    c45.finish(u45);
    double A[81], V[81], D[9];
    int k = 0;
    for (int i = 0; i < 9; ++i) for (int j = i; j < 9; ++j) { A[i * 9 + j] = u45[k]; A[j * 9 + i] = u45[k]; ++k; }
    jacobi_sym(9, A, V, D);
    int jm = 0;
    for (int j = 1; j < 9; ++j) if (D[j] < D[jm]) jm = j;
    double g[9];
    for (int j = 0; j < 9; ++j) g[j] = V[j * 9 + jm];
    fund_finish(g, cx1, cy1, s1, cx2, cy2, s2, F_out);
    return count;
}

// ---------------------------------------------------------------------------
// 9c. Per-correspondence refinement of GetFundamentalMatrixAndRefineData (M/MultiH.cpp:807-838):
//     OptimalTriangulation (:1116-1188), GetAffineConsistency/GetBetaScale (:1057-1114),
//     GetOptimalAffineTransformation (:1190-1223).  cv::solvePoly (OpenCV, unpinned) is replaced by
//     Durand-Kerner with fixed start values; the 6x6 inverse of :1219 by its closed form.
// ---------------------------------------------------------------------------
namespace {
struct Cplx { double re, im; };
inline Cplx cmul(Cplx a, Cplx b) { return { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; }
inline Cplx csub(Cplx a, Cplx b) { return { a.re - b.re, a.im - b.im }; }
inline Cplx cdiv(Cplx a, Cplx b)
{
    const double den = b.re * b.re + b.im * b.im;
    return { (a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den };
}
void poly_roots(const double* c, int n, Cplx* z)
{
    double m[7];
    for (int k = 0; k <= n; ++k) m[k] = c[k] / c[n];
    Cplx seed = { 1.0, 0.0 };
    const Cplx base = { 0.4, 0.9 };
    for (int k = 0; k < n; ++k) { z[k] = seed; seed = cmul(seed, base); }
    for (int it = 0; it < 200; ++it) {
        double moved = 0.0;
        for (int k = 0; k < n; ++k) {
            Cplx p = { 1.0, 0.0 };
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
} // namespace

MHO_API void mho_poly_roots(const double* c, int n, double* re, double* im)
{
    Cplx z[6];
    poly_roots(c, n, z);
    for (int i = 0; i < n; ++i) { re[i] = z[i].re; im[i] = z[i].im; }
}

// reason (nullable, N): the stage a row left at — 0 kept, 1 not in the mask (:809), 2 OptimalTriangulation failed (:815-817),
// 3 distanceError > 1 (:826) — the product's mh_get_refine_reasons.
MHO_API void mho_refine_points_ex(const double* x1, const double* y1, const double* x2, const double* y2,
                                  const double* aff, int N, const double* F, const double* e1, const double* e2,
                                  const unsigned char* in_mask, unsigned char* keep, double* out /* N*8 */, unsigned char* reason)
{
    const double e1x = e1[0], e1y = e1[1], e2x = e2[0], e2y = e2[1];
    for (int n = 0; n < N; ++n) {
        keep[n] = 0;
        for (int q = 0; q < 8; ++q) out[8 * (size_t)n + q] = 0.0;
        if (reason) reason[n] = 1;
        if (in_mask && !in_mask[n]) continue;
        if (reason) reason[n] = 2;
        const double px = x1[n], py = y1[n], qx = x2[n], qy = y2[n];
        double G[9], F2[9], M2[9], F3[9];
        for (int r = 0; r < 3; ++r) {
            G[3 * r] = F[3 * r]; G[3 * r + 1] = F[3 * r + 1];
            G[3 * r + 2] = (F[3 * r] * px + F[3 * r + 1] * py) + F[3 * r + 2];
        }
        for (int c = 0; c < 3; ++c) { F2[c] = G[c]; F2[3 + c] = G[3 + c]; F2[6 + c] = (qx * G[c] + qy * G[3 + c]) + G[6 + c]; }
        for (int c = 0; c < 3; ++c) {
            M2[c] = (-e2x) * F2[c] + (-e2y) * F2[3 + c];
            M2[3 + c] = e2y * F2[c] + (-e2x) * F2[3 + c];
            M2[6 + c] = F2[6 + c];
        }
        for (int r = 0; r < 3; ++r) {
            F3[3 * r] = M2[3 * r] * e1x + M2[3 * r + 1] * e1y;
            F3[3 * r + 1] = M2[3 * r] * (-e1y) + M2[3 * r + 1] * e1x;
            F3[3 * r + 2] = M2[3 * r + 2];
        }
        const double f1 = 1.0, f2 = 1.0;
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
        double bestS = 2147483647.0, bestT = 0.0;
        if (deg > 0) {
            Cplx z[6];
            poly_roots(t, deg, z);
            for (int i = 0; i < deg; ++i)
                if (fabs(z[i].im) <= 1e-10) {
                    const double tt = z[i].re;
                    const double ct = c * tt + d, at = a * tt + b;
                    const double val = tt * tt / (1 + f12 * tt * tt) + (ct * ct) / (at * at + f22 * (ct * ct));
                    if (val < bestS) { bestS = val; bestT = tt; }
                }
        }
        const double valInf = 1 / f12 + (c * c) / (a * a + f22 * c * c);
        if (valInf < bestS) continue;
        if (reason) reason[n] = 3;
        const double l0 = F3[1] * bestT + F3[2], l1 = F3[4] * bestT + F3[5], l2 = F3[7] * bestT + F3[8];
        const double w2 = l0 * l0 + l1 * l1;
        const double iw = 1.0 / w2;
        const double p2x = (-l0 * l2) * iw, p2y = (-l1 * l2) * iw;
        const double s1 = e1x * e1x + e1y * e1y, s2 = e2x * e2x + e2y * e2y;
        const double ux = (e1x * 0.0 - e1y * bestT) / s1 + px;
        const double uy = (e1y * 0.0 + e1x * bestT) / s1 + py;
        const double vx = (-e2x * p2x + e2y * p2y) / s2 + qx;
        const double vy = (-e2y * p2x - e2x * p2y) / s2 + qy;
        const double A11 = aff[4 * n], A12 = aff[4 * n + 1], A21 = aff[4 * n + 2], A22 = aff[4 * n + 3];
        const double L1[3] = { (F[0] * vx + F[3] * vy) + F[6], (F[1] * vx + F[4] * vy) + F[7], (F[2] * vx + F[5] * vy) + F[8] };
        const double L2[3] = { (F[0] * ux + F[1] * uy) + F[2], (F[3] * ux + F[4] * uy) + F[5], (F[6] * ux + F[7] * uy) + F[8] };
        const double xn1 = ux + 1.0;
        const double yn1 = -(L1[0] * xn1 + L1[2]) / L1[1];
        double d1x = xn1 - ux, d1y = yn1 - uy;
        const double nd1 = sqrt(d1x * d1x + d1y * d1y);
        d1x = d1x / nd1; d1y = d1y / nd1;
        const double beta = fabs(sqrt(L2[0] * L2[0] + L2[1] * L2[1]) /
                                 ((-F[0] * d1y + F[1] * d1x) * vx + (-F[3] * d1y + F[4] * d1x) * vy - F[6] * d1y + F[7] * d1x));
        double n1x = L1[0] / L1[2], n1y = L1[1] / L1[2], n2x = L2[0] / L2[2], n2y = L2[1] / L2[2];
        const double nn1 = sqrt(n1x * n1x + n1y * n1y), nn2 = sqrt(n2x * n2x + n2y * n2y);
        n1x = n1x / nn1; n1y = n1y / nn1; n2x = n2x / nn2; n2y = n2y / nn2;
        const double det = A11 * A22 - A12 * A21;
        const double r1x = (A22 * n1x - A21 * n1y) / det, r1y = (-A12 * n1x + A11 * n1y) / det;
        const double ex_ = r1x - beta * n2x, ey_ = r1y - beta * n2y;
        const double distanceError = sqrt(ex_ * ex_ + ey_ * ey_);
        if (!(distanceError <= 1.0)) continue;
        if (n1x * n2x + n1y * n2y < 0) { n2x = -n2x; n2y = -n2y; }
        const double ppx = beta * n2x, ppy = beta * n2y, pp = ppx * ppx + ppy * ppy;
        const double lam1 = (n1x - (ppx * A11 + ppy * A21)) / pp;
        const double lam2 = (n1y - (ppx * A12 + ppy * A22)) / pp;
        double* o = out + 8 * (size_t)n;
        o[0] = ux; o[1] = uy; o[2] = vx; o[3] = vy;
        o[4] = A11 + ppx * lam1; o[5] = A12 + ppx * lam2; o[6] = A21 + ppy * lam1; o[7] = A22 + ppy * lam2;
        keep[n] = 1;
        if (reason) reason[n] = 0;
    }
}

MHO_API void mho_refine_points(const double* x1, const double* y1, const double* x2, const double* y2,
                               const double* aff, int N, const double* F, const double* e1, const double* e2,
                               const unsigned char* in_mask, unsigned char* keep, double* out /* N*8 */)
{
    mho_refine_points_ex(x1, y1, x2, y2, aff, N, F, e1, e2, in_mask, keep, out, nullptr);
}

// ---------------------------------------------------------------------------
// 10. LabelingStep and the alternating loop              
// ---------------------------------------------------------------------------
// LabelingStep, M/MultiH.cpp:513-602: data cost -> expansion (warm start iff
// !changed) -> labels shifted by -1 -> per-label HAF re-estimation.
// labeling: in = previous labels (-1..Nh-1, used only if warm), out = new.
MHO_API int mho_labeling_step(const double* x1, const double* y1, const double* x2,
                              const double* y2, const double* aff, int N,
                              double* H /* Nh*9 in/out */, int Nh, double lambda, double thr2,
                              const int* hit_rowptr, const int* hit_col, int warm,
                              const double* F, const double* e2, int* labeling, int* cycles_out)
{
    const int L = Nh + 1;
    std::vector<int> cost((size_t)N * L);
    mho_data_cost(x1, y1, x2, y2, N, H, Nh, lambda, thr2, cost.data());
    std::vector<int> lab(N, 0);
    if (warm) for (int i = 0; i < N; ++i) lab[i] = labeling[i] + 1;      // :525-529
    const int energy = mho_expand(N, L, cost.data(), hit_rowptr, hit_col, mho_potts(lambda),
                                  lab.data(), 1000, cycles_out, nullptr);
    for (int i = 0; i < N; ++i) labeling[i] = lab[i] - 1;                // :547-568
    mho_haf_reestimate(x1, y1, x2, y2, aff, N, labeling, Nh, F, e2, H, nullptr);
    return energy;
}

// ---------------------------------------------------------------------------
// 11. MergingStep and the whole merge <-> label alternation       
// ---------------------------------------------------------------------------
// Written from the reference text alone (M/MultiH.cpp:263-311, :352-471, :995-1055,
// MeanShiftClustering.h:23-157, Homography_Refine3PTCallback.h, Utilities.hpp:750-879) — NOT from the
// product's host code — so that Process() has an independent checker for labels, model count, iteration
// number and energy.  OpenCV primitives the reference calls and /root/reference does not contain
// (cv::eigen, Mat::inv, A.inv(DECOMP_SVD), cv::solve / cv::invert with DECOMP_EIG, cv::norm, gemm) are
// DEFINED here in the plainest form (sequential sums, closed-form inverse of a similarity, symmetric
// systems through the Jacobi solver of section 5): "parity unpinned" at those boundaries.  The product's
// 3-point solver may therefore differ from this one in the last bits of a homography; what must agree is
// every DECISION taken from them (inlier sets, kept modes, `changed`, labels, energies).
// Deviations shared with the product: explicit splitmix64 seeds instead of rand();
// a climb whose window captures no row ends (the reference would loop on a NaN mean).

// Feature vector of a homography: the images of (0,0), (1,0), (0,1).  M/MultiH.cpp:364-390.
static void homography_feature(const double* h, double* f)
{
    const double s1 = h[8];
    f[0] = h[2] / s1;
    f[1] = h[5] / s1;
    const double s2 = h[6] + h[8];
    f[2] = (h[0] + h[2]) / s2;
    f[3] = (h[3] + h[5]) / s2;
    const double s3 = h[7] + h[8];
    f[4] = (h[1] + h[2]) / s3;
    f[5] = (h[4] + h[5]) / s3;
}

// MeanShiftClustering<double>::Cluster, MeanShiftClustering.h:23-157, the reference's own summation order
// (members added one after the other, :85-96).  Returns the number of modes; modes: up to n x d.
static int mean_shift_reference_order(const double* data, int n, int d, double band_width, uint64_t seed,
                                      std::vector<double>& modes, uint64_t* draws_out)
{
    const double band_sq = band_width * band_width;                  // :31
    const double stop = 1e-3 * band_width;                           // :48
    std::vector<int> init(n), visited(n, 0);
    for (int i = 0; i < n; ++i) init[i] = i;
    modes.clear();
    int k = 0;
    uint64_t draws = 0;
    std::vector<double> mean(d), old(d), sum(d);
    while (!init.empty()) {                                          // :52
        const uint64_t z = splitmix64(seed + draws++);
        const double rnd = (double)(z >> 11) * (1.0 / 9007199254740992.0);
        const int st = init[(int)round(rnd * (double)(init.size() - 1))];      // :54-56
        for (int j = 0; j < d; ++j) mean[j] = data[(size_t)st * d + j];
        for (;;) {
            old = mean;
            for (int j = 0; j < d; ++j) sum[j] = 0.0;
            int members = 0;
            for (int i = 0; i < n; ++i) {
                double dist = 0.0;                                   // "sqDistToAll": an L1 norm, :78-83 (SURVEY A-8)
                for (int j = 0; j < d; ++j) {
                    const double r = old[j] - data[(size_t)i * d + j];
                    dist = dist + sqrt(r * r);
                }
                if (dist < band_sq) {                                // :85
                    for (int j = 0; j < d; ++j) sum[j] = sum[j] + data[(size_t)i * d + j];
                    visited[i] = 1;
                    ++members;
                }
            }
            if (members == 0) { visited[st] = 1; break; }            // deviation: no row captured -> the climb ends
            // :96 `myMean = myMean / inInds.size()` is cv::Mat / double, which OpenCV evaluates as a SCALE by the
            // reciprocal: operator/(const Mat&, double) builds MatOp_AddEx(a, alpha = 1./s) and the assignment is
            // a.convertTo(m, type, alpha), i.e. every element is src * (1/s) — not src / s (OpenCV 3.1.0 core/matop.cpp,
            // outside /root/reference: restated from the published source, unpinnable here).  Product and oracle both
            // multiply by the reciprocal (r04: until r03 this line divided).
            const double inv_members = 1.0 / (double)members;
            for (int j = 0; j < d; ++j) mean[j] = sum[j] * inv_members;
            double nrm = 0.0;
            for (int j = 0; j < d; ++j) { const double r = mean[j] - old[j]; nrm = nrm + r * r; }
            if (sqrt(nrm) < stop) {                                  // :98
                int merge_with = -1;
                for (int c = 0; c < k; ++c) {
                    double dd = 0.0;
                    for (int j = 0; j < d; ++j) { const double r = mean[j] - modes[(size_t)c * d + j]; dd = dd + r * r; }
                    if (sqrt(dd) < band_width / 2) { merge_with = c; break; }  // :101-109
                }
                if (merge_with > -1) {
                    for (int j = 0; j < d; ++j)
                        modes[(size_t)merge_with * d + j] = 0.5 * (modes[(size_t)merge_with * d + j] + mean[j]);   // :113
                } else {
                    modes.insert(modes.end(), mean.begin(), mean.end());
                    ++k;
                }
                break;
            }
        }
        init.clear();                                                // :125-130
        for (int i = 0; i < n; ++i) if (!visited[i]) init.push_back(i);
    }
    if (draws_out) *draws_out = draws;
    return k;
}

static void mat3_mul(const double* a, const double* b, double* c)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            c[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}

// NormalizePoints<double> for an N x 2 CV_64F matrix, Homography_Refine3PTCallback.h:164-199.
static void normalize_points(const double* pts, int n, std::vector<double>& out, double T[9])
{
    double mx = 0.0, my = 0.0;
    for (int i = 0; i < n; ++i) { mx = mx + pts[2 * i]; my = my + pts[2 * i + 1]; }
    const double inv_n = 1 / (double)n;
    mx = inv_n * mx; my = inv_n * my;                                // :171
    out.resize(2 * (size_t)n);
    double avg = 0.0;
    for (int i = 0; i < n; ++i) {
        const double x = pts[2 * i] - mx, y = pts[2 * i + 1] - my;
        out[2 * i] = x; out[2 * i + 1] = y;
        avg = avg + sqrt(x * x + y * y);                             // :178
    }
    avg = avg / n;
    const double ratio = sqrt(2.0) / avg;                            // :182
    for (int i = 0; i < 2 * n; ++i) out[i] = out[i] * ratio;
    T[0] = ratio; T[1] = 0; T[2] = -mx * ratio;                      // :191-196
    T[3] = 0; T[4] = ratio; T[5] = -my * ratio;
    T[6] = 0; T[7] = 0; T[8] = 1;
}

// inverse of T = [r 0 tx; 0 r ty; 0 0 1] (stands in for cv::Mat::inv, LU)
static void similarity_inverse(const double* T, double* Ti)
{
    const double ir = 1.0 / T[0];
    Ti[0] = ir; Ti[1] = 0; Ti[2] = -T[2] * ir;
    Ti[3] = 0; Ti[4] = ir; Ti[5] = -T[5] * ir;
    Ti[6] = 0; Ti[7] = 0; Ti[8] = 1;
}

// x = pinv(A) b for a symmetric n x n A through its eigen-decomposition, eigenvalues below the cut count as zero
// (what cv::solve / cv::invert do with DECOMP_EIG).  Ainv (nullable) receives the pseudo-inverse.
static void sym_solve_eig(int n, const double* A, const double* b, double* x, double* Ainv)
{
    std::vector<double> a(A, A + n * n), v(n * n), w(n);
    jacobi_sym(n, a.data(), v.data(), w.data());
    double cut = 0.0;
    for (int k = 0; k < n; ++k) cut = cut + fabs(w[k]);
    cut = cut * (2.0 * 2.220446049250313e-16);
    if (x) for (int i = 0; i < n; ++i) x[i] = 0.0;
    if (Ainv) for (int i = 0; i < n * n; ++i) Ainv[i] = 0.0;
    for (int k = 0; k < n; ++k) {
        if (fabs(w[k]) <= cut) continue;
        if (x) {
            double proj = 0.0;
            for (int i = 0; i < n; ++i) proj = proj + v[i * n + k] * b[i];
            proj = proj / w[k];
            for (int i = 0; i < n; ++i) x[i] = x[i] + proj * v[i * n + k];
        }
        if (Ainv)
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) Ainv[i * n + j] = Ainv[i * n + j] + v[i * n + k] * v[j * n + k] / w[k];
    }
}

struct Refine3PT {                      // Homography_Refine3PTCallback<double>, :62-145
    const double* src; const double* dst; int count;
    const double* F; double ex, ey;
    void compute(const double* h, double* err, double* J) const
    {
        for (int i = 0; i < count; ++i) {
            const double x1 = src[2 * i], y1 = src[2 * i + 1], x2 = dst[2 * i], y2 = dst[2 * i + 1];
            double s = h[0] * x1 + h[1] * y1 + h[2];
            s = fabs(s) > 2.220446049250313e-16 ? 1. / s : 0;        // :112
            const double h21 = ey * h[0] - F[0], h22 = ey * h[1] - F[1], h23 = ey * h[2] - F[2];
            const double h11 = ex * h[0] + F[3], h12 = ex * h[1] + F[4], h13 = ex * h[2] + F[5];
            const double xi = (h11 * x1 + h12 * y1 + h13) * s;
            const double yi = (h21 * x1 + h22 * y1 + h23) * s;
            err[2 * i] = x2 - xi;
            err[2 * i + 1] = y2 - yi;
            if (J) {
                double* j = J + 6 * i;
                j[0] = ex * s * x1; j[1] = ex * s * y1; j[2] = ex * s;
                j[3] = ey * s * x1; j[4] = ey * s * y1; j[5] = ey * s;
            }
        }
    }
};

// cv::LMSolverImpl::run (M/Utilities.hpp:762-869) for 3 parameters, maxIters = 1000, epsx = epsf = FLT_EPSILON.
static void lm_refine3(const Refine3PT& cb, double x[3])
{
    const int m = 2 * cb.count;
    const double eps = 2.220446049250313e-16, feps = 1.1920928955078125e-07;
    std::vector<double> r(m), rd(m), J(3 * (size_t)m);
    double A[9], v[3], D[3], xd[3], d[3], Ap[9];
    auto normal_eq = [&]() {                                          // mulTransposed(J, A, true); gemm(J, r, ..., GEMM_1_T)
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) {
                double s = 0.0;
                for (int i = 0; i < m; ++i) s = s + J[3 * i + a] * J[3 * i + b];
                A[3 * a + b] = s;
            }
            double s = 0.0;
            for (int i = 0; i < m; ++i) s = s + J[3 * i + a] * r[i];
            v[a] = s;
        }
    };
    auto sumsq = [&](const std::vector<double>& q) { double s = 0.0; for (int i = 0; i < m; ++i) s = s + q[i] * q[i]; return s; };
    cb.compute(x, r.data(), J.data());
    double S = sumsq(r);
    normal_eq();
    for (int i = 0; i < 3; ++i) D[i] = A[4 * i];                     // :783, taken once
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        for (int i = 0; i < 9; ++i) Ap[i] = A[i];
        for (int i = 0; i < 3; ++i) Ap[4 * i] += lambda * D[i];
        sym_solve_eig(3, Ap, v, d, nullptr);                         // solve(Ap, v, d, DECOMP_EIG)
        for (int i = 0; i < 3; ++i) xd[i] = x[i] - d[i];
        cb.compute(xd, rd.data(), nullptr);
        const double Sd = sumsq(rd);
        double dS = 0.0, t = 0.0;
        for (int i = 0; i < 3; ++i) {
            const double Ad = (A[3 * i] * d[0] + A[3 * i + 1] * d[1]) + A[3 * i + 2] * d[2];
            dS = dS + d[i] * (-Ad + 2 * v[i]);                       // gemm(A, d, -1, v, 2, temp_d); d.dot(temp_d)
            t = t + d[i] * v[i];
        }
        const double R = (S - Sd) / (fabs(dS) > eps ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            double nu = (Sd - S) / (fabs(t) > eps ? t : 1) + 2;
            nu = std::min(std::max(nu, 2.), 10.);
            if (lambda == 0) {
                double Ai[9];
                sym_solve_eig(3, A, nullptr, nullptr, Ai);           // invert(A, Ap, DECOMP_EIG)
                double maxval = eps;
                for (int i = 0; i < 3; ++i) maxval = std::max(maxval, fabs(Ai[4 * i]));
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            for (int i = 0; i < 3; ++i) x[i] = xd[i];
            cb.compute(x, r.data(), J.data());
            normal_eq();
        }
        ++iter;
        double dinf = 0.0, rinf = 0.0;
        for (int i = 0; i < 3; ++i) dinf = std::max(dinf, fabs(d[i]));
        for (int i = 0; i < m; ++i) rinf = std::max(rinf, fabs(r[i]));
        if (!(iter < 1000 && dinf >= feps && rinf >= feps)) break;
    }
}

// GetHomography3PT (M/MultiH.cpp:995-1055) with RefineHomography3PT (Homography_Refine3PTCallback.h:7-58).
static bool homography_3pt(const double* p1, const double* p2, int n, const double* Fund, double* H, bool refine)
{
    std::vector<double> n1, n2;
    double T1[9], T2[9], T1i[9], T2i[9];
    normalize_points(p1, n, n1, T1);
    normalize_points(p2, n, n2, T2);
    similarity_inverse(T1, T1i);
    similarity_inverse(T2, T2i);
    double T2it[9], tmp[9], Fn[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T2it[3 * i + j] = T2i[3 * j + i];
    mat3_mul(T2it, Fund, tmp);
    mat3_mul(tmp, T1i, Fn);                                          // :1010
    double FFt[9], V[9], W[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            FFt[3 * i + j] = (Fn[3 * i] * Fn[3 * j] + Fn[3 * i + 1] * Fn[3 * j + 1]) + Fn[3 * i + 2] * Fn[3 * j + 2];
    jacobi_sym(3, FFt, V, W);
    int jm = 0;
    for (int j = 1; j < 3; ++j) if (W[j] < W[jm]) jm = j;            // last row of cv::eigen's descending order
    const double ex = V[0 * 3 + jm] / V[2 * 3 + jm], ey = V[1 * 3 + jm] / V[2 * 3 + jm];   // :1018
    std::vector<double> A(6 * (size_t)n), b(2 * (size_t)n);
    for (int i = 0; i < n; ++i) {                                    // :1025-1037
        const double x1 = n1[2 * i], y1 = n1[2 * i + 1], x2 = n2[2 * i], y2 = n2[2 * i + 1];
        double* a = &A[6 * (size_t)i];
        a[0] = ex * x1 - x2 * x1; a[1] = ex * y1 - x2 * y1; a[2] = ex - x2;
        a[3] = ey * x1 - y2 * x1; a[4] = ey * y1 - y2 * y1; a[5] = ey - y2;
        b[2 * i] = -(x1 * Fn[3] + y1 * Fn[4] + Fn[5]);
        b[2 * i + 1] = (x1 * Fn[0] + y1 * Fn[1] + Fn[2]);
    }
    // res = A.inv(DECOMP_SVD) * b: the least-squares solution, here through the normal equations
    double AtA[9], Atb[3], h3[3];
    for (int a = 0; a < 3; ++a) {
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int i = 0; i < 2 * n; ++i) s = s + A[3 * (size_t)i + a] * A[3 * (size_t)i + c];
            AtA[3 * a + c] = s;
        }
        double s = 0.0;
        for (int i = 0; i < 2 * n; ++i) s = s + A[3 * (size_t)i + a] * b[i];
        Atb[a] = s;
    }
    sym_solve_eig(3, AtA, Atb, h3, nullptr);
    if (refine) {
        Refine3PT cb{ n1.data(), n2.data(), n, Fn, ex, ey };
        lm_refine3(cb, h3);
    }
    double Hn[9];
    Hn[6] = h3[0]; Hn[7] = h3[1]; Hn[8] = h3[2];
    Hn[3] = ey * h3[0] - Fn[0]; Hn[4] = ey * h3[1] - Fn[1]; Hn[5] = ey * h3[2] - Fn[2];
    Hn[0] = ex * h3[0] + Fn[3]; Hn[1] = ex * h3[1] + Fn[4]; Hn[2] = ex * h3[2] + Fn[5];
    mat3_mul(T2i, Hn, tmp);
    mat3_mul(tmp, T1, H);                                            // :1054
    for (int i = 0; i < 9; ++i) if (!std::isfinite(H[i])) return false;
    return true;
}

MHO_API int mho_homography_3pt(const double* p1, const double* p2, int n, const double* F, double* H, int refine)
{
    return homography_3pt(p1, p2, n, F, H, refine != 0) ? 1 : 0;
}

// MergingStep, M/MultiH.cpp:352-471.  H: Nh*9 in; kept: capacity >= Nh*9 (the modes that survive).  Returns the number
// of kept candidates; *changed = (that number != Nh).  The caller replaces its model set only when changed (:468-470).
// The host half of MergingStep (M/MultiH.cpp:352-428) for tests: features (:364-390), modes (:394-397), one 3-point
// homography per mode (:408-427).  feat: Nh x 6; modes: up to Nh x 6, *n_modes of them; cand: up to Nh x 9 (9 per mode
// whose fit succeeded), cand_mode: the mode each came from.  Returns the number of candidates.
MHO_API int mho_merge_candidates(const double* H, int Nh, const double* F, double thr_h, uint64_t seed, double* feat,
                                 double* modes_out, int* n_modes, double* cand, int* cand_mode, uint64_t* draws)
{
    std::vector<double> modes;
    for (int i = 0; i < Nh; ++i) homography_feature(H + 9 * (size_t)i, feat + 6 * (size_t)i);
    const int k = mean_shift_reference_order(feat, Nh, 6, thr_h, seed, modes, draws);            // :394-397
    for (size_t q = 0; q < modes.size(); ++q) modes_out[q] = modes[q];
    *n_modes = k;
    const double pts1[6] = { 0, 0, 1, 0, 0, 1 };                                                // :408
    int nc = 0;
    for (int i = 0; i < k; ++i) {
        double Hc[9];
        if (!homography_3pt(pts1, &modes[6 * (size_t)i], 3, F, Hc, true)) continue;             // :427
        for (int q = 0; q < 9; ++q) cand[9 * (size_t)nc + q] = Hc[q];
        cand_mode[nc++] = i;
    }
    return nc;
}

// MergingStep, M/MultiH.cpp:352-471.  H: Nh*9 in; kept: capacity >= Nh*9 (the modes that survive).  Returns the number
// of kept candidates; *changed = (that number != Nh).  The caller replaces its model set only when changed (:468-470).
MHO_API int mho_merging_step(const double* x1, const double* y1, const double* x2, const double* y2, int N,
                             const double* H, int Nh, const double* F, double thr_h, double straightness,
                             uint64_t seed, double* kept /* up to Nh*9 */, int* changed, uint64_t* draws)
{
    std::vector<double> feat(6 * (size_t)Nh), modes(6 * (size_t)Nh), cand(9 * (size_t)Nh);
    std::vector<int> cand_mode(Nh);
    int k = 0;
    const int nc = mho_merge_candidates(H, Nh, F, thr_h, seed, feat.data(), modes.data(), &k, cand.data(), cand_mode.data(), draws);
    int nk = 0;
    for (int i = 0; i < nc; ++i) {
        const double* Hc = &cand[9 * (size_t)i];
        double mom[6], mineig = 0.0;
        mho_inlier_moments(x1, y1, x2, y2, N, Hc, 1, thr_h * thr_h, mom, &mineig);              // :430-461
        if (mineig < straightness || (int)mom[0] < 3) continue;                                 // :462
        for (int q = 0; q < 9; ++q) kept[9 * (size_t)nk + q] = Hc[q];
        ++nk;
    }
    *changed = nk != Nh;                                                                        // :468
    return nk;
}

typedef int (*mho_expand_hook)(int N, int L, const int* cost, const int* hit_rowptr, const int* hit_col, int potts,
                               const int* init_labels, int* labels_out);

// ClusterMergingAndLabeling's loop, M/MultiH.cpp:263-311, started from `Nh` initial models (what
// EstablishStablePointSets hands over).  H: capacity max_models*9 in/out; labeling: N out.  expand (nullable) replaces
// the oracle's own alpha-expansion, e.g. by the reference's GCoptimization compiled unmodified (oracle/_ref).
// seed_of_step(c) = seed ^ 0x4d53 ^ (c << 20) for the c-th MergingStep, the product's convention.
// Returns the number of models; *iterations = final_iteration_number (:311), *energy = final_energy (0 unless the
// loop converged, :297).
// The product's SetFixedIterations (north_star configs[4]: "20 propose-expand iterations" instead of the convergence
// test): with n > 0 the loop below also stops after its n-th LabelingStep.  0 = the reference's stop rule alone.
static int g_fixed_iterations = 0;
MHO_API void mho_set_fixed_iterations(int n) { g_fixed_iterations = n; }
// The product's MultiH::SetProposalRefit (default on since r05): mho_process's DLT route selects with refitted winners.
static int g_select_refit = 1;
MHO_API void mho_set_select_refit(int on) { g_select_refit = on; }

MHO_API int mho_cluster_merging_and_labeling(const double* x1, const double* y1, const double* x2, const double* y2,
                                             const double* aff, int N, double* H, int Nh, int max_models,
                                             const double* F, const double* e2, double lambda, double thr_h,
                                             double straightness, const int* hit_rowptr, const int* hit_col,
                                             uint64_t seed, mho_expand_hook expand, int* labeling, int* iterations,
                                             double* energy_out)
{
    const double thr2 = thr_h * thr_h;
    std::vector<double> models(H, H + 9 * (size_t)Nh), kept;
    for (int i = 0; i < N; ++i) labeling[i] = -1;                    // :263
    double last_energy = 2147483647.0, final_energy = 0.0;           // INT_MAX, :264
    int not_changed = 0, iteration = 0;
    uint64_t step = 0;
    while (iteration++ < 500) {                                      // :267
        int nh = (int)(models.size() / 9);
        int changed = 0;
        if (nh > 0) {
            kept.assign(models.size(), 0.0);
            const int nk = mho_merging_step(x1, y1, x2, y2, N, models.data(), nh, F, thr_h, straightness,
                                            seed ^ 0x4d53u ^ (step << 20), kept.data(), &changed, nullptr);
            ++step;
            if (changed) models.assign(kept.begin(), kept.begin() + 9 * (size_t)nk);          // :469-470
        }
        if (changed) not_changed = 0; else ++not_changed;           // :275-278
        nh = (int)(models.size() / 9);
        if (nh == 1) {                                               // :280-285
            for (int i = 0; i < N; ++i)
                if (fwd_d2(models.data(), x1[i], y1[i], x2[i], y2[i]) < thr2) labeling[i] = 0;   // :743-768
            break;
        } else if (nh == 0)
            break;
        // LabelingStep, :513-602
        const int L = nh + 1;
        std::vector<int> cost((size_t)N * L), lab(N, 0);
        mho_data_cost(x1, y1, x2, y2, N, models.data(), nh, lambda, thr2, cost.data());
        if (!changed) for (int i = 0; i < N; ++i) lab[i] = labeling[i] + 1;                  // :525-529
        int energy_i;
        if (expand) {
            std::vector<int> out(N);
            energy_i = expand(N, L, cost.data(), hit_rowptr, hit_col, mho_potts(lambda), changed ? nullptr : lab.data(), out.data());
            lab = out;
        } else {
            energy_i = mho_expand(N, L, cost.data(), hit_rowptr, hit_col, mho_potts(lambda), lab.data(), 1000, nullptr, nullptr);
        }
        for (int i = 0; i < N; ++i) labeling[i] = lab[i] - 1;                                 // :547-568
        mho_haf_reestimate(x1, y1, x2, y2, aff, N, labeling, nh, F, e2, models.data(), nullptr);
        const double energy = (double)energy_i;
        if ((!changed && fabs(last_energy - energy) < 1e-5) || not_changed > 10 ||            // :295
            (g_fixed_iterations > 0 && iteration >= g_fixed_iterations)) {
            final_energy = energy;
            break;
        }
        last_energy = energy;
    }
    const int nh = (int)(models.size() / 9);
    for (int i = 0; i < nh && i < max_models; ++i)
        for (int q = 0; q < 9; ++q) H[9 * (size_t)i + q] = models[9 * (size_t)i + q];
    if (iterations) *iterations = iteration - 1;                     // :311
    if (energy_out) *energy_out = final_energy;
    return nh;
}

// ---------------------------------------------------------------------------
// 12. Around the loop: stable point sets, propose + greedy selection, post-filter, Process()   
// ---------------------------------------------------------------------------
// Process() of the reference (M/MultiH.cpp:42-98) from "F is known" on:
//     ComputeLocalHomographies + EstablishStablePointSets (:696-717, :604-694)   -> mho_establish_stable_point_sets
//     ClusterMergingAndLabeling (:224-312)                                       -> section 11
//     HomographyCompatibilityCheck when more than one cluster is left (:78-86, :100-222) -> mho_compatibility_check
//     at most one cluster left: labels reset, HandleDegenerateCase (:88-94, :719-741)   -> mho_handle_degenerate
// north_star's propose route (random 4-tuples -> DLT -> take the best-supported hypothesis, take its inliers out of the
// support set, score again; the scheme of the dead M/MultipleHomographies.h:146-175) has no live reference code:
// mho_select_greedy is its definition.  rand() is replaced by the splitmix64 counters of the product everywhere.

// Sequential best-first selection over a fixed hypothesis list.  Per round: inlier counts over the points still in the
// support set (strict d2 < thr2, the score of :430-443); the best count wins, the LOWEST hypothesis index on ties; stop
// when the best count is below `need`; the winner's inliers leave the support set.  mask: in/out (1 = in the set).
struct SelectRefit { const double* aff; const double* F; const double* e2; };      // non-null: refit every winner (see below)
static int select_greedy_impl(const double* x1, const double* y1, const double* x2, const double* y2, int N,
                              const double* H, int M, double thr2, int need, int max_models, unsigned char* mask,
                              double* H_out, long long* index_out, int* counts_out, bool symmetric, const SelectRefit* refit = nullptr);

MHO_API int mho_select_greedy(const double* x1, const double* y1, const double* x2, const double* y2, int N,
                              const double* H, int M, double thr2, int need, int max_models, unsigned char* mask,
                              double* H_out /* max_models*9 */, long long* index_out, int* counts_out)
{
    return select_greedy_impl(x1, y1, x2, y2, N, H, M, thr2, need, max_models, mask, H_out, index_out, counts_out, false);
}

// ... on the symmetric transfer error (scores and claims both)
MHO_API int mho_select_greedy_sym(const double* x1, const double* y1, const double* x2, const double* y2, int N,
                                  const double* H, int M, double thr2, int need, int max_models, unsigned char* mask,
                                  double* H_out /* max_models*9 */, long long* index_out, int* counts_out)
{
    return select_greedy_impl(x1, y1, x2, y2, N, H, M, thr2, need, max_models, mask, H_out, index_out, counts_out, true);
}

// ... with every round's winner REFITTED to the points of the support set it explains before it claims them (the product's
// mh_set_tuning key 30 / MultiH::SetProposalRefit, the default of Process()'s DLT route since r05): the per-label HAF least
// squares of the loop (mho_haf_reestimate, M/MultiH.cpp:913-989) with one label; the refit takes the hypothesis' place when
// it is finite and explains at least as many points of the support set.
MHO_API int mho_select_greedy_refit(const double* x1, const double* y1, const double* x2, const double* y2, const double* aff, int N,
                                    const double* F, const double* e2, const double* H, int M, double thr2, int need, int max_models,
                                    unsigned char* mask, double* H_out /* max_models*9 */, long long* index_out, int* counts_out)
{
    const SelectRefit r{ aff, F, e2 };
    return select_greedy_impl(x1, y1, x2, y2, N, H, M, thr2, need, max_models, mask, H_out, index_out, counts_out, false, &r);
}

static int select_greedy_impl(const double* x1, const double* y1, const double* x2, const double* y2, int N,
                              const double* H, int M, double thr2, int need, int max_models, unsigned char* mask,
                              double* H_out, long long* index_out, int* counts_out, bool symmetric, const SelectRefit* refit)
{
    auto score = symmetric ? mho_score_sym : mho_score;
    int selected = 0;
    std::vector<int> counts(M);
    std::vector<unsigned char> taken(M, 0);
    for (int round = 0; round < max_models; ++round) {
        // the score of :430-443 for every hypothesis over the support set; the hypotheses are independent, so a large
        // batch is spread over the host's cores (integer counts: the same whatever the number of threads) — at BASELINE
        // configs[4] size, 100 000 x 50 000 pairs per round, one core would need 8.5 s per round
        if ((long long)M * N >= 100000000ll) {
#pragma omp parallel for schedule(static)
            for (int m = 0; m < M; ++m) score(x1, y1, x2, y2, N, H + 9 * (size_t)m, 1, thr2, mask, &counts[m]);
        } else
            score(x1, y1, x2, y2, N, H, M, thr2, mask, counts.data());
        // a hypothesis is selected at most once (without the refit a winner is left with no inlier in the support set, so this only
        // matters with it: the refit's claim can leave some of the hypothesis' own inliers behind)
        int best = -1, bm = -1;
        for (int m = 0; m < M; ++m) if (!taken[m] && counts[m] > best) { best = counts[m]; bm = m; }
        if (bm < 0 || best < need) break;
        taken[bm] = 1;
        const double* h = H + 9 * (size_t)bm;
        double hr[9];
        if (refit) {
            auto d2_of = [&](const double* g, int i) {
                if (!symmetric) return fwd_d2(g, x1[i], y1[i], x2[i], y2[i]);
                double ga[9];
                adjugate(g, ga);
                return fwd_d2(g, x1[i], y1[i], x2[i], y2[i]) + fwd_d2(ga, x2[i], y2[i], x1[i], y1[i]);
            };
            std::vector<int> lab(N, -1);
            for (int i = 0; i < N; ++i) if (mask[i] && d2_of(h, i) < thr2) lab[i] = 0;
            for (int q = 0; q < 9; ++q) hr[q] = h[q];
            mho_haf_reestimate(x1, y1, x2, y2, refit->aff, N, lab.data(), 1, refit->F, refit->e2, hr, nullptr);
            bool finite = true;
            for (int q = 0; q < 9; ++q) finite = finite && std::fabs(hr[q]) < 0x1p1000;
            int c = 0;
            if (finite) for (int i = 0; i < N; ++i) if (mask[i] && d2_of(hr, i) < thr2) ++c;
            if (finite && c >= best) h = hr;
        }
        for (int q = 0; q < 9; ++q) H_out[9 * (size_t)selected + q] = h[q];
        if (index_out) index_out[selected] = bm;
        if (counts_out) counts_out[selected] = best;
        ++selected;
        double a[9];
        adjugate(h, a);
        for (int i = 0; i < N; ++i) {
            if (!mask[i]) continue;
            const double d2 = symmetric ? fwd_d2(h, x1[i], y1[i], x2[i], y2[i]) + fwd_d2(a, x2[i], y2[i], x1[i], y1[i])
                                        : fwd_d2(h, x1[i], y1[i], x2[i], y2[i]);
            if (d2 < thr2) mask[i] = 0;
        }
    }
    return selected;
}

// ComputeLocalHomographies (:696-717) + EstablishStablePointSets (:604-694): per-point HAF homographies and their 10-D
// features (mho_haf_point), mean shift with band width thr_h (the engine-order restatement mho_mean_shift, see its
// header and HISTORY.md 3.8 for what that order defines), one LM-refined 3-point homography per cluster of >= 3 points
// (:664-688).  A per-point solve that degenerates leaves non-finite features; the product parks such rows at 1e300 so
// that the L1 ball test never sees a NaN, and so does this.  Returns the number of models (H_out: capacity max_models*9).
MHO_API int mho_establish_stable_point_sets(const double* x1, const double* y1, const double* x2, const double* y2,
                                            const double* aff, int N, const double* F, const double* e2, double locality,
                                            double thr_h, uint64_t ms_seed, double* H_out, int max_models)
{
    std::vector<double> feat(10 * (size_t)N);
    mho_haf_point(x1, y1, x2, y2, aff, N, F, e2, locality, nullptr, feat.data());
    for (double& f : feat) if (!std::isfinite(f)) f = 1e300;
    std::vector<int> assign(N);
    std::vector<double> modes(10 * (size_t)N);
    const int k = mho_mean_shift(feat.data(), N, 10, thr_h, ms_seed, modes.data(), N, assign.data());
    std::vector<std::vector<int>> members(k);
    for (int i = 0; i < N; ++i) if (assign[i] >= 0) members[assign[i]].push_back(i);       // MeanShiftClustering.h:148-156
    int nh = 0;
    for (int c = 0; c < k; ++c) {
        const int ni = (int)members[c].size();
        if (ni < 3) continue;                                                               // :667
        std::vector<double> p1(2 * (size_t)ni), p2(2 * (size_t)ni);
        for (int j = 0; j < ni; ++j) {
            const int i = members[c][j];
            p1[2 * j] = x1[i]; p1[2 * j + 1] = y1[i]; p2[2 * j] = x2[i]; p2[2 * j + 1] = y2[i];
        }
        double Hc[9];
        if (!homography_3pt(p1.data(), p2.data(), ni, F, Hc, true)) continue;               // :685 (a non-finite fit is dropped)
        if (nh < max_models) for (int q = 0; q < 9; ++q) H_out[9 * (size_t)nh + q] = Hc[q];
        ++nh;
    }
    return nh;
}

// HomographyCompatibilityCheck, M/MultiH.cpp:100-222, the loop as written: per cluster with at least
// max(min_inliers, 4) points, 501 trials (:128, SURVEY A-10) of { erase three randomly indexed points from the cluster's
// vectors (:138-151), 3-point homography without refinement (:154), squared transfer error of the REMAINING points into
// the first N-3 entries of an N-entry buffer (:158-173), sort all N entries — the last three are leftovers of the
// previous trial, zeros in the first (:136,:175) — "median" = element rest/2 or the mean of elements rest/2 and
// rest/2 + 1 (:176), re-append the three points at positions N-1, N-2, N-3 (:178-189) }; the cluster is removed when the
// same kind of median over the 501 trial values exceeds thr^2 * 81/16 (:192-195) or when it has fewer than min_inliers
// points (:199-200); labels are compacted from the last cluster down (:205-221).
// rand() (:142): draw number c is u = (splitmix64(seed + c) >> 11) * 2^-53, one counter running through the clusters in
// order.  A trial whose fit has a non-finite entry counts every distance as 1e300, and so does a NaN distance
// (std::sort on NaN is undefined in the reference; this fixes an order).  medians (nullable): per cluster, NaN = not tested.
MHO_API int mho_compatibility_check(const double* src_xy, const double* dst_xy, int n, int* labels, double* H, int nh,
                                    const double* F, double sqr_thr, int min_inliers, uint64_t seed, double* medians)
{
    std::vector<std::vector<double>> src(nh), dst(nh);                           // :102-115
    for (int i = 0; i < n; ++i) {
        const int l = labels[i];
        if (l > -1 && l < nh) {
            src[l].push_back(src_xy[2 * i]); src[l].push_back(src_xy[2 * i + 1]);
            dst[l].push_back(dst_xy[2 * i]); dst[l].push_back(dst_xy[2 * i + 1]);
        }
    }
    std::vector<char> remove(nh, 0);
    uint64_t counter = 0;
    for (int c = 0; c < nh; ++c) {
        std::vector<double>& sp = src[c];
        std::vector<double>& dp = dst[c];
        const int N = (int)(sp.size() / 2);
        const int trials = 501;                                                  // MAX(501, MIN(501, ...)), :128
        if (medians) medians[c] = std::nan("");
        if (N >= std::max(min_inliers, 4)) {                                     // :134
            std::vector<double> distances(trials), dist((size_t)N, 0.0);        // :132, :136
            for (int t = 0; t < trials; ++t) {
                double ms[6], md[6];
                for (int j = 0; j < 3; ++j) {                                    // :139-151
                    const double u = (double)(splitmix64(seed + counter++) >> 11) * (1.0 / 9007199254740992.0);
                    const int cur = (int)(sp.size() / 2);
                    const int idx = (int)((cur - 1) * u);                        // :142
                    ms[2 * j] = sp[2 * idx]; ms[2 * j + 1] = sp[2 * idx + 1];
                    md[2 * j] = dp[2 * idx]; md[2 * j + 1] = dp[2 * idx + 1];
                    sp.erase(sp.begin() + 2 * idx, sp.begin() + 2 * idx + 2);
                    dp.erase(dp.begin() + 2 * idx, dp.begin() + 2 * idx + 2);
                }
                double Hc[9];
                const bool ok = homography_3pt(ms, md, 3, F, Hc, false);         // :154
                const int rest = (int)(sp.size() / 2);
                for (int j = 0; j < rest; ++j) {                                 // :158-173
                    double d2 = std::nan("");
                    if (ok) d2 = fwd_d2(Hc, sp[2 * j], sp[2 * j + 1], dp[2 * j], dp[2 * j + 1]);
                    dist[j] = std::isnan(d2) ? 1e300 : d2;
                }
                std::sort(dist.begin(), dist.end());                             // :175 — all N entries
                distances[t] = rest % 2 ? dist[rest / 2] : 0.5 * (dist[rest / 2] + dist[rest / 2 + 1]);   // :176
                sp.resize(2 * (size_t)N);                                        // :178-189
                dp.resize(2 * (size_t)N);
                for (int j = 0; j < 3; ++j) {
                    const int p2 = N - j - 1;
                    sp[2 * p2] = ms[2 * j]; sp[2 * p2 + 1] = ms[2 * j + 1];
                    dp[2 * p2] = md[2 * j]; dp[2 * p2 + 1] = md[2 * j + 1];
                }
            }
            std::sort(distances.begin(), distances.end());                       // :192
            const double median = trials % 2 ? distances[trials / 2] : 0.5 * (distances[trials / 2] + distances[trials / 2 + 1]);
            if (medians) medians[c] = median;
            remove[c] = median > sqr_thr * 81.0 / 16.0;                          // :195
        } else if (N < min_inliers)
            remove[c] = 1;                                                       // :199-200
    }
    int kept = nh;
    for (int c = nh - 1; c >= 0; --c) {                                          // :205-221
        if (!remove[c]) continue;
        for (int j = 0; j < n; ++j) {
            if (labels[j] == c) labels[j] = -1;
            else if (labels[j] > c) --labels[j];
        }
        for (int q = c; q + 1 < kept; ++q) memcpy(H + 9 * (size_t)q, H + 9 * (size_t)(q + 1), 9 * sizeof(double));
        --kept;
    }
    return kept;
}

// HandleDegenerateCase, M/MultiH.cpp:719-741: cv::findHomography(RANSAC) on the correspondences, its inliers get label
// 0, everything else -1, one model.  cv::findHomography is not under /root/reference ("parity unpinned"); the build's
// definition, here as in the product: the best-supported (first maximum) of max(hypotheses, 1000) 4-point DLT
// hypotheses drawn with counters 0.. of `seed`.
MHO_API void mho_handle_degenerate(const double* x1, const double* y1, const double* x2, const double* y2, int N,
                                   double thr2, uint64_t seed, int hypotheses, int* labeling, double* H_out)
{
    const int M = std::max(hypotheses, 1000);
    std::vector<int> idx(4 * (size_t)M), counts(M);
    std::vector<double> H(9 * (size_t)M);
    mho_sample4(seed, 0, M, N, idx.data());
    mho_dlt4(x1, y1, x2, y2, idx.data(), M, H.data(), nullptr, nullptr);
    mho_score(x1, y1, x2, y2, N, H.data(), M, thr2, nullptr, counts.data());
    int best = 0;
    for (int m = 1; m < M; ++m) if (counts[m] > counts[best]) best = m;
    for (int i = 0; i < N; ++i)
        labeling[i] = fwd_d2(&H[9 * (size_t)best], x1[i], y1[i], x2[i], y2[i]) < thr2 ? 0 : -1;
    for (int q = 0; q < 9; ++q) H_out[q] = H[9 * (size_t)best + q];
}

// Process(), M/MultiH.cpp:42-98, from a known F (the product's SetEpipolarGeometry route: the correspondences count as
// already refined, so "original" and working points coincide).  init_mode: 0 = the given H0 (n_init models, what
// EstablishStablePointSets would hand over), 1 = stable point sets (the reference's own route), 2 = north_star's
// propose (hypotheses 4-point DLT models from counters 0.., greedy selection of at most max_propose with at least
// max(min_inliers, 8) inliers).  Seeds follow the product: mean shift of the stable sets seed ^ 0x57ab1e, MergingSteps
// seed ^ 0x4d53 ^ (c << 20), post-filter seed ^ 0xc0117a7, degenerate case seed ^ 0xdead.  post_filter = 0 skips
// :78-86 (the loop's own output).  Returns the number of models; H: capacity max_models*9.
MHO_API int mho_process(const double* x1, const double* y1, const double* x2, const double* y2, const double* aff, int N,
                        const double* F, const double* e2, double thr_h, double locality, double lambda, int min_inliers,
                        double straightness, uint64_t seed, int init_mode, const double* H0, int n_init, int hypotheses,
                        int max_propose, const int* hit_rowptr, const int* hit_col, mho_expand_hook expand,
                        int post_filter, double* H, int max_models, int* labeling, int* iterations, double* energy_out,
                        int* removed_by_filter, int* took_degenerate_tail)
{
    const double thr2 = thr_h * thr_h;
    std::vector<double> models((size_t)max_models * 9, 0.0);
    int nh = 0;
    if (init_mode == 0) {
        nh = std::min(n_init, max_models);
        std::copy(H0, H0 + 9 * (size_t)nh, models.begin());
    } else if (init_mode == 1) {
        nh = mho_establish_stable_point_sets(x1, y1, x2, y2, aff, N, F, e2, locality, thr_h, seed ^ 0x57ab1eull,
                                             models.data(), max_models);
        if (nh > max_models) return -1;
    } else {
        std::vector<int> idx(4 * (size_t)hypotheses);
        std::vector<double> Hh(9 * (size_t)hypotheses);
        mho_sample4(seed, 0, hypotheses, N, idx.data());
        mho_dlt4(x1, y1, x2, y2, idx.data(), hypotheses, Hh.data(), nullptr, nullptr);
        std::vector<unsigned char> mask(N, 1);
        nh = g_select_refit
                 ? mho_select_greedy_refit(x1, y1, x2, y2, aff, N, F, e2, Hh.data(), hypotheses, thr2, std::max(min_inliers, 8),
                                           std::min(max_propose, max_models), mask.data(), models.data(), nullptr, nullptr)
                 : mho_select_greedy(x1, y1, x2, y2, N, Hh.data(), hypotheses, thr2, std::max(min_inliers, 8),
                                     std::min(max_propose, max_models), mask.data(), models.data(), nullptr, nullptr);
    }
    if (iterations) *iterations = 0;
    if (energy_out) *energy_out = 0.0;
    nh = mho_cluster_merging_and_labeling(x1, y1, x2, y2, aff, N, models.data(), nh, max_models, F, e2, lambda, thr_h,
                                          straightness, hit_rowptr, hit_col, seed, expand, labeling, iterations, energy_out);
    if (nh > max_models) return -1;
    if (removed_by_filter) *removed_by_filter = 0;
    if (took_degenerate_tail) *took_degenerate_tail = 0;
    if (nh > 1 && post_filter) {                                                 // :78-86
        std::vector<double> s(2 * (size_t)N), d(2 * (size_t)N);
        for (int i = 0; i < N; ++i) { s[2 * i] = x1[i]; s[2 * i + 1] = y1[i]; d[2 * i] = x2[i]; d[2 * i + 1] = y2[i]; }
        const int kept = mho_compatibility_check(s.data(), d.data(), N, labeling, models.data(), nh, F, thr2, min_inliers,
                                                 seed ^ 0xc0117a7ull, nullptr);
        if (removed_by_filter) *removed_by_filter = nh - kept;
        nh = kept;
    }
    if (nh <= 1) {                                                               // :88-94
        mho_handle_degenerate(x1, y1, x2, y2, N, thr2, seed ^ 0xdeadull, hypotheses, labeling, models.data());
        nh = 1;
        if (took_degenerate_tail) *took_degenerate_tail = 1;
    }
    for (int i = 0; i < nh; ++i) for (int q = 0; q < 9; ++q) H[9 * (size_t)i + q] = models[9 * (size_t)i + q];
    return nh;
}

// ---------------------------------------------------------------------------
// 13. Process() from RAW correspondences (M/MultiH.cpp:42-98 with GetFundamentalMatrixAndRefineData, :770-848)
// ---------------------------------------------------------------------------
// Epipoles with third coordinate 1 (:786-799): e2 = eigenvector of F F^T, e1 = eigenvector of F^T F with the smallest
// eigenvalue (the last row of cv::eigen's descending order), divided by its third entry.  cv::eigen -> the Jacobi of
// section 5 (parity unpinned at that boundary, like every cv::eigen call of the path).
MHO_API void mho_epipoles(const double* F, double* e1, double* e2)
{
    for (int which = 0; which < 2; ++which) {
        double A[9], V[9], W[3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double a = 0.0;
                for (int k = 0; k < 3; ++k)
                    a = a + (which == 0 ? F[3 * k + i] * F[3 * k + j]          // Ft * F  (:795)
                                        : F[3 * i + k] * F[3 * j + k]);        // F * Ft  (:789)
                A[3 * i + j] = a;
            }
        jacobi_sym(3, A, V, W);
        int jm = 0;
        for (int j = 1; j < 3; ++j) if (W[j] < W[jm]) jm = j;
        double* out = which == 0 ? e1 : e2;
        out[0] = V[0 * 3 + jm] / V[2 * 3 + jm];
        out[1] = V[1 * 3 + jm] / V[2 * 3 + jm];
    }
}

// The front half as the build defines it where the reference calls cv::findFundamentalMat(RANSAC) (:775; OpenCV is outside
// /root/reference — parity unpinned): `hypotheses` normalised 8-point fits from counter-RNG 8-tuples
// (seed, counters 0..), inlier counts at thr_f^2 (Sampson's distance, or — mho_set_fundamental_metric(1) — the point-to-epipolar-line distance), the best-supported (lowest index on ties), two rounds of
// {inliers -> least-squares 8-point -> rank 2}, the inlier mask of the result; then the reference's own steps: the
// degenerate test ||F|| < 1e-5 (:779), the epipoles (:786-799), and per masked correspondence the Hartley-Sturm
// correction, the affine-consistency filter and the optimal affinity (:807-838, mho_refine_points).
// keep: N flags; refined: N x 8 (x1 y1 x2 y2 a11 a12 a21 a22 of the survivors, zero elsewhere).  Returns the number kept,
// or -1 in the degenerate case.
MHO_API int mho_front_half_ex(const double* x1, const double* y1, const double* x2, const double* y2, const double* aff, int N,
                              uint64_t seed, int hypotheses, double thr_f, double* F_out, double* e1_out, double* e2_out,
                              unsigned char* keep, double* refined, unsigned char* reason /* nullable: mho_refine_points_ex */)
{
    std::vector<int> idx(8 * (size_t)hypotheses), counts(hypotheses);
    std::vector<double> Fh(9 * (size_t)hypotheses);
    mho_sample8(seed, 0, hypotheses, N, idx.data());
    mho_fund8(x1, y1, x2, y2, idx.data(), hypotheses, Fh.data());
    const double thr2 = thr_f * thr_f;
#pragma omp parallel for schedule(static)
    for (int m = 0; m < hypotheses; ++m) mho_sampson_score(x1, y1, x2, y2, N, Fh.data() + 9 * (size_t)m, 1, thr2, &counts[m]);
    int best = 0;
    for (int m = 1; m < hypotheses; ++m) if (counts[m] > counts[best]) best = m;
    double Fa[9], Fb[9], Fc[9];
    std::vector<unsigned char> mask(N);
    mho_fund_refit(x1, y1, x2, y2, N, Fh.data() + 9 * (size_t)best, thr2, Fa, nullptr);
    mho_fund_refit(x1, y1, x2, y2, N, Fa, thr2, Fb, nullptr);
    mho_fund_refit(x1, y1, x2, y2, N, Fb, thr2, Fc, mask.data());            // (its mask = the inliers of Fb, the F returned)
    for (int i = 0; i < 9; ++i) F_out[i] = Fb[i];
    mho_epipoles(Fb, e1_out, e2_out);
    double nrm = 0.0;
    for (int i = 0; i < 9; ++i) nrm += Fb[i] * Fb[i];
    if (!(sqrt(nrm) >= 1e-5) || !std::isfinite(e2_out[0]) || !std::isfinite(e2_out[1])) return -1;     // :779
    mho_refine_points_ex(x1, y1, x2, y2, aff, N, Fb, e1_out, e2_out, mask.data(), keep, refined, reason);
    int kept = 0;
    for (int i = 0; i < N; ++i) kept += keep[i] ? 1 : 0;
    return kept;
}

MHO_API int mho_front_half(const double* x1, const double* y1, const double* x2, const double* y2, const double* aff, int N,
                           uint64_t seed, int hypotheses, double thr_f, double* F_out, double* e1_out, double* e2_out,
                           unsigned char* keep, double* refined)
{
    return mho_front_half_ex(x1, y1, x2, y2, aff, N, seed, hypotheses, thr_f, F_out, e1_out, e2_out, keep, refined, nullptr);
}

MHO_API int mho_abi_version(void) { return 3; }
