// r06 study (CPU, oracle code — never on the product path): how many alpha-moves of an expansion could be skipped EXACTLY by
// tracking, per label, whether anything that matters to its move has changed since the move last ran?
//
// Rule under study (the test of csrc/expand.hip's concurrent moves, applied across time instead of inside a batch): after a
// move on alpha has run, alpha is CLEAN; an accepted move on another label that changes the sites S makes beta DIRTY when some
// s in S u N(S) is not unary-kept for beta (D_beta(s) - D_label(s) <= potts * wsum(s), under the old or — for s in S — the new
// label).  A clean label's move cannot lower the energy.  The reference's own rule (skip when NOTHING was accepted since the
// label's last move) is the special case the engine already has.
//
//   g++ -O2 -o /tmp/dirty_label_study tools/dirty_label_study.cpp && /tmp/dirty_label_study problem.bin
// problem.bin (tools/dirty_label_study.py writes it): int32 N, L, potts, nhits; cost[N*L]; hit_rowptr[N+1]; hit_col[nhits]
#include "../oracle/mh_oracle.cpp"
#include <cstdio>

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[4];
    if (fread(hdr, 4, 4, f) != 4) return 2;
    const int N = hdr[0], L = hdr[1], potts = hdr[2], nh = hdr[3];
    std::vector<int> cost((size_t)N * L), rp(N + 1), col(nh);
    if (fread(cost.data(), 4, cost.size(), f) != cost.size() || fread(rp.data(), 4, rp.size(), f) != rp.size() || fread(col.data(), 4, col.size(), f) != col.size()) return 2;
    fclose(f);
    SymGraph g;
    build_sym(N, rp.data(), col.data(), g);
    std::vector<long long> wsum(N, 0);
    for (int i = 0; i < N; ++i) for (int k = g.rowptr[i]; k < g.rowptr[i + 1]; ++k) wsum[i] += g.w[k];
    Expander ex;
    ex.N = N; ex.L = L; ex.cost = cost.data(); ex.g = &g; ex.potts = potts;
    ex.label.assign(N, 0);
    ex.curCost.resize(N);
    ex.lookup.assign(N, -1);
    for (int i = 0; i < N; ++i) ex.curCost[i] = cost[(size_t)i * L];
    std::vector<char> dirty(L, 1);
    std::vector<int> mark(N, -1);
    int tlast = -1, t = 0, violations = 0;
    int new_energy = ex.compute_energy(), old_energy;
    printf("N %d, L %d, potts %d, arcs %zu\n", N, L, potts, g.col.size());
    for (int cycle = 1; cycle <= 1000; ++cycle) {
        old_energy = new_energy;
        long long moves = 0, accepted = 0, by_tlast = 0, clean_new = 0, open_all = 0, open_tlast = 0, open_clean = 0, dirtied = 0;
        for (int a = 0; a < L; ++a, ++t) {
            // how much of the graph the move would have to look at: sites its own numbers do not settle
            long long open = 0;
            for (int i = 0; i < N; ++i)
                if (ex.label[i] != a && std::llabs((long long)cost[(size_t)i * L + a] - ex.curCost[i]) <= 2ll * potts * wsum[i]) ++open;
            const bool skip_tlast = t >= L && tlast <= t - L;
            const bool clean = !dirty[a];
            std::vector<int> before = ex.label;
            const bool acc = ex.alpha_expansion(a);
            ++moves; open_all += open;
            if (skip_tlast) { ++by_tlast; open_tlast += open; }
            else if (clean) { ++clean_new; open_clean += open; }
            if (clean && acc) { ++violations; printf("VIOLATION: clean label %d accepted in cycle %d\n", a, cycle); }
            if (skip_tlast && !clean) { ++violations; printf("VIOLATION: label %d is skipped by the reference's rule but dirty\n", a); }
            dirty[a] = 0;
            if (acc) {
                ++accepted; tlast = t;
                for (int p = 0; p < N; ++p) {
                    if (before[p] == ex.label[p]) continue;
                    // p itself: under the old and under the new label; its neighbours (unless they moved too): under their label
                    for (int k = g.rowptr[p] - 1; k < g.rowptr[p + 1]; ++k) {
                        const int s = k < g.rowptr[p] ? p : g.col[k];
                        if (s != p && (before[s] != ex.label[s] || mark[s] == t)) continue;
                        if (s != p) mark[s] = t;
                        const long long W = (long long)potts * wsum[s];
                        const long long Dold = cost[(size_t)s * L + before[s]], Dnew = cost[(size_t)s * L + ex.label[s]];
                        for (int b = 0; b < L; ++b) {
                            if (b == a || dirty[b]) continue;
                            const long long Db = cost[(size_t)s * L + b];
                            bool ok = before[s] == b || Db - Dold > W;
                            if (ok && s == p) ok = Db - Dnew > W;
                            if (!ok) { dirty[b] = 1; ++dirtied; }
                        }
                    }
                }
            }
        }
        new_energy = ex.compute_energy();
        printf("cycle %d: %lld moves, %lld accepted; skipped by the reference's rule %lld; CLEAN beyond that %lld; labels dirtied %lld | open sites: all moves %lld, "
               "the reference's rule saves %lld, the clean ones %lld (%.1f %% of what is run now) | energy %d\n",
               cycle, moves, accepted, by_tlast, clean_new, dirtied, open_all, open_tlast, open_clean,
               open_all - open_tlast > 0 ? 100.0 * open_clean / (open_all - open_tlast) : 0.0, new_energy);
        if (new_energy == old_energy) break;
    }
    printf("violations of 'a clean move is never accepted': %d\n", violations);
    return violations ? 1 : 0;
}
