// hull2d.h -- the oriented box of a fitted surface (run_multi_view.py:66-67 -> src/utils/box_utils.py:319-410) without Python.
//
// The reference takes scipy.spatial.ConvexHull of the 1000 surface points' xy, walks `hull.vertices` as an OPEN polygon (the
// edge from the last vertex back to the first is never tried) and picks the smallest rectangle over the edge directions.  Which
// edge is left out is decided by where qhull's vertex walk starts: at the first facet of qhull's facet list, i.e. at the OLDEST
// facet that survived the incremental construction.  So a restatement has to reproduce not only the hull (any algorithm gives
// that) but the order in which qhull 2019.1 (libqhull_r 7.3.2, the version scipy bundles; default options + Qt) creates its
// facets.  `Qhull2` below follows that construction for dimension 2:
//   initial simplex   qh_maxmin / qh_maxsimplex: min-x and max-x point, then of {min-y, max-y} the one with the larger
//                     |determinant|; facets created by qh_createsimplex in the order "omits the newest vertex first";
//   first partition   qh_partitionall: every other point goes to the FIRST facet (list order) it is outside of, the farthest
//                     point of a facet kept last in its outside set;
//                     qh_furthestnext (once): the facet whose farthest point is the farthest of all goes to the front;
//   main loop         qh_nextfurthest: first facet in list order with outside points, its farthest point p;
//                     qh_findhorizon: visible facets breadth-first from that facet, neighbour slot 0 (opposite the vertex with
//                     the higher id) before slot 1; qh_makenewfacets: one new facet per horizon vertex, in the order the
//                     horizon is met, appended to the END of the list; qh_partitionvisible: the outside points of the visible
//                     facets, in that order, go to the first new facet they are outside of (qh_findbest without `bestoutside`),
//                     else to the second; visible facets leave the list.
//   distances         qh_sethyperplane_det / qh_normalize2 / qh_distplane for dim 2, operation by operation (compiled without
//                     contraction), so a "farthest point" is the one qhull finds.
// What is NOT restated is qhull's handling of points within its round-off tolerance of a facet (coplanar points, merged
// facets, narrow hulls: thresholds of ~1e-15 of the coordinate range).  Every decision whose distance falls inside a band
// three orders of magnitude wider than those thresholds marks the result AMBIGUOUS, and so does a near-tie between two
// candidate rectangles (the reference's winner then hangs on the last bit of a BLAS product): the caller recomputes such an
// object with scipy, exactly as the reference does.  On float32 surface points the band is hit by exact coincidences only.
// tests/test_oriented_bbox.py checks vertices, start vertex and boxes against scipy itself on thousands of fitted surfaces.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace hull2d {

struct Qhull2 {
    struct Facet {
        int v[2];           // vertex ids, v[0] the newer one (qhull keeps a facet's vertices sorted by descending id)
        int nb[2];          // nb[i] is the neighbour opposite v[i], i.e. the facet across v[1 - i]
        double n0, n1, off; // outward unit normal and offset: dist(p) = off + p.x n0 + p.y n1
        std::vector<int> out;   // outside set (point indices), the farthest one last
        double furthest;
        bool alive, visible, newfacet;
        int visit;
        int next, prev;     // qh.facet_list
    };
    const double* P;        // [n][2]
    int n;
    std::vector<int> vpoint;            // vertex id -> point index
    std::vector<Facet> F;
    int head = -1, last = -1, facet_next = -1;      // facet_next == -1: at qh.facet_tail
    int visit = 0;
    double ix, iy;                      // interior point (mean of the initial simplex)
    double guard;
    bool ambiguous = false;
    int why = 0;        // diagnostics: bit mask of the places that raised `ambiguous`

    double dist(const Facet& f, int p) const { return f.off + P[2 * p] * f.n0 + P[2 * p + 1] * f.n1; }
    bool same_point(int p, int vtx) const { const int q = vpoint[vtx]; return P[2 * p] == P[2 * q] && P[2 * p + 1] == P[2 * q + 1]; }
    // Is p outside f (qhull: dist >= MINoutside / > MINvisible, thresholds ~1e-15 of the coordinate range)?  A distance inside the
    // band is "not outside"; unless p is a copy of one of f's own vertices (the xy projection of a sampled surface is full of exact
    // duplicates: they lie at a round-off distance ~1e-16, which qhull's thresholds classify the same way) the case is left to qhull.
    bool outside(const Facet& f, int p, double d) {
        if (std::fabs(d) >= guard) return d > 0;
        if (!same_point(p, f.v[0]) && !same_point(p, f.v[1])) { ambiguous = true; why |= 1; }
        return false;
    }

    void list_append(int f) {       // qh_appendfacet
        F[f].prev = last; F[f].next = -1;
        if (last >= 0) F[last].next = f; else head = f;
        last = f;
        if (facet_next < 0) facet_next = f;
    }
    void list_remove(int f) {       // qh_removefacet
        if (facet_next == f) facet_next = F[f].next;
        if (F[f].prev >= 0) F[F[f].prev].next = F[f].next; else head = F[f].next;
        if (F[f].next >= 0) F[F[f].next].prev = F[f].prev; else last = F[f].prev;
    }

    void plane(Facet& f) {      // qh_setfacetplane -> qh_sethyperplane_det (dim 2) -> qh_normalize2; orientation by the interior point
        const double* p0 = P + 2 * vpoint[f.v[0]];
        const double* p1 = P + 2 * vpoint[f.v[1]];
        double n0 = p1[1] - p0[1], n1 = p0[0] - p1[0];
        double norm = std::sqrt(n0 * n0 + n1 * n1);
        if (!(norm > 0)) { ambiguous = true; norm = 1; }
        n0 /= norm; n1 /= norm;
        double off = -(p0[0] * n0 + p0[1] * n1);
        const double di = off + ix * n0 + iy * n1;
        if (di > 0) { n0 = -n0; n1 = -n1; off = -(p0[0] * n0 + p0[1] * n1); }      // x / (-norm) == -(x / norm) exactly
        f.n0 = n0; f.n1 = n1; f.off = off;
    }

    int new_facet(int va, int vb) {
        Facet f;
        f.v[0] = std::max(va, vb); f.v[1] = std::min(va, vb);
        f.nb[0] = f.nb[1] = -1; f.furthest = 0; f.alive = true; f.visible = false; f.newfacet = true; f.visit = 0; f.next = f.prev = -1;
        plane(f);
        F.push_back(std::move(f));
        list_append((int)F.size() - 1);
        return (int)F.size() - 1;
    }

    // qh_findbesthorizon (not ischeckmax) with a search distance of zero: from `best` climb to neighbours (ANY facet, new or old)
    // that lie farther from p; a neighbour within the band below the best one would be searched by qhull too -> ambiguous
    void best_horizon(int p, int& best, double& bestdist) {
        std::vector<int> stack;
        visit++;
        F[best].visit = visit;
        int facet = best;
        for (;;) {
            int nextfacet = -1;
            for (int k = 0; k < 2; k++) {
                const int nb = F[facet].nb[k];
                if (F[nb].visit == visit) continue;
                F[nb].visit = visit;
                const double d = dist(F[nb], p);
                if (d > bestdist) { best = nb; bestdist = d; }
                else {
                    if (bestdist - d < guard && !(same_point(p, F[nb].v[0]) || same_point(p, F[nb].v[1]))) {
                        ambiguous = true; why |= 2;
#ifdef HULL2D_DEBUG
                        fprintf(stderr, "band: p %d (%.9g %.9g) best %d bd %.3e nb %d d %.3e nbverts (%d %d)\n", p, P[2*p], P[2*p+1], best, bestdist, nb, d, vpoint[F[nb].v[0]], vpoint[F[nb].v[1]]);
#endif
                    }
                    if (!(bestdist - d < guard)) continue;      // clearly below: not searched
                }
                if (nextfacet >= 0) stack.push_back(nextfacet);
                nextfacet = nb;
            }
            if (nextfacet >= 0) { facet = nextfacet; continue; }
            if (stack.empty()) break;
            facet = stack.back(); stack.pop_back();
        }
    }

    // qh_findbestnew: the facets from `start` to the end of the list, then from the first new facet up to `start`: the first one p is
    // outside of, else the farthest + horizon search
    void find_best_new(int p, int start, int first_new, int& best, double& bestdist, bool& is_out) {
        best = -1; bestdist = -1e300; is_out = false;
        for (int pass = 0; pass < 2; pass++)
            for (int f = pass ? first_new : start; f >= 0; f = F[f].next) {
                if (pass && f == start) break;
                const double d = dist(F[f], p);
                if (d > bestdist) {
                    best = f;
                    if (outside(F[f], p, d)) { bestdist = d; is_out = true; return; }
                    bestdist = d;
                }
            }
        best_horizon(p, best, bestdist);
        is_out = outside(F[best], p, bestdist);
    }

    // qh_partitionpoint for a point of a visible facet's outside set
    void partition_point(int p, int first_new, bool& use_findbestnew, bool& notsharp) {
        int best; double bd; bool is_out = false;
        if (use_findbestnew) find_best_new(p, first_new, first_new, best, bd, is_out);
        else {
            // qh_findbest(isnewfacets, !bestoutside): directed search over facets flagged new
            visit++;
            best = first_new; bd = dist(F[best], p);
            F[best].visit = visit;
            if (outside(F[best], p, bd)) is_out = true;
            else {
                int facet = best;
                while (facet >= 0 && !is_out) {
                    int found = -1;
                    for (int k = 0; k < 2; k++) {
                        const int nb = F[facet].nb[k];
                        if (!F[nb].newfacet || F[nb].visit == visit) continue;
                        F[nb].visit = visit;
                        const double d = dist(F[nb], p);
                        if (d > bd) {
                            best = nb; bd = d; found = nb;
                            if (outside(F[nb], p, d)) is_out = true;
                            break;
                        }
                    }
                    facet = found;
                }
                if (!is_out) {
                    bool via_new = false;
                    if (!notsharp) {
                        if (std::fabs(bd) < guard) { if (!same_point(p, F[best].v[0]) && !same_point(p, F[best].v[1])) { ambiguous = true; why |= 4; } }
                        if (bd < 0) {       // qh_sharpnewfacets: do the normals of the new facets lie in different quadrants?
                            bool sharp = false;
                            const bool q0 = F[first_new].n0 > 0, q1 = F[first_new].n1 > 0;
                            for (int f = F[first_new].next; f >= 0 && !sharp; f = F[f].next)
                                if ((F[f].n0 > 0) != q0 || (F[f].n1 > 0) != q1) sharp = true;
                            if (sharp) { find_best_new(p, best, first_new, best, bd, is_out); use_findbestnew = true; via_new = true; }
                            else notsharp = true;
                        }
                    }
                    if (!via_new) { best_horizon(p, best, bd); is_out = outside(F[best], p, bd); }
                }
            }
        }
        if (!is_out) return;        // inside, or coplanar within qhull's tolerance (dropped: no Qc)
        Facet& f = F[best];
        if (f.out.empty()) {
            f.out.push_back(p); f.furthest = bd;
            if (facet_next != best) {
                if (f.newfacet) { if (facet_next >= 0 && F[facet_next].newfacet) facet_next = first_new; }
                else { list_remove(best); list_append(best); f.newfacet = true; }      // an old facet gets its first outside point: to the end of the list
            }
        } else if (f.furthest < bd) { f.out.push_back(p); f.furthest = bd; }
        else f.out.insert(f.out.end() - 1, p);
    }

    // returns false when the input is degenerate beyond what is restated (caller: ambiguous)
    bool build(const double* pts, int n_) {
        P = pts; n = n_;
        if (n < 3) return false;
        // qh_maxmin: per coordinate the first strict minimum / maximum ("else if": a new maximum is not tested as a minimum)
        int mn[2], mx[2];
        double maxabs = 0, maxwidth = 0;
        for (int k = 0; k < 2; k++) {
            int lo = 0, hi = 0;
            for (int i = 0; i < n; i++) {
                if (P[2 * hi + k] < P[2 * i + k]) hi = i;
                else if (P[2 * lo + k] > P[2 * i + k]) lo = i;
            }
            mn[k] = lo; mx[k] = hi;
            maxabs = std::max(maxabs, std::max(P[2 * hi + k], -P[2 * lo + k]));
            maxwidth = std::max(maxwidth, P[2 * hi + k] - P[2 * lo + k]);
        }
        if (!(maxwidth > 0)) return false;
        guard = 1e-13 * std::max(1.0, maxabs);
        // qh_maxsimplex: min-x, max-x among the four extreme points, then the extreme point with the largest |det|
        const int cand[4] = {mn[0], mx[0], mn[1], mx[1]};
        int minx = -1, maxx = -1;
        double mincoord = INFINITY, maxcoord = -INFINITY;
        for (int c : cand) {
            if (maxcoord < P[2 * c]) { maxcoord = P[2 * c]; maxx = c; }
            if (mincoord > P[2 * c]) { mincoord = P[2 * c]; minx = c; }
        }
        if (minx == maxx) return false;
        const double prevdet = maxcoord - mincoord;
        int third = -1;
        double maxdet = -1.0;
        for (int c : cand) {
            if (c == minx || c == maxx) continue;
            const double a0 = P[2 * minx] - P[2 * c], a1 = P[2 * minx + 1] - P[2 * c + 1];
            const double b0 = P[2 * maxx] - P[2 * c], b1 = P[2 * maxx + 1] - P[2 * c + 1];
            const double det = std::fabs(a0 * b1 - a1 * b0);
            if (det > maxdet) { maxdet = det; third = c; }
        }
        // a narrow distribution makes qhull search all points for a better third vertex (ratio < 1e-3): not restated
        if (third < 0 || !(maxdet / (prevdet * maxwidth) > 1e-2)) return false;
        // qh_initialvertices: ids in simplex order; qh_createsimplex: facet i omits the vertex with the i-th HIGHEST id
        vpoint = {minx, maxx, third};
        ix = (P[2 * minx] + P[2 * maxx] + P[2 * third]) / 3; iy = (P[2 * minx + 1] + P[2 * maxx + 1] + P[2 * third + 1]) / 3;
        F.reserve(512);
        const int f0 = new_facet(1, 0), f1 = new_facet(2, 0), f2 = new_facet(2, 1);
        F[f0].nb[0] = f1; F[f0].nb[1] = f2; F[f1].nb[0] = f0; F[f1].nb[1] = f2; F[f2].nb[0] = f0; F[f2].nb[1] = f1;
        {   // a narrow initial simplex (qh_initialhull: facet normals nearly opposite) switches qhull to other partition rules
            double minangle = 1;
            for (int a = 0; a < 3; a++) for (int b = a + 1; b < 3; b++) minangle = std::min(minangle, F[a].n0 * F[b].n0 + F[a].n1 * F[b].n1);
            if (minangle < -0.99999) return false;
        }
        // qh_partitionall: first fit in facet order
        std::vector<int> rest;
        rest.reserve(n);
        for (int i = 0; i < n; i++) if (i != minx && i != maxx && i != third) rest.push_back(i);
        for (int fi = 0; fi < 3; fi++) {
            Facet& f = F[fi];
            std::vector<int> keep;
            keep.reserve(rest.size());
            int best = -1; double bestd = 0;
            for (int p : rest) {
                const double d = dist(f, p);
                if (!outside(f, p, d)) { keep.push_back(p); continue; }
                if (best < 0) { best = p; bestd = d; }
                else if (d > bestd) { f.out.push_back(best); best = p; bestd = d; }
                else f.out.push_back(p);
            }
            if (best >= 0) { f.out.push_back(best); f.furthest = bestd; }
            rest.swap(keep);
        }
        for (int fi = 0; fi < 3; fi++) F[fi].newfacet = false;       // qh_resetlists
        facet_next = head;
        {   // qh_initbuild -> qh_furthestnext: the facet with the farthest of the farthest points moves to the front of the list
            int bf = -1; double bd = -1e300;
            for (int f = head; f >= 0; f = F[f].next)
                if (!F[f].out.empty() && F[f].furthest > bd) { bf = f; bd = F[f].furthest; }
            if (bf >= 0 && bf != head) {
                list_remove(bf);
                F[bf].prev = -1; F[bf].next = head; F[head].prev = bf; head = bf;
            }
            facet_next = head;
        }
        // qh_buildhull
        std::vector<int> vis, created;
        int guard_iter = 0;
        for (;;) {
            while (facet_next >= 0 && F[facet_next].out.empty()) facet_next = F[facet_next].next;      // qh_nextfurthest
            if (facet_next < 0) break;
            if (++guard_iter > 4 * n) return false;
            const int start = facet_next;
            const int p = F[start].out.back();
            F[start].out.pop_back();
            // qh_findhorizon: the visible facets move to the end of the list in the order they are found
            vis.clear(); created.clear();
            visit++;
            list_remove(start); list_append(start);
            F[start].visible = true; F[start].visit = visit; vis.push_back(start);
            for (size_t q = 0; q < vis.size(); q++) {
                const int vf = vis[q];
                for (int k = 0; k < 2; k++) {
                    const int nbf = F[vf].nb[k];
                    if (F[nbf].visit == visit) continue;
                    F[nbf].visit = visit;
                    const double d = dist(F[nbf], p);
                    if (outside(F[nbf], p, d)) { list_remove(nbf); list_append(nbf); F[nbf].visible = true; vis.push_back(nbf); }
                }
            }
            int alive_count = 0;
            for (int f = head; f >= 0; f = F[f].next) alive_count++;
            if ((int)vis.size() >= alive_count) return false;
            // qh_makenewfacets
            const int apex = (int)vpoint.size();
            vpoint.push_back(p);
            for (size_t qi = 0; qi < vis.size(); qi++)
                for (int k = 0; k < 2; k++) {
                    const int vf = vis[qi];
                    const int h = F[vf].nb[k];
                    if (F[h].visible) continue;
                    const int shared = F[vf].v[1 - k];
                    const int nf = new_facet(apex, shared);
                    F[nf].nb[0] = h;                                    // opposite the apex
                    for (int s = 0; s < 2; s++) if (F[h].nb[s] == vf && F[h].v[1 - s] == shared) F[h].nb[s] = nf;
                    created.push_back(nf);
                }
            if (created.size() != 2) return false;                      // a convex polygon has two horizon vertices
            F[created[0]].nb[1] = created[1]; F[created[1]].nb[1] = created[0];
            // qh_partitionvisible
            bool use_findbestnew = false, notsharp = false;
            for (size_t qi = 0; qi < vis.size(); qi++) {
                const int vf = vis[qi];
                std::vector<int> pts_out;
                pts_out.swap(F[vf].out);
                for (int q : pts_out) partition_point(q, created[0], use_findbestnew, notsharp);
            }
            // qh_deletevisible, qh_resetlists
            for (int vf : vis) { list_remove(vf); F[vf].alive = false; }
            for (int f = created[0]; f >= 0; f = F[f].next) F[f].newfacet = false;
#ifdef HULL2D_DEBUG
            printf("add %d :", p);
            for (int f = head; f >= 0; f = F[f].next) printf(" %d-%d", vpoint[F[f].v[0]], vpoint[F[f].v[1]]);
            printf("\n");
#endif
        }
        return !ambiguous;
    }

    // hull vertices as scipy's ConvexHull.vertices lists them (point indices, counter-clockwise, starting at the first facet of the list)
    void vertices(std::vector<int>& out) const {
        out.clear();
        const int s = head;
        int f = s;
        do {
            const Facet& ft = F[f];
            // walk with the interior on the left: from A to B where the outward normal (n0, n1) is the edge direction turned by -90 deg
            const double* a = P + 2 * vpoint[ft.v[0]];
            const double* b = P + 2 * vpoint[ft.v[1]];
            const bool a_first = (b[0] - a[0]) * ft.n1 - (b[1] - a[1]) * ft.n0 < 0;      // cross(edge, normal) < 0 <=> normal to the right
            out.push_back(vpoint[a_first ? ft.v[0] : ft.v[1]]);
            f = a_first ? ft.nb[0] : ft.nb[1];                                           // the facet across B
        } while (f != s && (int)out.size() <= (int)F.size());
    }
};

// CPython's float % for a positive divisor (floatobject.c float_rem): fmod, moved into [0, w)
inline double py_mod(double v, double w) {
    double m = std::fmod(v, w);
    if (m != 0.0) { if ((w < 0) != (m < 0)) m += w; }
    else m = std::copysign(0.0, w);
    return m;
}

// box_utils.py:319-410 on float32 points [n][3] -> corners [8][3] float64.  Returns 0, or 1 when the object has to be recomputed
// with scipy (degenerate / inside qhull's tolerance band / two candidate rectangles within 1e-9 of each other).
inline int oriented_bbox(const float* pts, int n, double* corners, std::vector<int>* hull_out = nullptr) {
    std::vector<double> xy(2 * (size_t)n);
    float zmin = pts[2], zmax = pts[2];
    for (int i = 0; i < n; i++) {
        xy[2 * i] = pts[3 * i]; xy[2 * i + 1] = pts[3 * i + 1];
        zmin = std::min(zmin, pts[3 * i + 2]); zmax = std::max(zmax, pts[3 * i + 2]);
    }
    Qhull2 q;
    if (!q.build(xy.data(), n)) return 1;
    std::vector<int> hv;
    q.vertices(hv);
    if (hull_out) *hull_out = hv;
    const int V = (int)hv.size();
    if (V < 3) return 1;
    // contour (float32), its mean over axis 0 (numpy adds the rows one after the other in float32, then divides), centred in float32
    std::vector<float> cx(V), cy(V);
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < V; i++) { cx[i] = pts[3 * hv[i]]; cy[i] = pts[3 * hv[i] + 1]; }
    sx = cx[0]; sy = cy[0];
    for (int i = 1; i < V; i++) { sx += cx[i]; sy += cy[i]; }
    const float mxf = sx / (float)V, myf = sy / (float)V;
    for (int i = 0; i < V; i++) { cx[i] -= mxf; cy[i] -= myf; }
    // candidate directions: the V - 1 edges of the OPEN polygon, folded into [0, pi/2), sorted, duplicates removed
    const double half_pi = M_PI / 2;
    std::vector<double> ang(V - 1);
    for (int i = 0; i + 1 < V; i++) {
        const float ex = cx[i + 1] - cx[i], ey = cy[i + 1] - cy[i];
        ang[i] = std::fabs(py_mod(std::atan2((double)ey, (double)ex), half_pi));
    }
    std::sort(ang.begin(), ang.end());
    ang.erase(std::unique(ang.begin(), ang.end()), ang.end());
    double best_area = 10000000000.0, best[5] = {0, 0, 0, 0, 0}, second = INFINITY;
    bool found = false;
    for (double a : ang) {
        const double c = std::cos(a), cm = std::cos(a - half_pi), cp = std::cos(a + half_pi);
        double lox = INFINITY, hix = -INFINITY, loy = INFINITY, hiy = -INFINITY;
        for (int i = 0; i < V; i++) {
            const double x = cx[i], y = cy[i];
            const double rx = c * x + cm * y, ry = cp * x + c * y;
            lox = std::min(lox, rx); hix = std::max(hix, rx); loy = std::min(loy, ry); hiy = std::max(hiy, ry);
        }
        const double area = (hix - lox) * (hiy - loy);
        if (area < best_area) {
            if (found) second = std::min(second, best_area);
            best_area = area; best[0] = a; best[1] = lox; best[2] = hix; best[3] = loy; best[4] = hiy; found = true;
        } else second = std::min(second, area);
    }
    if (!found) return 1;
    if (second - best_area <= 1e-9 * best_area) return 1;      // the reference's choice between the two hangs on the last bits of a BLAS product
    const double a = best[0], lox = best[1], hix = best[2], loy = best[3], hiy = best[4];
    const double R00 = std::cos(a), R01 = std::cos(a - half_pi), R10 = std::cos(a + half_pi), R11 = std::cos(a);
    const double c2[4][2] = {{hix, hiy}, {hix, loy}, {lox, loy}, {lox, hiy}};
    for (int i = 0; i < 4; i++) {
        const double x = c2[i][0] * R00 + c2[i][1] * R10 + (double)mxf;
        const double y = c2[i][0] * R01 + c2[i][1] * R11 + (double)myf;
        corners[3 * i] = x; corners[3 * i + 1] = y; corners[3 * i + 2] = (double)zmax;
        corners[3 * (i + 4)] = x; corners[3 * (i + 4) + 1] = y; corners[3 * (i + 4) + 2] = (double)zmin;
    }
    return 0;
}

}  // namespace hull2d
