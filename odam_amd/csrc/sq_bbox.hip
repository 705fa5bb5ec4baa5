// sq_bbox.hip -- odam_sq_oriented_bbox (include/odam_sq.h): the reference's compute_oriented_bbox (src/utils/box_utils.py:319-410)
// for a batch of fitted surfaces, host code (hull2d.h says why the vertex order of qhull is part of the result), one thread per
// group of objects.  No device call is made.
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/odam_sq.h"
#include "hull2d.h"
#include "odam_err.h"

extern "C" int odam_sq_oriented_bbox(const float* points, int n_obj, int n_pts, double* corners, int* status) {
    if (n_obj < 0 || n_pts < 3 || (n_obj && (!points || !corners || !status))) return odam_fail(1, "odam_sq_oriented_bbox: bad argument");
    if (n_obj == 0) return 0;
    std::atomic<int> next{0};
    auto work = [&] {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n_obj) return;
            status[i] = hull2d::oriented_bbox(points + (size_t)i * n_pts * 3, n_pts, corners + (size_t)i * 24);
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (int)std::min<unsigned>(std::min<unsigned>(hw ? hw : 1, 16), (unsigned)((n_obj + 3) / 4));
    if (nt <= 1) { work(); return 0; }
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    for (int t = 0; t + 1 < nt; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    return 0;
}

// the hull alone (tests): hull [n_pts] receives the point indices in scipy's ConvexHull.vertices order, *n_hull their number
extern "C" int odam_sq_hull2d(const float* points, int n_pts, int* hull, int* n_hull, int* status) {
    if (!points || n_pts < 3 || !hull || !n_hull || !status) return odam_fail(1, "odam_sq_hull2d: bad argument");
    double c[24];
    std::vector<int> hv;
    *status = hull2d::oriented_bbox(points, n_pts, c, &hv);
    *n_hull = (int)hv.size();
    for (size_t i = 0; i < hv.size() && i < (size_t)n_pts; i++) hull[i] = hv[i];
    return 0;
}

// diagnostics: the surviving facets in list (= creation) order as point-index pairs, what scipy's ConvexHull.simplices lists
extern "C" int odam_sq_hull2d_facets(const float* points, int n_pts, int* pairs, int* n_facets) {
    std::vector<double> xy(2 * (size_t)n_pts);
    for (int i = 0; i < n_pts; i++) { xy[2 * i] = points[3 * i]; xy[2 * i + 1] = points[3 * i + 1]; }
    hull2d::Qhull2 q;
    q.build(xy.data(), n_pts);
    int k = 0;
    for (int f = q.head; f >= 0; f = q.F[f].next) { pairs[2 * k] = q.vpoint[q.F[f].v[0]]; pairs[2 * k + 1] = q.vpoint[q.F[f].v[1]]; k++; }
    *n_facets = k;
    return q.why | (q.ambiguous ? 256 : 0);
}
