/*
 * odam_sq.h -- C ABI of the MI355X (gfx950) super-quadric multi-view fit.
 *
 * Drop-in boundary for the reference's (likojack/ODAM) SQ path:
 *   src/scripts/run_multi_view.py:22-76       optim_process       (per-object driver, host Python)
 *   src/super_quadric/sq_libs.py:432-475      SuperQuadricOptimizer.run   -> odam_sq_fit_batch
 *   src/super_quadric/sq_libs.py:577-595      compute_ellipsoid_points    -> odam_sq_points_batch
 *   learnable_primitives/fast_sampler/sampling.hpp:5-15  sample_on_batch  -> odam_sq_sample
 *
 * Conventions: plain pointers and sizes, no exceptions across the boundary, int return code
 * (0 = OK, otherwise an ODAM_E_* value; odam_last_error() returns a static message), caller-owned
 * buffers, work is enqueued on the caller's hipStream_t (passed as void*).  All scratch lives in the handle: the
 * *_batch calls neither synchronise nor allocate, except that odam_sq_fit_batch grows the handle's exchange
 * buffer (hipMalloc) the first time a larger problem is seen and, for a call with more objects than the device has CUs, reads the
 * view counts back (one small copy and a synchronisation of the caller's stream) to fit the objects longest first.  A handle serves one launch at a time (use one
 * per stream / thread).
 * Pointers marked [dev] are device pointers, [host] host pointers.
 */
#ifndef ODAM_SQ_H
#define ODAM_SQ_H
#ifdef __cplusplus
extern "C" {
#endif

#define ODAM_OK 0
#define ODAM_E_INVALID 1   /* bad argument (null pointer, size out of range) */
#define ODAM_E_HIP 2       /* a HIP runtime call failed */
#define ODAM_E_LIMIT 3     /* problem exceeds a compiled limit (views per object, iterations) */

#define ODAM_SQ_POINTS 1000      /* surface samples per super-quadric   (sq_libs.py:545) */
#define ODAM_SQ_MAX_VIEWS 1024   /* views one workgroup reduces; an object split over k workgroups may have k times as many
                                    (k <= 32, odam_config sq.split, and k * padded object count <= number of CUs; odam_sq_fit_batch takes at most
                                    16 * ODAM_SQ_MAX_VIEWS views per object) */

/* representation codes, sq_libs.py:362-386 */
#define ODAM_SQ_SUPER_QUADRIC 0  /* 9 parameters optimised */
#define ODAM_SQ_CUBE 1           /* shapes frozen (caller passes -10000), 7 optimised */
#define ODAM_SQ_QUADRIC 2        /* shapes frozen at -0, 7 optimised */

typedef struct odam_sq_ctx odam_sq_ctx;

const char* odam_last_error(void);

/* Creates the per-device context: uploads the sampler's constant random draws
 * (std::mt19937(0), _sampler.pyx:438) and the Adam bias-correction tables for up to max_iters
 * steps (lr 0.01 / 0.1 for the shape logits, sq_libs.py:373-387). */
int odam_sq_create(int max_iters, odam_sq_ctx** out);
int odam_sq_destroy(odam_sq_ctx* ctx);

/*
 * Fits n_obj super-quadrics, n_iters Adam steps each, entirely on device: one 1024-thread workgroup per object,
 * or -- when the grid would leave most CUs idle -- up to thirty-two per object that split the views of the per-view
 * extent search by residue class and exchange one sub-tree root each per step (results are bit-identical either way).
 *   init_params  [dev] [n_obj][9]   translate[3], angle, scales[3] (= sqrt(dims/2)), shapes[2]
 *   class_id     [dev] [n_obj]      0..7 selects the scale prior row; < 0 = no prior (prior=False)
 *   view_offsets [dev] [n_obj+1]    object o owns views view_offsets[o] .. view_offsets[o+1]-1
 *   P            [dev] [sumF][12]   row-major 3x4 projection K @ inv(T_wc)[:3,:] as float32
 *   tgt          [dev] [sumF][4]    bbox edge in pixels per direction x_min,x_max,y_min,y_max
 *                                   (the reference stores gt = -pixel and compares with -gt)
 *   mask         [dev] [sumF][4]    1 = edge is a constraint, 0 = dropped (near the image border)
 *   prior_icov   [dev] [8][9]       row-major 3x3 inverse covariances, CLASS_MAPPER order
 *   out_params   [dev] [n_obj][9]
 *   out_points   [dev] [n_obj][1000][3]  surface of the fitted SQ (compute_ellipsoid_points); nullable
 *   loss_log     [dev] [n_obj][n_iters]  loss_2d per step (sq_libs.py:471); nullable
 *   traj         [dev] [n_obj][n_iters][9] parameters after every step; nullable (parity tests)
 *   max_views    largest view count of any object (host-known; sizes the workgroup's LDS); up to 16 * ODAM_SQ_MAX_VIEWS
 *                when few enough objects are fitted per call for the view split to cover it, else ODAM_E_LIMIT
 */
int odam_sq_fit_batch(odam_sq_ctx* ctx, int n_obj, const float* init_params, const int* class_id,
                      const int* view_offsets, const float* P, const float* tgt, const float* mask,
                      const float* prior_icov, int n_iters, int representation, int max_views,
                      float* out_params, float* out_points, float* loss_log, float* traj,
                      void* stream);

/* Surface points of n super-quadrics: params [dev][n][9] -> out_points [dev][n][1000][3]. */
int odam_sq_points_batch(odam_sq_ctx* ctx, int n, const float* params, float* out_points, void* stream);

/* Projected extent of n super-quadric surfaces in one camera -- what OdamProcess._prepare_tracks (src/processor.py:181-207)
 * computes per live track and frame: params [dev][n][9] -> surface points (as odam_sq_points_batch) -> float64
 * [p, 1] T_cw^T K^T, divide by depth, min / max.  T_cw12_K9 [host][21]: rows 0..2 of inv(T_wc) (12 values, row-major), then K
 * (9, row-major); out_px [dev][n][4] float64 = x_min, y_min, x_max, y_max.  Stream-ordered. */
int odam_sq_project_extents(odam_sq_ctx* ctx, int n, const float* params, const double* T_cw12_K9, double* out_px, void* stream);

/* The reference's one native symbol under its own name and signature
 * (learnable_primitives/fast_sampler/sampling.hpp:5-15; bound by _sampler.pyx:430-439, which passes
 * buffer_size = 201, seed = 0):  shapes [host][B][M][3], epsilons [host][B][M][2] -> etas, omegas [host][B][M][N].
 * One std::mt19937(seed) stream over all primitives in (b, m) order.  void, no error path, re-entrant -- as upstream. */
void sample_on_batch(float* shapes, float* epsilons, float* etas, float* omegas, int B, int M, int N,
                     int buffer_size, int seed);

/* sample_on_batch for the call the pipeline makes (B = M = 1, N = 1000, buffer_size = 201, seed = 0) with this
 * library's return-code convention:  a[3], e[2] -> etas[1000], omegas[1000]  (all [host]). */
int odam_sq_sample(const float* a, const float* e, float* etas, float* omegas);

/* Result extraction of a fit pass (run_multi_view.py:66-67): the reference's compute_oriented_bbox (src/utils/box_utils.py:319-410)
 * for n_obj surfaces -- qhull's 2-D hull of the xy projection walked as an OPEN polygon from qhull's own start vertex, the
 * smallest rectangle over the edge directions, the z extent.  points [host][n_obj][n_pts][3] float32 (the fit's out_points,
 * downloaded) -> corners [host][n_obj][8][3] float64, status [host][n_obj]: 0 = done; 1 = this object sits inside qhull's
 * round-off tolerance band (or two candidate rectangles tie to 1e-9) and must be recomputed with scipy / qhull itself, which is
 * what the reference calls -- odam_amd/multi_view.py does.  Host code, multi-threaded, no device call. */
int odam_sq_oriented_bbox(const float* points, int n_obj, int n_pts, double* corners, int* status);
/* the hull alone (tests): hull [host][n_pts] <- point indices in the order of scipy's ConvexHull(points[:, :2]).vertices */
int odam_sq_hull2d(const float* points, int n_pts, int* hull, int* n_hull, int* status);

#ifdef __cplusplus
}
#endif
#endif
