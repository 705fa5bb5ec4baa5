/*
 * odam_assoc.h -- C ABI of the association network forward pass on MI355X.
 *
 * Replaces, for the inference call OdamProcess.process_frame makes (src/processor.py:327-337):
 *   src/models/associator.py:202-254  Associator.forward up to and including log_optimal_transport
 *   src/models/associator.py:283-312  log_optimal_transport / log_sinkhorn_iterations   -> odam_assoc_sinkhorn
 * The Hungarian step (associator.py:19-35, scipy.optimize.linear_sum_assignment) stays on the host.
 * Conventions as in odam_sq.h.
 */
#ifndef ODAM_ASSOC_H
#define ODAM_ASSOC_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct odam_assoc odam_assoc;

/* gnn_is_cross[n_gnn_layers]: 1 for a 'cross' layer, 0 for 'self' (cfg.GNN_layers); n_self_layers = len(cfg.self_GNN_layers) */
int odam_assoc_create(int max_tracks, int n_self_layers, const int* gnn_is_cross, int n_gnn_layers, int sinkhorn_iters,
                      odam_assoc** out);
int odam_assoc_destroy(odam_assoc* m);
/* state_dict entries under the reference's key names ("encoder.0.weight", "fuser.layers.0.attn.proj.1.bias",
 * "gnn.layers.3.mlp.2.weight", "final_proj.weight", "bin_score", ...), float32 [host]; plus "pe_div_term" [128]
 * = exp(arange(0, 256, 2) * (-ln(10000) / 256)) as PositionalEncoding computes it (associator.py:317-321). */
int odam_assoc_set_weight(odam_assoc* m, const char* name, const float* data, const long long* shape, int ndim);
int odam_assoc_finalize(odam_assoc* m);

/* tracks [dev][T,79,100] and detections [dev][79,30] in the associator's input layout (channel 0 = frame index,
 * -1 padding; processor.py:49-68, 155-179); Z_out [dev][(T+1),(n_det+1)] = log assignment incl. dustbins.
 * tracks, detections and Z_out may also be mapped pinned HOST memory (the kernels then read / write it themselves: no copy
 * commands; what odam_amd.associator does inside its checked call).
 * Stream-ordered, no host synchronisation.  The encoder and the per-track fuser layers are launches on 100 T + 30 rows; the
 * matching GNN layers and the final projection run as ONE persistent launch of 256 workgroups -- by default with the rows dealt
 * to the 8 XCDs: one device-wide barrier per layer (q | k | v exchanged through sc1 accesses), the row-wise stages behind
 * XCD-local barriers (assoc.persist = 2); assoc.persist = 1: five device-wide barriers per layer, bit-identical results --,
 * then the score matrix and the Sinkhorn kernel.
 * The persistent launch is used only when the runtime says all its workgroups can be resident at once on this device
 * (odam_assoc_info; never under a CU mask); the odam_config key assoc.persist = 0 (read when a handle is finalised) or
 * odam_assoc_set_persistent(m, 0) per handle issue the matching layers as separate launches instead.  The library reads no
 * environment variable.
 * One handle serves one stream at a time.
 * Residency can still be lost to a foreign kernel that holds CUs: every barrier wait is bounded (20 ms by default); a launch
 * that hits the bound is abandoned, Z_out is filled with NaN and the handle's lost-launch counter moves.  The caller checks
 * odam_assoc_lost_launches after synchronising the stream and re-runs that frame with odam_assoc_forward_sequence (same
 * arguments, no residency assumption) -- odam_amd/associator.py does. */
int odam_assoc_forward(odam_assoc* m, const float* tracks, int T, const float* detections, int n_det, float* Z_out,
                       void* stream);

int odam_assoc_forward_sequence(odam_assoc* m, const float* tracks, int T, const float* detections, int n_det, float* Z_out,
                                void* stream);
/* count [host] = persistent launches abandoned at a barrier since odam_assoc_finalize (a pinned host word the device bumps:
 * meaningful after the stream of the forward in question has been synchronised; no device call is made) */
int odam_assoc_lost_launches(odam_assoc* m, unsigned* count);
/* on != 0: use the persistent launch where the residency gate allows it; 0: the launch sequence */
int odam_assoc_set_persistent(odam_assoc* m, int on);
/* persistent: the path the next forward takes; resident_capacity: workgroups of the persistent kernel this device holds at
 * once (occupancy query for the compiled kernel, less one per CU of margin, times the CUs; 0 under a CU mask);
 * workgroups: what the launch needs.  Any pointer may be null. */
int odam_assoc_info(odam_assoc* m, int* persistent, int* resident_capacity, int* workgroups);
/* bound of one barrier wait in microseconds (default 20000); 0 abandons every persistent launch at its first barrier
 * (tests of the fallback path) */
int odam_assoc_set_barrier_timeout_us(odam_assoc* m, long long us);
/* tests: with on != 0 the row-partitioned matching kernel (odam_config assoc.persist = 2) treats its workgroup placement as wrong
 * -- the check it makes on every launch (the 32 workgroups of a group must run behind one L2: hardware XCC id compared inside the
 * group) -- so the launch is abandoned, counted in odam_assoc_lost_launches and the frame re-run through the launch sequence */
int odam_assoc_debug_misplace(odam_assoc* m, int on);

/* ---- device-resident track windows: the device side of OdamProcess._preprocess_tracks (src/processor.py:133-170) ----
 * The observations of every live track stay on the device in the world frame (14 float64 each: frame id, class, bbox px x4,
 * dims x3, t_wo x3, az_wo, score = columns 0-13 of a track row, layout src/processor.py:98-108); _build writes the
 * associator's track input [T][79][window] float32 for the current camera from them.
 *   _append: n <= 32 observations of one frame; track_ids [host][n] (every id < max_tracks; a new track simply uses the next
 *            id), rows14 [host][n][14].  Within one call every id occurs at most once (a frame adds one observation per track).
 *   _build:  proj_px [dev][T][4] float64 (odam_sq_project_extents), cam15 [host][15] = rows 0..2 of inv(T_wc) (12), the camera's
 *            azimuth, img_w, img_h; out [dev][T][79][window].
 *   _load:   bulk (re)build -- lengths [host][T] observations per track, rows14 [host][sum_t min(lengths[t], window)][14] = the
 *            last min(length, window) observations of track 0, then of track 1, ...; leaves the store exactly as lengths[t]
 *            appends per track would.  One upload, one launch; synchronises the stream.
 * Stream-ordered; host arrays are copied before the call returns (_append stages them in a pinned ring of eight slots whose
 * reuse waits for the upload issued from that slot, however far behind the stream is). */
typedef struct odam_trackwin odam_trackwin;
int odam_trackwin_create(int max_tracks, int window, odam_trackwin** out);
int odam_trackwin_destroy(odam_trackwin* w);
int odam_trackwin_reset(odam_trackwin* w, void* stream);
int odam_trackwin_append(odam_trackwin* w, int n, const int* track_ids, const double* rows14, void* stream);
int odam_trackwin_load(odam_trackwin* w, int T, const int* lengths, const double* rows14, void* stream);
int odam_trackwin_build(odam_trackwin* w, int T, const double* proj_px, const double* cam15, float* out, void* stream);

/* The host side of OdamProcess._prepare_tracks (src/processor.py:172-180) inside the store: every _append also extends running sums of
 * the track's centre, dimensions (row after row -- numpy's axis-0 reduce) and keeps its azimuths (reduced pairwise, as numpy reduces
 * a 1-D array), so the parameter rows a frame needs -- mean centre, mean azimuth, sqrt(max(mean dims, 0.05) / 2), exponents -0 -- come
 * without touching the host track list.
 *   _load_sums:    after _load: cols7 [host][sum_t lengths[t]][7] = columns 6..12 of EVERY observation, track after track.
 *   _params:       out [host][T][9] float32, bit for bit what the numpy code of the reference computes; code 4 if the sums do not
 *                  cover exactly T tracks (appends skipped an id, or a _load without _load_sums).
 *   _build_tracks: _params -> upload -> odam_sq_project_extents on `sq` (include/odam_sq.h) -> _build: the whole device chain of a
 *                  frame's track input in one call; T_cw12_K9 [host][21] as odam_sq_project_extents takes it. */
struct odam_sq_ctx;
int odam_trackwin_load_sums(odam_trackwin* w, int T, const int* lengths, const double* cols7);
int odam_trackwin_params(odam_trackwin* w, int T, float* out);
int odam_trackwin_build_tracks(odam_trackwin* w, struct odam_sq_ctx* sq, int T, const double* T_cw12_K9, double cam_azi, double img_w,
                               double img_h, float* out, void* stream);

/* One frame of OdamProcess.process_frame's device work behind one call (src/processor.py:320-337): _append of the n_app <= 30
 * observations the PREVIOUS frame attached (app_ids / app_rows14 as in odam_trackwin_append; n_app = 0: none), _build_tracks for this
 * frame's camera into tracks_out [dev][T][79][window], odam_assoc_forward on it.  The same entry points in the same order -- results are
 * theirs bit for bit --, without the host time between three binding calls.  Stream-ordered, no host synchronisation; the caller
 * synchronises, checks odam_assoc_lost_launches and, if it moved, re-runs odam_assoc_forward_sequence on tracks_out. */
int odam_assoc_step(odam_assoc* m, odam_trackwin* w, struct odam_sq_ctx* sq, int n_app, const int* app_ids, const double* app_rows14,
                    int T, const double* T_cw12_K9, double cam_azi, double img_w, double img_h, const float* detections, int n_det,
                    float* tracks_out, float* Z_out, void* stream);

/* diagnostics of the persistent matching kernel: enable != 0 makes the following forwards record a 100 MHz timestamp of
 * workgroup 0 at kernel start and after every stage (5 per GNN layer, final projection, score matrix, Sinkhorn loop);
 * out [host][n <= 128], if not null, receives the stamps of the last forward (synchronises the device). */
int odam_assoc_stage_stamps(odam_assoc* m, int enable, unsigned long long* out, int n);

/* diagnostics: copies a workspace buffer of the last forward to out [host][n]: which = 0 the row block X [(T+30), 512],
 * 1 the descriptors [(T+30), 256], 2 the score matrix [T, 32] (synchronises the device) */
int odam_assoc_debug_read(odam_assoc* m, int which, float* out, long long n);

/* hungarian_matching (src/models/associator.py:19-35) on the device, stream-ordered: the assignment scipy.optimize.linear_sum_assignment
 * returns for the cost matrix 1 - scores (its shortest-augmenting-path solver restated with its tie order, binary64 duals), scores = exp(Z)
 * when log_domain != 0 (Z = the log assignment of odam_assoc_forward without its dustbin row / column: n_tracks x n_det, row stride ldz) or
 * Z itself; match_out [n_det] (device or pinned host) receives the track index of every detection whose matched score exceeds threshold,
 * -1 otherwise; *status (device or pinned host) 0, or 1 for an infeasible matrix.  Up to 32 x 128 after the solver's transposition
 * (min(n_tracks, n_det) <= 32, max <= 128), ODAM_E_LIMIT (3) beyond: the caller then solves on the host. */
int odam_assoc_hungarian(const float* Z, int n_tracks, int n_det, int ldz, double threshold, int log_domain, int* match_out, int* status,
                         void* stream);

/* hungarian_matching (src/models/associator.py:19-35) and the score test of _attach_to_tracks (src/processor.py:228-231) on the HOST, no device
 * work: score [host] = exp of the log assignment, (n_tracks + 1) x (n_det + 1) float32 with row stride lds (the last row / column are the
 * dustbins); match_out [n_det] receives the track of every detection whose matched score exceeds match_threshold (scipy's
 * linear_sum_assignment(1 - score[:-1, :-1]) restated sequentially with its tie order), -1 otherwise; keep_out [n_det] is 0 where the
 * score of the detection's pair -- the dustbin row for an unmatched one -- is below score_threshold (both comparisons in float32, as numpy and
 * torch compare a float32 array with a Python float).  4: NaN / -inf in the cost matrix, 5: infeasible (scipy raises ValueError for both). */
int odam_assoc_attach(const float* score, int n_tracks, int n_det, int lds, double match_threshold, double score_threshold, int* match_out,
                      unsigned char* keep_out);

/* log_optimal_transport alone: scores [dev][m, n] with row stride lds, alpha = bin_score */
int odam_assoc_sinkhorn(const float* scores, int lds, int m, int n, float alpha, int iters, float* Z_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
