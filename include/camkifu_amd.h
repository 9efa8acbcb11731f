/*
 * camkifu_amd.h -- C-ABI of the MI355X-native CamKifu vision hot path (libck_hip.so).
 *
 * Drop-in boundary: the reference (ArnaudPel/CamKifu) is pure Python and reaches all of
 * its arithmetic through cv2 / Keras calls made from BoardFinderAuto._detect and
 * StonesFinder._doframe / SfNeural._find.  Each entry point below replaces one such call
 * site (cited as file:line under the reference's src/camkifu/), takes plain pointers and
 * sizes, and is what a ctypes binding in the reference would bind.
 *
 * Conventions
 *   - every function returns CK_OK (0) or a CK_ERR_* code; nothing throws or aborts across
 *     the boundary; ck_last_error() gives the message for the last failing call on a ctx.
 *   - images are uint8, row-major, channels interleaved (BGR) exactly as cv2 hands them,
 *     no row padding (stride = w * channels); batches are n such images back to back.
 *   - `*_space` arguments say where a pointer lives: CK_HOST or CK_DEVICE (HBM).  Input
 *     pointers are borrowed for the duration of the call only.
 *   - one ck_ctx per finder instance / thread: it owns a HIP stream and scratch buffers
 *     and is not re-entrant; different contexts may be used concurrently.
 *   - calls are synchronous: results are complete when the call returns.
 */
#ifndef CAMKIFU_AMD_H
#define CAMKIFU_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { CK_OK = 0, CK_ERR_ARG = 1, CK_ERR_HIP = 2, CK_ERR_CAPACITY = 3, CK_ERR_STATE = 4 };
enum { CK_HOST = 0, CK_DEVICE = 1 };
enum { CK_BACKEND_HIP = 1 };
enum { CK_CNN_FP32 = 0, CK_CNN_BF16 = 1, CK_CNN_F16X2 = 2, CK_CNN_F16Q8 = 3 };

/* per-frame status of the board path, mirrors the early exits of
 * BoardFinderAuto._detect (board/bf_auto.py:76-82) */
enum { CK_BOARD_LINES = 0,        /* lines were searched (n_lines may be 0)          */
       CK_BOARD_NO_CONTOUR = 1,   /* len(contours) == 0                 bf_auto.py:76 */
       CK_BOARD_TOO_SMALL = 2 };  /* not frame_area/3 < biggest.area     bf_auto.py:82 */

typedef struct ck_ctx ck_ctx;

typedef struct ck_board_result {
    int32_t status;        /* CK_BOARD_*                                              */
    int32_t n_contours;    /* len(contours) returned by findContours(RETR_EXTERNAL)   */
    int32_t n_lines;       /* number of Hough lines found (may exceed the caller cap) */
    int32_t reserved;
    double  biggest_area;  /* sorted_boxes[-1].area (minAreaRect w*h)                 */
} ck_board_result;

/* ---- context ------------------------------------------------------------------------ */
int  ck_ctx_create(int device, ck_ctx** out);
/* The same with a HIP stream priority for the context's stream: 0 normal, 1 the device's highest, -1 its lowest.  A finder
 * whose calls are short and waited for (the board finder of the hold-off-aware file mode: a few frames per call, the fold
 * waits for the answer) gets its kernels onto the CUs ahead of the long launches of other contexts. */
int  ck_ctx_create_prio(int device, int priority, ck_ctx** out);
void ck_ctx_destroy(ck_ctx* ctx);
/* The same with a status: CK_OK when the context was freed; CK_ERR_STATE when another thread was still inside one of its
 * calls after 5 s -- the context is then NOT freed (nothing is pulled from under that thread) and the handle stays valid:
 * call again later.  ck_ctx_destroy is this call with the status dropped (it prints a line to stderr instead). */
int  ck_ctx_destroy2(ck_ctx* ctx);
/* Stream-ordered hand-over of device memory from a host framework (PyTorch: the caller passes torch's CURRENT stream, as a
 * hipStream_t): everything queued on `stream` up to now runs before anything this context launches from now on
 * (hipEventRecord on `stream`, hipStreamWaitEvent on the context's).  No host wait.  A host that allocates its outputs or
 * produces its inputs on a stream of its own calls this before it passes their pointers to an entry point below; results
 * need no call in the other direction: every entry point returns with its work complete. */
int  ck_stream_wait(ck_ctx* ctx, void* stream);
const char* ck_last_error(const ck_ctx* ctx);     /* ctx may be NULL: last create error */
int  ck_backend(const ck_ctx* ctx);               /* CK_BACKEND_HIP                     */
int  ck_version(void);
void* ck_stream(ck_ctx* ctx);                     /* the hipStream_t the ctx launches on */
/* kernel-time accounting with HIP events on the ctx stream (used by bench.py) */
int  ck_timing_enable(ck_ctx* ctx, int on);
int  ck_timing_reset(ck_ctx* ctx);
/* name: "median", "canny_nms", "ccl", "hough_vote", "warp", "cnn", ...; returns total ms
 * and number of launches recorded since the last reset */
int  ck_timing_get(ck_ctx* ctx, const char* name, double* total_ms, int* launches);

/* ---- K1  cv2.medianBlur(frame, 15)                              board/bf_auto.py:72 */
int ck_median15(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space,
                uint8_t* out, int out_space);
/*          cv2.medianBlur(img, ksize), ksize odd in 3..17 (13 and 7 are SfContours.get_canny's,
 *          stone/sf_contours.py:336-337); same exact median, replicate border */
int ck_median(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int ksize, int in_space,
              uint8_t* out, int out_space);

/* ---- K2  cv2.Canny(median, low, high)  3-channel, aperture 3, L1  board/bf_auto.py:73
 * map_out (optional, same space as edges): NMS map before hysteresis
 * (0 candidate, 1 suppressed, 2 strong). */
int ck_canny(ck_ctx* ctx, const uint8_t* img3, int n, int h, int w, int in_space,
             int low, int high, uint8_t* edges, uint8_t* map_out, int out_space);

/* ---- SfContours.get_canny(img)                 stone/sf_contours.py:332-340
 *   median = medianBlur(medianBlur(img, 13), 7); otsu = threshold(cvtColor(median, BGR2GRAY), 12, 255, THRESH_OTSU)[0];
 *   return Canny(median, otsu / 2, otsu)
 * edges: n*h*w bytes {0,255}; otsu_out (nullable, host): n doubles, the Otsu levels used */
int ck_goban_canny(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space,
                   uint8_t* edges, int out_space, double* otsu_out);

/* ---- K1+K2 fused pipeline: the "filter pass"                  board/bf_auto.py:72-73 */
int ck_board_edges(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space,
                   uint8_t* edges, int out_space);

/* ---- K3..K6  findContours(RETR_EXTERNAL) -> 3 biggest minAreaRect -> drawContours
 *      -> HoughLines(1, pi/180, hough_thresh)          board/bf_auto.py:75-84, 105-133
 * lines: host, n * cap * 2 floats (rho, theta) in OpenCV's order; res: host, n entries.
 * ghost_out optional (n*h*w).  hough_thresh < 0 means int(min(h,w)/5). */
int ck_board_lines(ck_ctx* ctx, const uint8_t* edges, int n, int h, int w, int in_space,
                   int hough_thresh, float* lines, int cap, ck_board_result* res,
                   uint8_t* ghost_out, int ghost_space);

/* ---- K1..K6 stateless core of BoardFinderAuto._detect        board/bf_auto.py:72-84 */
int ck_board_detect(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space,
                    int hough_thresh, float* lines, int cap, ck_board_result* res);

/* ---- frame source: what cv2.VideoCapture.read() hands every consumer   core/vmanager.py:506-509, 584
 * (the reference decodes to BGR on the CPU).  A file reader that holds planar YUV 4:2:0 frames
 * (I420: Y h*w, U, V (h/2)*(w/2) each; e.g. a .y4m file) uploads 1.5 B/px and converts in HBM:
 * BT.601 studio range, the 20-bit fixed-point arithmetic of cv2.cvtColor(COLOR_YUV2BGR_I420).
 * i420: n frames of h*w*3/2 bytes; bgr: n x h x w x 3.  h and w must be even. */
int ck_i420_to_bgr(ck_ctx* ctx, const uint8_t* i420, int n, int h, int w, int in_space,
                   uint8_t* bgr, int out_space);

/* ---- K7  cv2.getPerspectiveTransform(src4, dst4)            board/boardfinder.py:43-45
 * host only; src/dst 4x2 float32, M 3x3 float64 row-major. */
int ck_get_perspective_transform(const float* src4, const float* dst4, double* M9);

/* ---- K8  cv2.warpPerspective(frame, M, (dsize,dsize))         stone/stonesfinder.py:140
 * M: host, m_count x 9 doubles (m_count == 1: shared by all frames, else == n). */
int ck_warp_perspective(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space,
                        const double* M, int m_count, int dsize,
                        uint8_t* out, int out_space);

/* ---- K9  BackgroundSubtractorMOG2(detectShadows=False).apply stone/stonesfinder.py:113-115,171-176
 * one model per stream; stateful, frames must be applied in order. */
int ck_mog2_create(ck_ctx* ctx, int h, int w, int* handle);
int ck_mog2_apply(ck_ctx* ctx, int handle, const uint8_t* img3, int in_space,
                  double learning_rate, uint8_t* fgmask, int out_space);
int ck_mog2_destroy(ck_ctx* ctx, int handle);

/* ---- K10..K12 stone classifier                stone/nn_manager.py:216-298, nn_cache.py:16-52
 * weights: 12 float32 arrays in Keras-1 'tf' layout, order
 *   c1w[5,5,3,32] c1b c2w[5,5,32,32] c2b c3w[3,3,32,90] c3b c4w[3,3,90,90] c4b
 *   d1w[3240,160] d1b d2w[160,81] d2b
 * (conv kernels are applied as true convolutions, as Keras-1 on Theano does).
 * `space` may be CK_DEVICE: e.g. data_ptr() of PyTorch-ROCm tensors. */
int ck_cnn_set_weights(ck_ctx* ctx, const float* const weights[12], int space);
int ck_cnn_set_mode(ck_ctx* ctx, int mode);      /* CK_CNN_F16X2 (default): f32 operands split into hi + lo fp16, three fp16 MFMAs per product,
                                                    f32 accumulate -- as close to a float64 evaluation as the f32 chain is (profiles/r01_cnn_precision.txt);
                                                    CK_CNN_FP32: k-ordered f32 MFMA chain; CK_CNN_BF16: bf16 operands;
                                                    CK_CNN_F16Q8 (opt-in): the split of CK_CNN_F16X2 with its two cross terms (2^-11 of a product)
                                                    rounded to e4m3 and issued as one block-scaled MFMA per two taps -- a quarter faster, the
                                                    filter maps within 5e-5 of their scale instead of 1e-6; out-of-range values fall back */
/* goban: n x 380 x 380 x 3.  Any of y (n*100*81 softmax), labels (n*361, 0=E 1=B 2=W),
 * conf (n*361 doubles, max(y)/sum(y)) may be NULL. */
int ck_cnn_predict(ck_ctx* ctx, const uint8_t* goban, int n, int in_space,
                   float* y, uint8_t* labels, double* conf, int out_space);
/* The classifier's intermediate filter maps, for inspection and parity tests ("intermediate float filter maps within
 * 1e-4"): what the network of create_net (nn_manager.py:280-295) holds after its two MaxPooling2D layers, as computed by
 * the kernels of the context's mode, channels-last float32 in HOST memory:
 *   pool2 (n*100*16*16*32): relu(conv2(relu(conv1(x)))) max-pooled 2x2   (nn_manager.py:281-286)
 *   pool4 (n*100*6*6*90):   ... relu(conv4(relu(conv3(.)))) max-pooled 2x2 (nn_manager.py:287-292), the Flatten input
 * Region order i*10+j as everywhere; either may be NULL; n <= 128 (one chunk of the classifier's frame loop). */
int ck_cnn_maps(ck_ctx* ctx, const uint8_t* goban, int n, int in_space, float* pool2, float* pool4);

/* ---- K8 + K10..K12: frame + M -> 19x19 labels   stonesfinder.py:140 + nn_cache.py:33-41 */
int ck_stones_detect(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space,
                     const double* M, int m_count, uint8_t* labels, double* conf,
                     int out_space);

/* ---- the classifier's answers per REGION: what NNCache.predict_4_stones / predict_stone read (stone/nn_cache.py:16-31).
 * region_label: n*100 argmax labels (0..80, region (i, j) at i*10+j); region_conf: n*100 doubles max(y)/sum(y). */
int ck_cnn_regions(ck_ctx* ctx, const uint8_t* goban, int n, int in_space,
                   uint8_t* region_label, double* region_conf, int out_space);

/* ---- an ORDERED RUN of the stones path over n consecutive frames of one stream, one call:
 *      K8 warp -> K9 MOG2 in frame order -> K10..K12            stone/stonesfinder.py:123-176 + sf_neural.py:37-55
 * mog2_handle < 0: no background model (fgcount untouched).  learning_rates: host, n doubles (0.01 during the
 * first bg_init_frames frames, 0.005 afterwards: stonesfinder.py:171-176).  fgcount: n*361 int32, the number of
 * foreground pixels of the frame's MOG2 mask inside StonesFinder.getrect(r, c) -- what SfNeural.is_agitated sums
 * (sf_neural.py:178-180).  labels / conf (n*361, as ck_stones_detect) may be NULL.  Feeds ck_policy_run. */
int ck_stones_run(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space,
                  const double* M, int m_count, int mog2_handle, const double* learning_rates,
                  uint8_t* region_label, double* region_conf, int32_t* fgcount,
                  uint8_t* labels, double* conf, int out_space);

/* ---- the fixed-size per-frame RESULT RECORD of the fast-file path (SURVEY 8e "Collective": labels + conf + n_lines +
 * lines, Lmax = 64): what the stateless core of one frame leaves for the ordered fold -- the board half is what
 * BoardFinderAuto._detect reads after the image chain (board/bf_auto.py:76-84, 125-135), the stones half what
 * NNCache.predict_4_stones reads (stone/nn_cache.py:16-31).  The two entry points below write their half of n records IN
 * PLACE, in host memory or in HBM (`rec_space`), and leave the other half untouched: with the records in HBM a multi-GPU
 * host gathers them to the folding rank without a host copy on the other ranks (camkifu_amd/pipeline.py).
 * lines beyond min(n_lines, CK_REC_LMAX) are zero; flags: CK_REC_LINES_CUT when n_lines > CK_REC_LMAX (the strongest
 * CK_REC_LMAX lines are kept: OpenCV's order is most votes first). */
#define CK_REC_LMAX 64
enum { CK_REC_LINES_CUT = 1, CK_REC_FAILED = 2 };
typedef struct ck_frame_record {
    int32_t status;                   /* CK_BOARD_*                                      board half: 536 bytes */
    int32_t n_contours;
    int32_t n_lines;
    int32_t flags;                    /* CK_REC_*                                                             */
    double  biggest_area;
    float   lines[CK_REC_LMAX][2];    /* (rho, theta)                                                         */
    double  region_conf[100];         /* max(y) / sum(y), region (i, j) at i*10+j       stones half: 900 bytes */
    uint8_t region_label[100];        /* argmax label 0..80                                                   */
    uint8_t pad[4];
} ck_frame_record;                    /* 1440 bytes, no implicit padding                                      */
/* K1..K6 of n frames -> the board half of rec[0 .. n)                                board/bf_auto.py:72-84, 125-135 */
int ck_board_detect_records(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space,
                            int hough_thresh, ck_frame_record* rec, int rec_space);
/* K10..K12 of n goban images -> the stones half of rec[0 .. n)                         stone/nn_cache.py:16-31 */
int ck_cnn_regions_records(ck_ctx* ctx, const uint8_t* goban, int n, int in_space,
                           ck_frame_record* rec, int rec_space);

/* ---- K9 sharded by PIXEL for multi-GPU batches: the model `handle` (ck_mog2_create(band_h, 380)) holds a horizontal band
 * of the goban image (a whole number of 20-pixel intersection rows; the last band is the one that ends at pixel row
 * 379).  band: n x band_h x 380 x 3, the band of n consecutive goban images in frame order; counts: n x (band_h/20
 * rounded up) x 19 int32 foreground pixels per intersection zone, the same numbers ck_stones_run gives for those rows. */
int ck_mog2_band_run(ck_ctx* ctx, int handle, const uint8_t* band, int n, int in_space,
                     const double* learning_rates, int last_band, int32_t* counts, int out_space);

/* ---- the 361 box sums of SfNeural.mark_targets / select_targets as one reduction   stone/sf_neural.py:72-83, 129-154
 * mask: n x 380 x 380 (non-zero = foreground); counts: n*361 int32 (pixels per getrect zone). */
int ck_zone_counts(ck_ctx* ctx, const uint8_t* mask, int n, int in_space, int32_t* counts, int out_space);

/* ---- SfContours.find_stones for n goban images in one call                         stone/sf_contours.py:48-111
 * (with analyse_fg :207-249, extract_contours_fg :251-300, _filter_contours :186-205, _find_centers :302-330,
 * find_color :128-184).  goban: n x side x side x 3 BGR, fg: n x side x side foreground masks (StonesFinder.get_foreground),
 * both in `in_space`.  rects: HOST, 19*19*4 int32 = StonesFinder.getrect(r, c) as (x0, y0, x1, y1), x along rows -- the
 * caller's grid, so a learnt PosGrid is honoured; rows [rs, re) and columns [cs, ce) are analysed, as the keyword
 * arguments of the reference method say.  Outputs on the HOST: stones n*19*19 (0 E, 1 B, 2 W; E outside the range),
 * zones (nullable) n*(re-rs)*(ce-cs)*4 int16 = the method's `zones` array, mask (nullable) n*hs*ws bytes = the hull
 * mask of the analysed view (hs = x1 - x0, ws = y1 - y0 of the range's corner rectangles).
 * CK_ERR_STATE where the reference itself raises (_find_centers dividing by a zero cell count). */
int ck_contour_stones(ck_ctx* ctx, const uint8_t* goban, const uint8_t* fg, int n, int side, int in_space, const int32_t* rects,
                      int rs, int re, int cs, int ce, uint8_t* stones, int16_t* zones, uint8_t* mask);

/* ---- cv2.findContours(edges, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) as SfContours reads it   stone/sf_contours.py:82, 266
 * For n edge maps (h x w, non-zero = edge): counts[f] = contours of map f; table: one row of 4 int32 per contour, map after
 * map, each map's contours in the order cv2 returns them (last found first): x, y of the contour's first point, the
 * length of its compressed vertex list (cont.shape[0]), the number of outer-border pixels; points (nullable): those
 * pixels as x, y pairs, contour after contour (the set drawContours(thickness=1) paints).  All outputs on the HOST. */
int ck_contours_external(ck_ctx* ctx, const uint8_t* edges, int n, int h, int w, int in_space,
                         int32_t* counts, int32_t* table, int table_cap, int32_t* points, int points_cap);

/* ---- StonesFinder.find_intersections for n goban images in one call                stone/stonesfinder.py:516-552
 * grey image, Otsu level, Canny(gray, level / 2, level), then cv2.HoughLinesP(zone, 1, pi / 180, int(3/4 side),
 * minLineLength=int(2/3 side), maxLineGap=0) in each of the 361 getrect zones and update_grid (:888-947) on what it finds.
 * goban: n x side x side x 3 BGR in `in_space`; mtx: HOST 19*19*2 int16 = PosGrid.mtx; rects: HOST 19*19*4 int32 =
 * StonesFinder.getrect(r, c).  Outputs on the HOST: grid n*19*19*2 int16 = the method's return value (positions negated
 * where a line was found, moved where a cross was found); nullable: lines n*361*CK_ZONE_LINES*4 int16 (x0, y0, x1, y1 per
 * line, in the order found), nlines n*361 int32, edges n*side*side (the Canny map). */
#define CK_ZONE_LINES 32
int ck_find_intersections(ck_ctx* ctx, const uint8_t* goban, int n, int side, int in_space, const int16_t* mtx, const int32_t* rects,
                          int16_t* grid, int16_t* lines, int32_t* nlines, uint8_t* edges);

/* update_grid for one zone (host only, no GPU needed)                                 stone/stonesfinder.py:888-947
 * lines: k x (x0, y0, x1, y1) in the zone's own coordinates, box: getrect of the zone, slot: the intersection's int16
 * (x, y), updated in place.  CK_ERR_ARG for a zero-length line (the reference divides by zero there). */
int ck_update_grid(const int32_t* lines, int k, const int32_t* box, int16_t* slot);

/* ======================================================================================================
 * Ordered (stateful) halves of the two finders -- host only, no GPU needed.  The per-frame finders call
 * them once per frame, the batch pipeline's fold calls them over the gathered per-frame records; the
 * arithmetic is the reference's Python float arithmetic restated operation for operation.
 * ====================================================================================================== */

/* ---- BoardFinderAuto._detect after the image chain: 4-frame line accumulation, group_intersections,
 *      connect_clusters, updt_corners                     board/bf_auto.py:76-102, 143-217; core/imgutil.py:38-68, 216-288, 464-530
 * status/lines/n_lines: this frame's result of ck_board_detect; frame_counter: VidProcessor.total_f_processed;
 * cur_hull: the 4 corners known so far (8 ints, x y) or NULL (GobanCorners.hull is None).
 * Out: *found (the return value of _detect), *update (corners must be replaced by `centers`),
 * centers (up to 4 x y pairs, hull-ordered), *n_centers, stats[2] = {clusters, intersections} or -1 when the
 * frame did not reach the grouping step.  CK_ERR_STATE: the reference would raise here (3-vertex hull indexed as 4). */
/* imgutil.get_ordered_hull (core/imgutil.py:236-288): convex hull of n integer points, clockwise on screen,
 * starting at the vertex nearest the image origin; out holds up to n x y pairs */
int  ck_ordered_hull(const int32_t* pts, int n, int32_t* out, int32_t* n_out);
typedef struct ck_boardfold ck_boardfold;
int  ck_boardfold_create(ck_boardfold** out);
void ck_boardfold_destroy(ck_boardfold* bf);
int  ck_boardfold_reset(ck_boardfold* bf);
int  ck_boardfold_step(ck_boardfold* bf, int h, int w, int status, const float* lines, int n_lines,
                       long long frame_counter, const int32_t* cur_hull, int32_t* found, int32_t* update,
                       int32_t* centers, int32_t* n_centers, int32_t* stats);

/* test hooks: Python's round(x, 10) as the fold computes it (integer arithmetic, exact) and by the long way (the decimal
 * string and back) -- tests hold the two, and CPython's own round, equal */
double ck_round10(double x);
double ck_round10_reference(double x);
/* The same fold over the board halves of n gathered records (ck_frame_record), frames *k_io .. n-1, with the hold-off
 * after a hit as a FRAME COUNT (the reference's 10 s of wall clock, bf_auto.py:43-49, for a file read at file_fps):
 * frames inside the hold-off are not looked at; every other frame goes through ck_boardfold_step with *counter_io as
 * its frame_counter.  order (nullable): frame k's record is recs[order[k]] -- the gather leaves the records rank by
 * rank, frame f at row (f mod world) * rows_per_rank + 1 + f / world, and the fold reads them where they lie.
 * Returns with *k_io == n (batch done) or just after the first frame whose step says `update` (outputs as
 * ck_boardfold_step's): the caller replaces its corners, derives the transform, sets *hold_io after a hit and calls
 * again with the new hull.  A hit WITHOUT update leaves the corners -- hence the transform -- as they are: the fold then
 * starts the hold-off itself (*hold_io = hold_after_same_hit) and goes on; hold_after_same_hit < 0 returns such hits too.
 * seen_looked_io[2]: frames offered / looked at (running totals).  On an error *k_io is the frame that raised it (not
 * counted), as an exception out of _detect leaves total_f_processed. */
int  ck_boardfold_run(ck_boardfold* bf, int h, int w, const ck_frame_record* recs, const int32_t* order, int n, int32_t* k_io,
                      long long* counter_io, int32_t* hold_io, long long* seen_looked_io, const int32_t* cur_hull,
                      int hold_after_same_hit, int32_t* found, int32_t* update, int32_t* centers, int32_t* n_centers,
                      int32_t* stats);

/* ---- SfNeural._find after the classifier: predict_all / mark_targets / select_targets / predict_moves /
 *      get_color_ratio / lookback / HeatPoint                                  stone/sf_neural.py:37-244
 * Runs frames [*frame_io, n) of an ordered run.  Per frame: region_label (100 argmax labels 0..80, region (i, j)
 * at i*10+j), region_conf (100 x max(y)/sum(y)), fgcount (361 foreground-pixel counts of the MOG2 mask over
 * StonesFinder.getrect(r, c); NULL = nothing moves), first_counter = total_f_processed of frame 0 of the run.
 * board: the goban as the controller holds it NOW (361 bytes, 0 E / 1 B / 2 W).
 * The call returns either with *frame_io == n (run finished, *kind == 0) or with a request the caller must apply
 * before calling again with the updated board: *kind 1 = StonesFinder.suggest (one triple), 2 = bulk_update;
 * moves = *n_moves triples (color, row, col); *frame_io / *phase_io say where to resume (pass them back unchanged;
 * start a run with 0 / 0).  cap >= 722 triples. */
typedef struct ck_policy ck_policy;
int  ck_policy_create(int bg_init_frames, ck_policy** out);
void ck_policy_destroy(ck_policy* p);
int  ck_policy_run(ck_policy* p, int n, long long first_counter, const uint8_t* region_label,
                   const double* region_conf, const int32_t* fgcount, const uint8_t* board,
                   int32_t* frame_io, int32_t* phase_io, int32_t* kind, int32_t* moves, int cap, int32_t* n_moves);
/* ck_policy_run reading the classifier's answers from the stones halves of n gathered records where they lie (order as
 * for ck_boardfold_run, nullable; fgcount stays in frame order): same protocol, same results */
int  ck_policy_run_records(ck_policy* p, int n, long long first_counter, const ck_frame_record* recs, const int32_t* order,
                           const int32_t* fgcount, const uint8_t* board,
                           int32_t* frame_io, int32_t* phase_io, int32_t* kind, int32_t* moves, int cap, int32_t* n_moves);
/* inspection / test hooks: any out pointer may be NULL.  targets 361 B, heat_color 361 B (0 = no watched prediction),
 * heat_energy 361 int32, heat_conf 361 doubles, flags[2] = {has_sampled, recolour events seen} */
int  ck_policy_get_state(const ck_policy* p, uint8_t* targets, uint8_t* heat_color, int32_t* heat_energy,
                         double* heat_conf, int32_t* flags);
int  ck_policy_set_state(ck_policy* p, const uint8_t* targets, int has_sampled /* <0: keep */);
int  ck_policy_watch(ck_policy* p, int r, int c, int color /*0 drops it*/, double confidence, long long stamp);

#ifdef __cplusplus
}
#endif
#endif
