// ck_common.h -- internal declarations shared by the HIP translation units of libck_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <exception>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/camkifu_amd.h"

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct TimingSlot {
    double ms = 0;
    int launches = 0;
};

struct PendingEvent {
    hipEvent_t a, b;
    std::string name;
};

struct Mog2State {
    int h = 0, w = 0, nframes = 0;
    DevBuf weight, variance, mean, nmodes;
    bool alive = false;
    // learning rates of an ordered run: the model's own pinned staging + device copy (k_mog2_run)
    float* rates_host = nullptr;
    float* rates_dev = nullptr;
    size_t rates_cap = 0;
    hipEvent_t rates_done = nullptr;
    bool rates_busy = false;
};

// the modes that carry f32 operands as split fp16 (and share the overflow flag and the f32 fallback)
static inline bool ck_cnn_split(int mode) { return mode == CK_CNN_F16X2 || mode == CK_CNN_F16Q8; }

struct CnnWeights {
    bool set = false;
    // repacked fp32 (correlation layout, [kh][kw][cin][cout] with the flip applied)
    DevBuf c1w, c1b, c2w, c2b, c3w, c3b, c4w, c4b, d1w, d1b, d2w, d2b;
    // bf16 packs for the MFMA path
    DevBuf c2w_bf, c3w_bf, c4w_bf, d1w_bf;
    DevBuf c1w_f16;      // conv1 for the fused bf16 kernels: fp16 fragments (k_cnn_bf16.hip)
    DevBuf d1w_bfp;      // dense 1 for fc1_bf16_kernel: bf16 fragments over the padded maps
    // hi / lo fp16 planes for the split-precision mode
    DevBuf c1w_h2, c2w_h2, c3w_h2, c4w_h2, d1w_h2;
    // CK_CNN_F16Q8 (k_cnn_q8.hip): conv1 as fp16 fragments of both planes, the cross-term weights of conv2 .. conv4 as e4m3
    DevBuf c1w_q8, c2x_q8, c3x_q8, c4x_q8;
    bool q8_ok = false;  // every weight inside the e4m3 range of its block scale (else the mode runs the three-MFMA kernels)
};

struct ck_ctx {
    // One thread at a time (include/camkifu_amd.h): `owner` is the token of the thread inside an entry point, 0 when
    // nobody is; a second thread gets CK_ERR_STATE instead of a silent race on the stream and the scratch buffers.
    std::atomic<unsigned long long> owner{0};
    int depth = 0;                   // entry points calling entry points on the owner's thread
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool timing = false;
    std::map<std::string, TimingSlot> slots;
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t handover = nullptr;   // ck_stream_wait: recorded on the host framework's stream, waited for on ours

    // scratch (grown on demand, never shrunk)
    DevBuf in_stage;     // staged host input
    DevBuf in_stage2;
    DevBuf planes;       // median output, planar n*3*h*pitch
    DevBuf trange;       // value bounds of the median's tiles (ck_range_bytes)
    DevBuf edges;        // n*h*w
    DevBuf map;          // n*h*w NMS map
    DevBuf labels;       // n*h*w int32 union-find parents
    DevBuf labels2;
    DevBuf runs;         // run-table labelling: bit map, rank prefix, row bases, run parents
    DevBuf ghost;        // n*h*w
    DevBuf misc;         // small per-frame counters
    DevBuf bflag;        // per frame: did Canny put an edge pixel on the image frame? (K2 -> K3 label reuse)
    DevBuf comp;         // per-frame component tables
    DevBuf lists;        // per-frame edge / border pixel lists
    DevBuf pts;          // compacted border points
    DevBuf accum;        // hough accumulators
    DevBuf peaks;
    DevBuf goban;        // n*380*380*3
    DevBuf act0, act1, act2;   // cnn activations
    DevBuf ybuf, lblbuf, confbuf, rlblbuf, rconfbuf, fgcbuf;
    DevBuf out_stage;
    DevBuf mats;
    void* host_pinned = nullptr;
    size_t host_pinned_cap = 0;
    void* host_pinned2 = nullptr;    // second arena (contour survey) so that a caller's data in the first one survives
    size_t host_pinned2_cap = 0;
    // per-frame result records (ck_board_detect_records / ck_cnn_regions_records): pinned staging of one half of n records,
    // and its device twin for records that live in HBM
    void* rec_host = nullptr;
    size_t rec_host_cap = 0;
    DevBuf rec_stage;

    CnnWeights cnn;
    int* cnn_flag_host = nullptr;    // host-mapped flag: the split-precision kernels met a value outside the fp16 range
    int* cnn_flag_dev = nullptr;
    int cnn_fallbacks = 0;           // batches the split-precision mode handed back to the f32 kernels
    int cnn_mode = CK_CNN_F16X2;     // f32-accurate and 2.3x faster than the k-ordered f32 chain (CK_CNN_FP32)
    std::vector<Mog2State> mog2;
};

extern thread_local std::string g_ck_create_error;

int ck_fail(ck_ctx* ctx, int code, const char* fmt, ...);
int ck_ensure(ck_ctx* ctx, DevBuf& b, size_t bytes);
int ck_ensure_pinned(ck_ctx* ctx, size_t bytes, int which = 0);

#define CK_HIP(ctx, call)                                                                 \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess)                                                            \
            return ck_fail((ctx), CK_ERR_HIP, "%s failed: %s (%s:%d)", #call,             \
                           hipGetErrorString(e__), __FILE__, __LINE__);                   \
    } while (0)

// ---- entry-point bracket: the one-thread-per-context contract, enforced, and nothing thrown across the C ABI ----
unsigned long long ck_thread_token();
void ck_note_busy(const ck_ctx* ctx);              // ck_last_error(ctx) on this thread then explains the refusal
struct CtxCall {
    ck_ctx* ctx;
    int code = CK_OK;
    explicit CtxCall(ck_ctx* c) : ctx(c)
    {
        if (!ctx) { code = CK_ERR_ARG; return; }
        const unsigned long long me = ck_thread_token();
        unsigned long long nobody = 0;
        if (ctx->owner.load(std::memory_order_acquire) == me) { ctx->depth++; return; }
        if (!ctx->owner.compare_exchange_strong(nobody, me, std::memory_order_acq_rel)) {
            ck_note_busy(ctx);
            code = CK_ERR_STATE;
            ctx = nullptr;
            return;
        }
        ctx->depth = 1;
        // the HIP device is per-thread state: whichever host thread drives this context now (a lane's worker, the
        // exchange thread of rank r > 0 ...) works on the context's device from here on
        if (hipSetDevice(ctx->device) != hipSuccess) {
            ctx->depth = 0;
            ctx->owner.store(0, std::memory_order_release);
            code = CK_ERR_HIP;
            ctx = nullptr;
        }
    }
    ~CtxCall()
    {
        if (ctx && --ctx->depth == 0) ctx->owner.store(0, std::memory_order_release);
    }
};
#define CK_API_BEGIN(ctx)                         \
    CtxCall call__(ctx);                          \
    if (call__.code != CK_OK) return call__.code; \
    try {
#define CK_API_END(ctx)                                                                                   \
    } catch (const std::exception& e__) {                                                                 \
        return ck_fail((ctx), CK_ERR_STATE, "C++ exception inside the library: %s", e__.what());          \
    } catch (...) {                                                                                       \
        return ck_fail((ctx), CK_ERR_STATE, "unknown C++ exception inside the library");                  \
    }

#define CK_TRY(call)                     \
    do {                                 \
        int rc__ = (call);               \
        if (rc__ != CK_OK) return rc__;  \
    } while (0)

#include "ck_pool.h"       // ck_parallel_for: host loops on the library's worker threads

// timing brackets: record HIP events on the ctx stream around a group of launches
struct TimeScope {
    ck_ctx* ctx;
    int idx = -1;
    TimeScope(ck_ctx* c, const char* name);
    ~TimeScope();
};
int ck_timing_collect(ck_ctx* ctx);

// bring a possibly-host input to the device (returns device pointer in *dev)
int ck_to_device(ck_ctx* ctx, const void* src, size_t bytes, int space, DevBuf& stage, const void** dev);
// deliver a device result to a possibly-host output
int ck_from_device(ck_ctx* ctx, void* dst, const void* dev, size_t bytes, int space);

static inline int ck_pitch(int w) { return (w + 63) & ~63; }

// ---- kernels (one launcher per stage; all asynchronous on ctx->stream) -----------------
// d_range (nullable): per (frame, channel, CK_RANGE_TILE^2 tile) two bytes lo, hi with lo <= every median of the tile <= hi
// (n * 3 * ceil(h / T) * ceil(w / T) * 2 bytes; what k_canny_planar takes to skip its flat tiles)
#define CK_RANGE_TILE 48
#ifndef CK_TILE_RANGE
#define CK_TILE_RANGE 1        // 0: no bounds written, no tile skipped (A/B builds)
#endif
static inline size_t ck_range_bytes(int n, int h, int w)
{
    return (size_t)n * 3 * ((h + CK_RANGE_TILE - 1) / CK_RANGE_TILE) * ((w + CK_RANGE_TILE - 1) / CK_RANGE_TILE) * 2;
}
int k_median15_planar(ck_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w, uint8_t* d_planes, int pitch, uint8_t* d_range = nullptr);
int k_gray_hist(ck_ctx* ctx, const uint8_t* d_planes, int n, int h, int w, int pitch, int* d_hist);
int k_median_planar(ck_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w, int ksize, uint8_t* d_planes, int pitch, uint8_t* d_range = nullptr);
int k_planar_to_interleaved(ck_ctx* ctx, const uint8_t* d_planes, int n, int h, int w, int pitch, uint8_t* d_out);
int k_interleaved_to_planar(ck_ctx* ctx, const uint8_t* d_in, int n, int h, int w, int pitch, uint8_t* d_planes);
// canny: planar 3-channel input -> map (0/1/2) -> edges (0/255); labels = scratch n*h*w int32
// d_border_flag (nullable, n ints): set to 1 for frames that have an edge pixel on the image frame
int k_canny_planar(ck_ctx* ctx, const uint8_t* d_planes, int n, int h, int w, int pitch, int low, int high,
                   uint8_t* d_map, int32_t* d_labels, uint8_t* d_edges, uint8_t* d_map_out, int* d_border_flag = nullptr,
                   const int* d_thr = nullptr /* per-frame (low, high) pairs on the device, override low / high */,
                   const uint8_t* d_range = nullptr /* value bounds of the planes' tiles, from k_median_planar */);
int k_i420_to_bgr(ck_ctx* ctx, const uint8_t* d_i420, int n, int h, int w, uint8_t* d_bgr);
int k_warp(ck_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w, const double* d_minv, int m_count,
           int dsize, uint8_t* d_out);
int k_board_lines(ck_ctx* ctx, const uint8_t* d_edges, int n, int h, int w, int hough_thresh,
                  float* lines, int cap, ck_board_result* res, uint8_t* d_ghost_out,
                  const int* d_canny_border_flag = nullptr);
// d_nonfinite (nullable): set to 1 when a softmax came out inf / NaN
// one external contour of an edge map (k_contour_survey): first pixel (raster index = discovery key), length of the
// CHAIN_APPROX_SIMPLE vertex list, outer-border pixels as x0, y0, x1, y1, ...
struct CkContour {
    int root = 0;
    int nvert = 0;
    std::vector<int32_t> pts;
};
int k_contour_survey(ck_ctx* ctx, const uint8_t* d_edges, int n, int h, int w, std::vector<std::vector<CkContour>>& out);
int ck_goban_canny_dev(ck_ctx* ctx, const uint8_t* d_in, int n, int h, int w, uint8_t* d_edges, double* otsu_out);   // ck_api.hip
int k_contour_stones(ck_ctx* ctx, const uint8_t* d_goban, const uint8_t* d_fg, int n, int side, const int32_t* rects,
                     int rs, int re, int cs, int ce, uint8_t* stones, int16_t* zones_out, uint8_t* mask_out);
double ck_otsu_level(const int* hist, size_t npx);                        // ck_api.hip: getThreshVal_Otsu_8u restated
// StonesFinder.find_intersections, device half: Canny of the grey image + HoughLinesP of the 361 zones -> host tables
// (*lines_out / *nlines_out point into the context's pinned host arena: valid until the next call on this context)
int k_grid_lines(ck_ctx* ctx, const uint8_t* d_goban, int n, int side, const int32_t* rects, const int16_t** lines_out,
                 const int32_t** nlines_out, uint8_t* edges_out);
int k_cnn_predict(ck_ctx* ctx, const uint8_t* d_goban, int n, float* d_y, uint8_t* d_labels, double* d_conf,
                  int* d_nonfinite = nullptr, uint8_t* d_rlabel = nullptr, double* d_rconf = nullptr);
int k_cnn_pack_weights(ck_ctx* ctx, const float* const w[12], int space);
// CK_CNN_BF16 (k_cnn_bf16.hip): conv1 + conv2 and conv3 + conv4 of np patches, pooled maps out as bf16
int k_cnn_bf16_pack_conv1(ck_ctx* ctx, const float* k1, DevBuf& dst);
int k_cnn_bf16_convs(ck_ctx* ctx, const uint8_t* gob, int np, uint16_t* p2, uint16_t* q4);
int k_cnn_bf16_pack_fc1(ck_ctx* ctx, const float* w, DevBuf& dst);
int k_cnn_bf16_fc1(ck_ctx* ctx, const uint16_t* q4, int np, float* h1);
// CK_CNN_F16Q8 (k_cnn_q8.hip): the same two fused kernels with f32 maps in and out (drop-ins for the split-precision pair of k_cnn.hip)
int k_cnn_q8_pack(ck_ctx* ctx, const float* k1, const float* k2, const float* k3, const float* k4);
int k_cnn_q8_conv12(ck_ctx* ctx, const uint8_t* gob, int np, float* p2, int* overflow);
int k_cnn_q8_conv34(ck_ctx* ctx, const float* p2, int np, float* p4, int* overflow);
int k_mog2_apply(ck_ctx* ctx, Mog2State& st, const uint8_t* d_img, double lr, uint8_t* d_fg);
int k_mog2_run(ck_ctx* ctx, Mog2State& st, const uint8_t* d_gobans, int n, const double* learning_rates,
               int32_t* d_fgcount, uint8_t* d_last_fg, int skip_row, int skip_col);
int k_zone_counts(ck_ctx* ctx, const uint8_t* d_mask, int n, int side, int32_t* d_fgcount);
// per-frame records in HBM (k_records.hip): the board half from n packed 536-byte parts, the stones half from the classifier's
// contiguous region outputs
int k_records_put_board(ck_ctx* ctx, const uint8_t* d_parts, int n, ck_frame_record* d_rec);
int k_records_put_regions(ck_ctx* ctx, const uint8_t* d_rlabel, const double* d_rconf, int n, ck_frame_record* d_rec);

// host geometry (ck_host_geom.cpp)
void ck_invert3x3(const double* s, double* d);
void ck_min_area_rect(const int32_t* pts, int n, float* out_wh);
void ck_min_area_rect_box(const int32_t* pts, int n, float* out_wha);     // + angle in degrees (cv2.minAreaRect's box[2])
std::vector<int32_t> ck_hull_points(const int32_t* pts, int n);           // strictly convex hull, x0, y0, x1, y1, ...
