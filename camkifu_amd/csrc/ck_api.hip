// ck_api.hip -- C-ABI entry points of libck_hip.so (see include/camkifu_amd.h).
// Host-side glue only: argument checks, staging between host and HBM, and the order in
// which the stage kernels are launched on the context's stream.
#include <stdarg.h>
#include <stddef.h>

#include <cfloat>
#include <cmath>
#include <algorithm>
#include <vector>
#include <thread>
#include <chrono>

#include "ck_common.h"
#include "ck_stonegeom.h"

thread_local std::string g_ck_create_error;
static thread_local const ck_ctx* g_ck_busy_ctx = nullptr;     // the context that refused this thread's last call

unsigned long long ck_thread_token()
{
    static std::atomic<unsigned long long> next{1};
    static thread_local unsigned long long mine = 0;
    if (!mine) mine = next.fetch_add(1);
    return mine;
}
void ck_note_busy(const ck_ctx* ctx) { g_ck_busy_ctx = ctx; }

int ck_fail(ck_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_ck_create_error = buf;
    return code;
}

int ck_ensure(ck_ctx* ctx, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap) return CK_OK;
    if (b.p) { CK_HIP(ctx, hipStreamSynchronize(ctx->stream)); CK_HIP(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    CK_HIP(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return CK_OK;
}

int ck_ensure_pinned(ck_ctx* ctx, size_t bytes, int which)
{
    void*& p = which ? ctx->host_pinned2 : ctx->host_pinned;
    size_t& cap = which ? ctx->host_pinned2_cap : ctx->host_pinned_cap;
    if (bytes <= cap) return CK_OK;
    if (p) { CK_HIP(ctx, hipHostFree(p)); p = nullptr; cap = 0; }
    bytes += bytes / 4;                       // a little headroom: the survey's point count changes from batch to batch
    CK_HIP(ctx, hipHostMalloc(&p, bytes, hipHostMallocDefault));
    cap = bytes;
    return CK_OK;
}

int ck_to_device(ck_ctx* ctx, const void* src, size_t bytes, int space, DevBuf& stage, const void** dev)
{
    if (space == CK_DEVICE) { *dev = src; return CK_OK; }
    if (space != CK_HOST) return ck_fail(ctx, CK_ERR_ARG, "bad memory space %d", space);
    CK_TRY(ck_ensure(ctx, stage, bytes));
    CK_HIP(ctx, hipMemcpyAsync(stage.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dev = stage.p;
    return CK_OK;
}

int ck_from_device(ck_ctx* ctx, void* dst, const void* dev, size_t bytes, int space)
{
    if (!dst || dst == dev) return CK_OK;
    if (space == CK_DEVICE) {
        CK_HIP(ctx, hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        CK_HIP(ctx, hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    return CK_OK;
}

// ---- timing ---------------------------------------------------------------------------
TimeScope::TimeScope(ck_ctx* c, const char* name) : ctx(c)
{
    if (!ctx->timing) return;
    PendingEvent pe;
    auto get = [&](hipEvent_t* e) {
        if (!ctx->event_pool.empty()) { *e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
        else (void)hipEventCreate(e);
    };
    get(&pe.a); get(&pe.b);
    pe.name = name;
    (void)hipEventRecord(pe.a, ctx->stream);
    ctx->pending.push_back(pe);
    idx = (int)ctx->pending.size() - 1;
}
TimeScope::~TimeScope()
{
    if (idx >= 0) (void)hipEventRecord(ctx->pending[idx].b, ctx->stream);
}

int ck_timing_collect(ck_ctx* ctx)
{
    for (auto& pe : ctx->pending) {
        float ms = 0;
        if (hipEventSynchronize(pe.b) == hipSuccess && hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
            auto& s = ctx->slots[pe.name];
            s.ms += ms; s.launches += 1;
        }
        ctx->event_pool.push_back(pe.a);
        ctx->event_pool.push_back(pe.b);
    }
    ctx->pending.clear();
    return CK_OK;
}

// OpenCV's getThreshVal_Otsu_8u restated (double arithmetic, the FLT_EPSILON guards, first maximum wins)
double ck_otsu_level(const int* hist, size_t npx)
{
    double mu = 0, scale = 1. / (double)npx;
    for (int i = 0; i < 256; i++) mu += i * (double)hist[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < 256; i++) {
        const double p_i = hist[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        const double q2 = 1. - q1;
        if (std::min(q1, q2) < FLT_EPSILON || std::max(q1, q2) > 1. - FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
    }
    return max_val;
}

// the device half of ck_goban_canny: interleaved BGR on the device -> edge maps on the device (one host round trip
// for the Otsu levels of the batch).  Scratch: planes, out_stage, misc, map, labels, mats.
int ck_goban_canny_dev(ck_ctx* ctx, const uint8_t* d_in, int n, int h, int w, uint8_t* d_edges, double* otsu_out)
{
    const size_t npx1 = (size_t)h * w, npx = (size_t)n * npx1;
    const int pitch = ck_pitch(w);
    CK_TRY(ck_ensure(ctx, ctx->planes, (size_t)n * 3 * h * pitch));
    CK_TRY(ck_ensure(ctx, ctx->out_stage, npx * 3));
    // medianBlur 13, then 7 (the kernel reads interleaved BGR and writes planes)
    CK_TRY(k_median_planar(ctx, d_in, n, h, w, 13, (uint8_t*)ctx->planes.p, pitch));
    CK_TRY(k_planar_to_interleaved(ctx, (const uint8_t*)ctx->planes.p, n, h, w, pitch, (uint8_t*)ctx->out_stage.p));
    CK_TRY(ck_ensure(ctx, ctx->trange, ck_range_bytes(n, h, w)));
    CK_TRY(k_median_planar(ctx, (const uint8_t*)ctx->out_stage.p, n, h, w, 7, (uint8_t*)ctx->planes.p, pitch, (uint8_t*)ctx->trange.p));
    // grey histogram per frame -> Otsu level on the host (256 bins, double arithmetic as the library does it)
    CK_TRY(ck_ensure(ctx, ctx->misc, (size_t)n * 256 * 4 + (size_t)n * 64 + 4096));
    int* d_hist = (int*)ctx->misc.p;
    CK_TRY(k_gray_hist(ctx, (const uint8_t*)ctx->planes.p, n, h, w, pitch, d_hist));
    std::vector<int> hist((size_t)n * 256);
    CK_HIP(ctx, hipMemcpyAsync(hist.data(), d_hist, hist.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    CK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CK_TRY(ck_ensure(ctx, ctx->map, npx));
    CK_TRY(ck_ensure(ctx, ctx->labels, npx * 4));
    // cv2.Canny(median, otsu / 2, otsu): thresholds are floored (L1 gradient); one pair per frame, one batched call
    std::vector<int> thr((size_t)n * 2);
    for (int f = 0; f < n; f++) {
        const double otsu = ck_otsu_level(&hist[(size_t)f * 256], npx1);
        if (otsu_out) otsu_out[f] = otsu;
        thr[2 * f] = (int)std::floor(otsu / 2);
        thr[2 * f + 1] = (int)std::floor(otsu);
    }
    CK_TRY(ck_ensure(ctx, ctx->mats, (size_t)n * 8 + 1024));
    int* d_thr = (int*)ctx->mats.p;
    CK_HIP(ctx, hipMemcpyAsync(d_thr, thr.data(), thr.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    CK_HIP(ctx, hipStreamSynchronize(ctx->stream));           // thr is a local: the copy must be done before it goes
    return k_canny_planar(ctx, (const uint8_t*)ctx->planes.p, n, h, w, pitch, 0, 0, (uint8_t*)ctx->map.p,
                          (int32_t*)ctx->labels.p, d_edges, nullptr, nullptr, d_thr, (const uint8_t*)ctx->trange.p);
}

extern "C" {

int ck_version(void) { return 100; }

int ck_ctx_create(int device, ck_ctx** out) { return ck_ctx_create_prio(device, 0, out); }

int ck_ctx_create_prio(int device, int priority, ck_ctx** out)
{
    if (!out) return ck_fail(nullptr, CK_ERR_ARG, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return ck_fail(nullptr, CK_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count) return ck_fail(nullptr, CK_ERR_ARG, "device %d out of range [0,%d)", device, count);
    e = hipSetDevice(device);
    if (e != hipSuccess) return ck_fail(nullptr, CK_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    ck_ctx* ctx = new ck_ctx();
    ctx->device = device;
    if (priority == 0)
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    else {
        int least = 0, greatest = 0;                     // (numerically: greatest priority = the smaller number)
        e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, priority > 0 ? greatest : least);
    }
    if (e != hipSuccess) { delete ctx; return ck_fail(nullptr, CK_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    *out = ctx;
    return CK_OK;
}

void ck_ctx_destroy(ck_ctx* ctx)
{
    if (ck_ctx_destroy2(ctx) == CK_ERR_STATE)
        fprintf(stderr, "ck_ctx_destroy: context %p is still in use by another thread after 5 s; not freed\n", (void*)ctx);
}

int ck_stream_wait(ck_ctx* ctx, void* stream)
{
    CK_API_BEGIN(ctx)
    if (!ctx->handover) CK_HIP(ctx, hipEventCreateWithFlags(&ctx->handover, hipEventDisableTiming));
    CK_HIP(ctx, hipEventRecord(ctx->handover, (hipStream_t)stream));
    CK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->handover, 0));
    return CK_OK;
    CK_API_END(ctx)
}

int ck_ctx_destroy2(ck_ctx* ctx)
{
    if (!ctx) return CK_OK;
    // A thread still inside an entry point owns the stream and the scratch buffers: freeing them under it would be a
    // use-after-free.  Wait for it to leave (a call is milliseconds); a context that stays busy is leaked, and said so.
    unsigned long long nobody = 0;
    const unsigned long long me = ck_thread_token();
    for (int spin = 0; !ctx->owner.compare_exchange_strong(nobody, me, std::memory_order_acq_rel); spin++) {
        if (nobody == me) break;                     // destroyed from inside one of its own calls: nothing to wait for
        if (spin >= 5000) return CK_ERR_STATE;          // still busy: not freed, the handle stays valid
        nobody = 0;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf* bufs[] = { &ctx->in_stage, &ctx->in_stage2, &ctx->planes, &ctx->trange, &ctx->edges, &ctx->map, &ctx->labels,
                       &ctx->labels2, &ctx->runs, &ctx->ghost, &ctx->misc, &ctx->bflag, &ctx->comp, &ctx->lists, &ctx->pts, &ctx->accum, &ctx->peaks,
                       &ctx->goban, &ctx->act0, &ctx->act1, &ctx->act2, &ctx->ybuf, &ctx->lblbuf, &ctx->confbuf,
                       &ctx->rlblbuf, &ctx->rconfbuf, &ctx->fgcbuf,
                       &ctx->out_stage, &ctx->mats, &ctx->rec_stage,
                       &ctx->cnn.c1w, &ctx->cnn.c1b, &ctx->cnn.c2w, &ctx->cnn.c2b, &ctx->cnn.c3w, &ctx->cnn.c3b,
                       &ctx->cnn.c4w, &ctx->cnn.c4b, &ctx->cnn.d1w, &ctx->cnn.d1b, &ctx->cnn.d2w, &ctx->cnn.d2b,
                       &ctx->cnn.c2w_bf, &ctx->cnn.c3w_bf, &ctx->cnn.c4w_bf, &ctx->cnn.d1w_bf, &ctx->cnn.c1w_f16, &ctx->cnn.d1w_bfp, &ctx->cnn.c1w_q8, &ctx->cnn.c2x_q8, &ctx->cnn.c3x_q8, &ctx->cnn.c4x_q8, &ctx->cnn.d1w_h2,
                       &ctx->cnn.c1w_h2, &ctx->cnn.c2w_h2, &ctx->cnn.c3w_h2, &ctx->cnn.c4w_h2 };
    for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
    if (ctx->cnn_flag_host) (void)hipHostFree(ctx->cnn_flag_host);
    for (auto& m : ctx->mog2) {
        DevBuf* mb[] = { &m.weight, &m.variance, &m.mean, &m.nmodes };
        for (DevBuf* b : mb) if (b->p) (void)hipFree(b->p);
        if (m.rates_host) (void)hipHostFree(m.rates_host);
        if (m.rates_dev) (void)hipFree(m.rates_dev);
        if (m.rates_done) (void)hipEventDestroy(m.rates_done);
    }
    for (auto& pe : ctx->pending) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->handover) (void)hipEventDestroy(ctx->handover);
    if (ctx->host_pinned) (void)hipHostFree(ctx->host_pinned);
    if (ctx->host_pinned2) (void)hipHostFree(ctx->host_pinned2);
    if (ctx->rec_host) (void)hipHostFree(ctx->rec_host);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CK_OK;
}

const char* ck_last_error(const ck_ctx* ctx)
{
    if (!ctx) return g_ck_create_error.c_str();
    if (g_ck_busy_ctx == ctx) {              // this thread's call was refused: ctx->err belongs to the thread inside
        g_ck_busy_ctx = nullptr;
        return "the context is in use by another thread (a ck_ctx is single-threaded: one context per finder / thread)";
    }
    return ctx->err.c_str();
}
int ck_backend(const ck_ctx*) { return CK_BACKEND_HIP; }
void* ck_stream(ck_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int ck_timing_enable(ck_ctx* ctx, int on)
{
    CK_API_BEGIN(ctx)
    ctx->timing = on != 0;
    return CK_OK;
    CK_API_END(ctx)
}
int ck_timing_reset(ck_ctx* ctx)
{
    CK_API_BEGIN(ctx)
    if (!ctx) return CK_ERR_ARG;
    ck_timing_collect(ctx);
    ctx->slots.clear();
    return CK_OK;
    CK_API_END(ctx)
}
int ck_timing_get(ck_ctx* ctx, const char* name, double* total_ms, int* launches)
{
    CK_API_BEGIN(ctx)
    if (!ctx || !name) return CK_ERR_ARG;
    ck_timing_collect(ctx);
    auto it = ctx->slots.find(name);
    if (total_ms) *total_ms = it == ctx->slots.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == ctx->slots.end() ? 0 : it->second.launches;
    return CK_OK;
    CK_API_END(ctx)
}

static int check_img(ck_ctx* ctx, const void* p, int n, int h, int w)
{
    if (!ctx) return CK_ERR_ARG;
    if (!p) return ck_fail(ctx, CK_ERR_ARG, "image pointer is NULL");
    if (n <= 0 || h <= 0 || w <= 0) return ck_fail(ctx, CK_ERR_ARG, "bad shape n=%d h=%d w=%d", n, h, w);
    if ((long long)h * w > (1LL << 28)) return ck_fail(ctx, CK_ERR_ARG, "image too large");
    CK_HIP(ctx, hipSetDevice(ctx->device));
    return CK_OK;
}

static int finish(ck_ctx* ctx)
{
    CK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}

int ck_median15(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space, uint8_t* out, int out_space)
{
    return ck_median(ctx, bgr, n, h, w, 15, in_space, out, out_space);
}

int ck_median(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int ksize, int in_space, uint8_t* out, int out_space)
{
    CK_API_BEGIN(ctx)
    CK_TRY(check_img(ctx, bgr, n, h, w));
    if (!out) return ck_fail(ctx, CK_ERR_ARG, "out is NULL");
    if (ksize < 3 || ksize > 17 || !(ksize & 1)) return ck_fail(ctx, CK_ERR_ARG, "median window %d: odd sizes 3..17 only", ksize);
    const size_t bytes = (size_t)n * h * w * 3;
    const int pitch = ck_pitch(w);
    const void* d_in;
    CK_TRY(ck_to_device(ctx, bgr, bytes, in_space, ctx->in_stage, &d_in));
    CK_TRY(ck_ensure(ctx, ctx->planes, (size_t)n * 3 * h * pitch));
    CK_TRY(k_median_planar(ctx, (const uint8_t*)d_in, n, h, w, ksize, (uint8_t*)ctx->planes.p, pitch));
    uint8_t* d_out = out;
    if (out_space == CK_HOST) { CK_TRY(ck_ensure(ctx, ctx->out_stage, bytes)); d_out = (uint8_t*)ctx->out_stage.p; }
    CK_TRY(k_planar_to_interleaved(ctx, (const uint8_t*)ctx->planes.p, n, h, w, pitch, d_out));
    if (out_space == CK_HOST) CK_TRY(ck_from_device(ctx, out, d_out, bytes, CK_HOST));
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_canny(ck_ctx* ctx, const uint8_t* img3, int n, int h, int w, int in_space,
             int low, int high, uint8_t* edges, uint8_t* map_out, int out_space)
{
    CK_API_BEGIN(ctx)
    CK_TRY(check_img(ctx, img3, n, h, w));
    if (!edges) return ck_fail(ctx, CK_ERR_ARG, "edges is NULL");
    const size_t npx = (size_t)n * h * w;
    const int pitch = ck_pitch(w);
    const void* d_in;
    CK_TRY(ck_to_device(ctx, img3, npx * 3, in_space, ctx->in_stage, &d_in));
    CK_TRY(ck_ensure(ctx, ctx->planes, (size_t)n * 3 * h * pitch));
    CK_TRY(k_interleaved_to_planar(ctx, (const uint8_t*)d_in, n, h, w, pitch, (uint8_t*)ctx->planes.p));
    CK_TRY(ck_ensure(ctx, ctx->map, npx));
    CK_TRY(ck_ensure(ctx, ctx->labels, npx * 4));
    uint8_t* d_edges = edges;
    uint8_t* d_mapout = map_out;
    if (out_space == CK_HOST) {
        CK_TRY(ck_ensure(ctx, ctx->edges, npx)); d_edges = (uint8_t*)ctx->edges.p;
        if (map_out) { CK_TRY(ck_ensure(ctx, ctx->out_stage, npx)); d_mapout = (uint8_t*)ctx->out_stage.p; }
    }
    CK_TRY(k_canny_planar(ctx, (const uint8_t*)ctx->planes.p, n, h, w, pitch, low, high,
                          (uint8_t*)ctx->map.p, (int32_t*)ctx->labels.p, d_edges, d_mapout));
    if (out_space == CK_HOST) {
        CK_TRY(ck_from_device(ctx, edges, d_edges, npx, CK_HOST));
        if (map_out) CK_TRY(ck_from_device(ctx, map_out, d_mapout, npx, CK_HOST));
    }
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_goban_canny(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space, uint8_t* edges, int out_space,
                   double* otsu_out)
{
    CK_API_BEGIN(ctx)
    CK_TRY(check_img(ctx, bgr, n, h, w));
    if (!edges) return ck_fail(ctx, CK_ERR_ARG, "edges is NULL");
    const size_t npx = (size_t)n * h * w;
    const void* d_in;
    CK_TRY(ck_to_device(ctx, bgr, npx * 3, in_space, ctx->in_stage, &d_in));
    uint8_t* d_edges = edges;
    if (out_space == CK_HOST) { CK_TRY(ck_ensure(ctx, ctx->edges, npx)); d_edges = (uint8_t*)ctx->edges.p; }
    CK_TRY(ck_goban_canny_dev(ctx, (const uint8_t*)d_in, n, h, w, d_edges, otsu_out));
    if (out_space == CK_HOST) CK_TRY(ck_from_device(ctx, edges, d_edges, npx, CK_HOST));
    return finish(ctx);
    CK_API_END(ctx)
}

static int board_edges_dev(ck_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w, uint8_t* d_edges,
                           int* d_border_flag = nullptr)
{
    const size_t npx = (size_t)n * h * w;
    const int pitch = ck_pitch(w);
    CK_TRY(ck_ensure(ctx, ctx->planes, (size_t)n * 3 * h * pitch));
    CK_TRY(ck_ensure(ctx, ctx->map, npx));
    CK_TRY(ck_ensure(ctx, ctx->labels, npx * 4));
    CK_TRY(ck_ensure(ctx, ctx->trange, ck_range_bytes(n, h, w)));
    CK_TRY(k_median15_planar(ctx, d_bgr, n, h, w, (uint8_t*)ctx->planes.p, pitch, (uint8_t*)ctx->trange.p));
    CK_TRY(k_canny_planar(ctx, (const uint8_t*)ctx->planes.p, n, h, w, pitch, 25, 75,
                          (uint8_t*)ctx->map.p, (int32_t*)ctx->labels.p, d_edges, nullptr, d_border_flag, nullptr,
                          (const uint8_t*)ctx->trange.p));
    return CK_OK;
}

int ck_board_edges(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space, uint8_t* edges, int out_space)
{
    CK_API_BEGIN(ctx)
    CK_TRY(check_img(ctx, bgr, n, h, w));
    if (!edges) return ck_fail(ctx, CK_ERR_ARG, "edges is NULL");
    const size_t npx = (size_t)n * h * w;
    const void* d_in;
    CK_TRY(ck_to_device(ctx, bgr, npx * 3, in_space, ctx->in_stage, &d_in));
    uint8_t* d_edges = edges;
    if (out_space == CK_HOST) { CK_TRY(ck_ensure(ctx, ctx->edges, npx)); d_edges = (uint8_t*)ctx->edges.p; }
    CK_TRY(board_edges_dev(ctx, (const uint8_t*)d_in, n, h, w, d_edges));
    if (out_space == CK_HOST) CK_TRY(ck_from_device(ctx, edges, d_edges, npx, CK_HOST));
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_board_lines(ck_ctx* ctx, const uint8_t* edges, int n, int h, int w, int in_space,
                   int hough_thresh, float* lines, int cap, ck_board_result* res,
                   uint8_t* ghost_out, int ghost_space)
{
    CK_API_BEGIN(ctx)
    CK_TRY(check_img(ctx, edges, n, h, w));
    if (!lines || !res || cap <= 0) return ck_fail(ctx, CK_ERR_ARG, "lines/res NULL or cap <= 0");
    if (h < 3 || w < 3) return ck_fail(ctx, CK_ERR_ARG, "image smaller than 3x3");
    const size_t npx = (size_t)n * h * w;
    const void* d_in;
    CK_TRY(ck_to_device(ctx, edges, npx, in_space, ctx->in_stage2, &d_in));
    if (hough_thresh < 0) hough_thresh = (int)((h < w ? h : w) / 5.0);
    uint8_t* d_ghost = nullptr;
    if (ghost_out) {
        if (ghost_space == CK_DEVICE) d_ghost = ghost_out;
        else { CK_TRY(ck_ensure(ctx, ctx->out_stage, npx)); d_ghost = (uint8_t*)ctx->out_stage.p; }
    }
    CK_TRY(k_board_lines(ctx, (const uint8_t*)d_in, n, h, w, hough_thresh, lines, cap, res, d_ghost));
    if (ghost_out && ghost_space == CK_HOST) CK_TRY(ck_from_device(ctx, ghost_out, d_ghost, npx, CK_HOST));
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_board_detect(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space,
                    int hough_thresh, float* lines, int cap, ck_board_result* res)
{
    CK_API_BEGIN(ctx)
    CK_TRY(check_img(ctx, bgr, n, h, w));
    if (!lines || !res || cap <= 0) return ck_fail(ctx, CK_ERR_ARG, "lines/res NULL or cap <= 0");
    if (h < 3 || w < 3) return ck_fail(ctx, CK_ERR_ARG, "image smaller than 3x3");
    const size_t npx = (size_t)n * h * w;
    const void* d_in;
    CK_TRY(ck_to_device(ctx, bgr, npx * 3, in_space, ctx->in_stage, &d_in));
    CK_TRY(ck_ensure(ctx, ctx->edges, npx));
    CK_TRY(ck_ensure(ctx, ctx->bflag, (size_t)n * 4));
    // K2 leaves its hysteresis labels in ctx->labels, which is also K3's parent image: K3 keeps the edge
    // components Canny already built (frames with an edge on the image frame are relabelled from scratch)
    CK_TRY(board_edges_dev(ctx, (const uint8_t*)d_in, n, h, w, (uint8_t*)ctx->edges.p, (int*)ctx->bflag.p));
    if (hough_thresh < 0) hough_thresh = (int)((h < w ? h : w) / 5.0);
    CK_TRY(k_board_lines(ctx, (const uint8_t*)ctx->edges.p, n, h, w, hough_thresh, lines, cap, res, nullptr,
                         (const int*)ctx->bflag.p));
    return finish(ctx);
    CK_API_END(ctx)
}

static int upload_minv(ck_ctx* ctx, const double* M, int m_count, int n, const double** d_minv)
{
    if (!M) return ck_fail(ctx, CK_ERR_ARG, "M is NULL");
    if (m_count != 1 && m_count != n) return ck_fail(ctx, CK_ERR_ARG, "m_count must be 1 or n");
    std::vector<double> inv((size_t)m_count * 9);
    for (int i = 0; i < m_count; i++) ck_invert3x3(M + 9 * i, inv.data() + 9 * i);
    CK_TRY(ck_ensure(ctx, ctx->mats, inv.size() * sizeof(double)));
    // pageable source: the copy is staged by the runtime before the call returns
    CK_HIP(ctx, hipMemcpyAsync(ctx->mats.p, inv.data(), inv.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    CK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *d_minv = (const double*)ctx->mats.p;
    return CK_OK;
}

int ck_i420_to_bgr(ck_ctx* ctx, const uint8_t* i420, int n, int h, int w, int in_space, uint8_t* bgr, int out_space)
{
    CK_API_BEGIN(ctx)
    CK_TRY(check_img(ctx, i420, n, h, w));
    if (!bgr) return ck_fail(ctx, CK_ERR_ARG, "bgr is NULL");
    if ((h & 1) || (w & 1)) return ck_fail(ctx, CK_ERR_ARG, "I420 needs even dimensions, got %dx%d", w, h);
    const void* d_in;
    CK_TRY(ck_to_device(ctx, i420, (size_t)n * h * w * 3 / 2, in_space, ctx->in_stage2, &d_in));
    const size_t obytes = (size_t)n * h * w * 3;
    uint8_t* d_out = bgr;
    if (out_space == CK_HOST) { CK_TRY(ck_ensure(ctx, ctx->out_stage, obytes)); d_out = (uint8_t*)ctx->out_stage.p; }
    CK_TRY(k_i420_to_bgr(ctx, (const uint8_t*)d_in, n, h, w, d_out));
    if (out_space == CK_HOST) CK_TRY(ck_from_device(ctx, bgr, d_out, obytes, CK_HOST));
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_warp_perspective(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space,
                        const double* M, int m_count, int dsize, uint8_t* out, int out_space)
{
    CK_API_BEGIN(ctx)
    CK_TRY(check_img(ctx, bgr, n, h, w));
    if (!out || dsize <= 0) return ck_fail(ctx, CK_ERR_ARG, "out NULL or dsize <= 0");
    const void* d_in;
    CK_TRY(ck_to_device(ctx, bgr, (size_t)n * h * w * 3, in_space, ctx->in_stage, &d_in));
    const double* d_minv;
    CK_TRY(upload_minv(ctx, M, m_count, n, &d_minv));
    const size_t obytes = (size_t)n * dsize * dsize * 3;
    uint8_t* d_out = out;
    if (out_space == CK_HOST) { CK_TRY(ck_ensure(ctx, ctx->goban, obytes)); d_out = (uint8_t*)ctx->goban.p; }
    CK_TRY(k_warp(ctx, (const uint8_t*)d_in, n, h, w, d_minv, m_count, dsize, d_out));
    if (out_space == CK_HOST) CK_TRY(ck_from_device(ctx, out, d_out, obytes, CK_HOST));
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_cnn_set_weights(ck_ctx* ctx, const float* const weights[12], int space)
{
    CK_API_BEGIN(ctx)
    if (!ctx || !weights) return CK_ERR_ARG;
    for (int i = 0; i < 12; i++) if (!weights[i]) return ck_fail(ctx, CK_ERR_ARG, "weights[%d] is NULL", i);
    CK_HIP(ctx, hipSetDevice(ctx->device));
    CK_TRY(k_cnn_pack_weights(ctx, weights, space));
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_cnn_set_mode(ck_ctx* ctx, int mode)
{
    CK_API_BEGIN(ctx)
    if (!ctx) return CK_ERR_ARG;
    if (mode != CK_CNN_FP32 && mode != CK_CNN_BF16 && mode != CK_CNN_F16X2 && mode != CK_CNN_F16Q8) return ck_fail(ctx, CK_ERR_ARG, "unknown cnn mode %d", mode);
    ctx->cnn_mode = mode;
    return CK_OK;
    CK_API_END(ctx)
}

// where the region outputs of a call go (all optional)
struct RegionOut { uint8_t* label = nullptr; double* conf = nullptr; ck_frame_record* rec = nullptr; int rec_space = CK_HOST; };

static int ensure_rec_host(ck_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->rec_host_cap) return CK_OK;
    if (ctx->rec_host) { CK_HIP(ctx, hipHostFree(ctx->rec_host)); ctx->rec_host = nullptr; ctx->rec_host_cap = 0; }
    bytes += bytes / 4 + 4096;
    CK_HIP(ctx, hipHostMalloc(&ctx->rec_host, bytes, hipHostMallocDefault));
    ctx->rec_host_cap = bytes;
    return CK_OK;
}

static int cnn_predict_dev(ck_ctx* ctx, const uint8_t* d_goban, int n, float* y, uint8_t* labels, double* conf, int out_space,
                           RegionOut ro = RegionOut())
{
    if (!ctx->cnn.set) return ck_fail(ctx, CK_ERR_STATE, "ck_cnn_set_weights has not been called");
    CK_TRY(ck_ensure(ctx, ctx->ybuf, (size_t)n * 8100 * sizeof(float)));
    CK_TRY(ck_ensure(ctx, ctx->lblbuf, (size_t)n * 361));
    CK_TRY(ck_ensure(ctx, ctx->confbuf, (size_t)n * 361 * sizeof(double)));
    CK_TRY(ck_ensure(ctx, ctx->rlblbuf, (size_t)n * 100));
    CK_TRY(ck_ensure(ctx, ctx->rconfbuf, (size_t)n * 100 * sizeof(double)));
    if (ck_cnn_split(ctx->cnn_mode)) {
        // Safety net of the split-precision mode: an activation beyond the fp16 range (|x| > 65000; never seen with
        // 8-bit images and sane weights) would turn into inf.  The kernels raise a flag in host-mapped memory, which
        // cnn_finish() looks at after the one synchronisation the call needs anyway.
        if (!ctx->cnn_flag_host) {
            CK_HIP(ctx, hipHostMalloc((void**)&ctx->cnn_flag_host, 64, hipHostMallocMapped));
            CK_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->cnn_flag_dev, ctx->cnn_flag_host, 0));
        }
        *ctx->cnn_flag_host = 0;
        CK_TRY(k_cnn_predict(ctx, d_goban, n, (float*)ctx->ybuf.p, (uint8_t*)ctx->lblbuf.p, (double*)ctx->confbuf.p, ctx->cnn_flag_dev,
                             (uint8_t*)ctx->rlblbuf.p, (double*)ctx->rconfbuf.p));
    } else {
        CK_TRY(k_cnn_predict(ctx, d_goban, n, (float*)ctx->ybuf.p, (uint8_t*)ctx->lblbuf.p, (double*)ctx->confbuf.p, nullptr,
                             (uint8_t*)ctx->rlblbuf.p, (double*)ctx->rconfbuf.p));
    }
    if (ro.rec && ro.rec_space == CK_DEVICE)
        CK_TRY(k_records_put_regions(ctx, (const uint8_t*)ctx->rlblbuf.p, (const double*)ctx->rconfbuf.p, n, ro.rec));
    else if (ro.rec) {           // records in host memory: both region arrays to pinned staging, scattered after the call's sync
        CK_TRY(ensure_rec_host(ctx, (size_t)n * 900));
        CK_TRY(ck_from_device(ctx, ctx->rec_host, ctx->rconfbuf.p, (size_t)n * 800, CK_HOST));
        CK_TRY(ck_from_device(ctx, (uint8_t*)ctx->rec_host + (size_t)n * 800, ctx->rlblbuf.p, (size_t)n * 100, CK_HOST));
    }
    if (ro.label) CK_TRY(ck_from_device(ctx, ro.label, ctx->rlblbuf.p, (size_t)n * 100, out_space));
    if (ro.conf) CK_TRY(ck_from_device(ctx, ro.conf, ctx->rconfbuf.p, (size_t)n * 100 * sizeof(double), out_space));
    if (y) CK_TRY(ck_from_device(ctx, y, ctx->ybuf.p, (size_t)n * 8100 * sizeof(float), out_space));
    if (labels) CK_TRY(ck_from_device(ctx, labels, ctx->lblbuf.p, (size_t)n * 361, out_space));
    if (conf) CK_TRY(ck_from_device(ctx, conf, ctx->confbuf.p, (size_t)n * 361 * sizeof(double), out_space));
    return CK_OK;
}

// synchronise; if the split-precision kernels flagged a value outside the fp16 range, recompute the batch with
// the f32 kernels (the goban images are still in place) and deliver again
static int cnn_finish(ck_ctx* ctx, const uint8_t* d_goban, int n, float* y, uint8_t* labels, double* conf, int out_space,
                      RegionOut ro = RegionOut())
{
    CK_TRY(finish(ctx));
    if (ck_cnn_split(ctx->cnn_mode) && ctx->cnn_flag_host && *(volatile int*)ctx->cnn_flag_host) {
        const int mode = ctx->cnn_mode;
        ctx->cnn_mode = CK_CNN_FP32;
        const int rc = cnn_predict_dev(ctx, d_goban, n, y, labels, conf, out_space, ro);
        ctx->cnn_mode = mode;
        ctx->cnn_fallbacks++;
        if (rc) return rc;
        return finish(ctx);
    }
    return CK_OK;
}

int ck_cnn_predict(ck_ctx* ctx, const uint8_t* goban, int n, int in_space,
                   float* y, uint8_t* labels, double* conf, int out_space)
{
    CK_API_BEGIN(ctx)
    if (!ctx) return CK_ERR_ARG;
    if (!goban || n <= 0) return ck_fail(ctx, CK_ERR_ARG, "goban NULL or n <= 0");
    CK_HIP(ctx, hipSetDevice(ctx->device));
    const void* d_in;
    CK_TRY(ck_to_device(ctx, goban, (size_t)n * 380 * 380 * 3, in_space, ctx->in_stage, &d_in));
    CK_TRY(cnn_predict_dev(ctx, (const uint8_t*)d_in, n, y, labels, conf, out_space));
    return cnn_finish(ctx, (const uint8_t*)d_in, n, y, labels, conf, out_space);
    CK_API_END(ctx)
}

int ck_cnn_maps(ck_ctx* ctx, const uint8_t* goban, int n, int in_space, float* pool2, float* pool4)
{
    CK_API_BEGIN(ctx)
    if (!ctx) return CK_ERR_ARG;
    if (!goban || n <= 0 || n > 128) return ck_fail(ctx, CK_ERR_ARG, "goban NULL or n outside 1 .. 128");
    const void* d_in;
    CK_TRY(ck_to_device(ctx, goban, (size_t)n * 380 * 380 * 3, in_space, ctx->in_stage, &d_in));
    CK_TRY(cnn_predict_dev(ctx, (const uint8_t*)d_in, n, nullptr, nullptr, nullptr, CK_HOST));
    CK_TRY(finish(ctx));
    if (ck_cnn_split(ctx->cnn_mode) && ctx->cnn_flag_host && *(volatile int*)ctx->cnn_flag_host)
        return ck_fail(ctx, CK_ERR_STATE, "an activation left the fp16 range: the maps of this batch are the f32 chain's (set CK_CNN_FP32)");
    // after one chunk the pooled conv2 output is still in act1 and the pooled conv4 output in act2 (k_cnn_predict)
    const size_t np = (size_t)n * 100;
    if (ctx->cnn_mode == CK_CNN_BF16) {
        // bf16 activations, conv4's channels padded to 96: widened on the host
        std::vector<uint16_t> raw;
        auto widen = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
        if (pool2) {
            raw.resize(np * 8192);
            CK_HIP(ctx, hipMemcpy(raw.data(), ctx->act1.p, raw.size() * 2, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < raw.size(); i++) pool2[i] = widen(raw[i]);
        }
        if (pool4) {
            raw.resize(np * 3456);
            CK_HIP(ctx, hipMemcpy(raw.data(), ctx->act2.p, raw.size() * 2, hipMemcpyDeviceToHost));
            for (size_t p = 0; p < np; p++)
                for (int px = 0; px < 36; px++)
                    for (int c = 0; c < 90; c++) pool4[(p * 36 + px) * 90 + c] = widen(raw[(p * 36 + px) * 96 + c]);
        }
        return CK_OK;
    }
    if (pool2) CK_HIP(ctx, hipMemcpy(pool2, ctx->act1.p, np * 8192 * sizeof(float), hipMemcpyDeviceToHost));
    if (pool4) CK_HIP(ctx, hipMemcpy(pool4, ctx->act2.p, np * 3240 * sizeof(float), hipMemcpyDeviceToHost));
    return CK_OK;
    CK_API_END(ctx)
}

int ck_stones_detect(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space,
                     const double* M, int m_count, uint8_t* labels, double* conf, int out_space)
{
    CK_API_BEGIN(ctx)
    CK_TRY(check_img(ctx, bgr, n, h, w));
    const void* d_in;
    CK_TRY(ck_to_device(ctx, bgr, (size_t)n * h * w * 3, in_space, ctx->in_stage, &d_in));
    const double* d_minv;
    CK_TRY(upload_minv(ctx, M, m_count, n, &d_minv));
    CK_TRY(ck_ensure(ctx, ctx->goban, (size_t)n * 380 * 380 * 3));
    CK_TRY(k_warp(ctx, (const uint8_t*)d_in, n, h, w, d_minv, m_count, 380, (uint8_t*)ctx->goban.p));
    CK_TRY(cnn_predict_dev(ctx, (const uint8_t*)ctx->goban.p, n, nullptr, labels, conf, out_space));
    return cnn_finish(ctx, (const uint8_t*)ctx->goban.p, n, nullptr, labels, conf, out_space);
    CK_API_END(ctx)
}

int ck_cnn_regions(ck_ctx* ctx, const uint8_t* goban, int n, int in_space, uint8_t* region_label, double* region_conf,
                   int out_space)
{
    CK_API_BEGIN(ctx)
    if (!ctx) return CK_ERR_ARG;
    if (!goban || n <= 0 || !region_label || !region_conf) return ck_fail(ctx, CK_ERR_ARG, "NULL argument or n <= 0");
    CK_HIP(ctx, hipSetDevice(ctx->device));
    const void* d_in;
    CK_TRY(ck_to_device(ctx, goban, (size_t)n * 380 * 380 * 3, in_space, ctx->in_stage, &d_in));
    RegionOut ro; ro.label = region_label; ro.conf = region_conf;
    CK_TRY(cnn_predict_dev(ctx, (const uint8_t*)d_in, n, nullptr, nullptr, nullptr, out_space, ro));
    return cnn_finish(ctx, (const uint8_t*)d_in, n, nullptr, nullptr, nullptr, out_space, ro);
    CK_API_END(ctx)
}

int ck_cnn_regions_records(ck_ctx* ctx, const uint8_t* goban, int n, int in_space, ck_frame_record* rec, int rec_space)
{
    CK_API_BEGIN(ctx)
    if (!ctx) return CK_ERR_ARG;
    if (!goban || n <= 0 || !rec) return ck_fail(ctx, CK_ERR_ARG, "NULL argument or n <= 0");
    if (rec_space != CK_HOST && rec_space != CK_DEVICE) return ck_fail(ctx, CK_ERR_ARG, "bad memory space %d", rec_space);
    CK_HIP(ctx, hipSetDevice(ctx->device));
    const void* d_in;
    CK_TRY(ck_to_device(ctx, goban, (size_t)n * 380 * 380 * 3, in_space, ctx->in_stage, &d_in));
    RegionOut ro; ro.rec = rec; ro.rec_space = rec_space;
    CK_TRY(cnn_predict_dev(ctx, (const uint8_t*)d_in, n, nullptr, nullptr, nullptr, CK_HOST, ro));
    CK_TRY(cnn_finish(ctx, (const uint8_t*)d_in, n, nullptr, nullptr, nullptr, CK_HOST, ro));
    if (rec_space == CK_HOST) {
        const double* conf = (const double*)ctx->rec_host;
        const uint8_t* lab = (const uint8_t*)ctx->rec_host + (size_t)n * 800;
        for (int f = 0; f < n; f++) {
            memcpy(rec[f].region_conf, conf + (size_t)f * 100, 800);
            memcpy(rec[f].region_label, lab + (size_t)f * 100, 100);
        }
    }
    return CK_OK;
    CK_API_END(ctx)
}

int ck_board_detect_records(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space,
                            int hough_thresh, ck_frame_record* rec, int rec_space)
{
    CK_API_BEGIN(ctx)
    CK_TRY(check_img(ctx, bgr, n, h, w));
    if (!rec) return ck_fail(ctx, CK_ERR_ARG, "rec is NULL");
    if (rec_space != CK_HOST && rec_space != CK_DEVICE) return ck_fail(ctx, CK_ERR_ARG, "bad memory space %d", rec_space);
    // the board results are born on the host (K4's boxes, K6's peak order): n packed board halves in pinned memory,
    // then into the records -- a memcpy per record on the host, one upload + one small kernel for records in HBM
    const size_t part = offsetof(ck_frame_record, region_conf);
    const size_t lines_bytes = (size_t)n * CK_REC_LMAX * 2 * sizeof(float), res_bytes = (size_t)n * sizeof(ck_board_result);
    CK_TRY(ensure_rec_host(ctx, (size_t)n * part + lines_bytes + res_bytes));
    uint8_t* parts = (uint8_t*)ctx->rec_host;
    float* lines = (float*)(parts + (size_t)n * part);
    ck_board_result* res = (ck_board_result*)((uint8_t*)lines + lines_bytes);
    CK_TRY(ck_board_detect(ctx, bgr, n, h, w, in_space, hough_thresh, lines, CK_REC_LMAX, res));
    for (int f = 0; f < n; f++) {
        ck_frame_record* r = (ck_frame_record*)(parts + (size_t)f * part);       // (only the head of it exists here)
        const int kept = res[f].n_lines < CK_REC_LMAX ? (res[f].n_lines < 0 ? 0 : res[f].n_lines) : CK_REC_LMAX;
        r->status = res[f].status; r->n_contours = res[f].n_contours; r->n_lines = res[f].n_lines;
        r->flags = res[f].n_lines > CK_REC_LMAX ? CK_REC_LINES_CUT : 0;
        r->biggest_area = res[f].biggest_area;
        memcpy(r->lines, lines + (size_t)f * CK_REC_LMAX * 2, (size_t)kept * 2 * sizeof(float));
        memset(&r->lines[kept][0], 0, (size_t)(CK_REC_LMAX - kept) * 2 * sizeof(float));
    }
    if (rec_space == CK_HOST) {
        for (int f = 0; f < n; f++) memcpy(&rec[f], parts + (size_t)f * part, part);
        return CK_OK;
    }
    CK_TRY(ck_ensure(ctx, ctx->rec_stage, (size_t)n * part));
    CK_HIP(ctx, hipMemcpyAsync(ctx->rec_stage.p, parts, (size_t)n * part, hipMemcpyHostToDevice, ctx->stream));
    CK_TRY(k_records_put_board(ctx, (const uint8_t*)ctx->rec_stage.p, n, rec));
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_stones_run(ck_ctx* ctx, const uint8_t* bgr, int n, int h, int w, int in_space, const double* M, int m_count,
                  int mog2_handle, const double* learning_rates, uint8_t* region_label, double* region_conf,
                  int32_t* fgcount, uint8_t* labels, double* conf, int out_space)
{
    CK_API_BEGIN(ctx)
    CK_TRY(check_img(ctx, bgr, n, h, w));
    if (!region_label || !region_conf) return ck_fail(ctx, CK_ERR_ARG, "region outputs are NULL");
    const bool bg = mog2_handle >= 0;
    if (bg && (mog2_handle >= (int)ctx->mog2.size() || !ctx->mog2[mog2_handle].alive))
        return ck_fail(ctx, CK_ERR_ARG, "bad mog2 handle %d", mog2_handle);
    if (bg && (!learning_rates || !fgcount)) return ck_fail(ctx, CK_ERR_ARG, "a background model needs learning_rates and fgcount");
    if (bg && (ctx->mog2[mog2_handle].h != 380 || ctx->mog2[mog2_handle].w != 380))
        return ck_fail(ctx, CK_ERR_ARG, "the background model of a stones run is 380x380");
    const void* d_in;
    CK_TRY(ck_to_device(ctx, bgr, (size_t)n * h * w * 3, in_space, ctx->in_stage, &d_in));
    const double* d_minv;
    CK_TRY(upload_minv(ctx, M, m_count, n, &d_minv));
    CK_TRY(ck_ensure(ctx, ctx->goban, (size_t)n * 380 * 380 * 3));
    CK_TRY(k_warp(ctx, (const uint8_t*)d_in, n, h, w, d_minv, m_count, 380, (uint8_t*)ctx->goban.p));
    if (bg) {
        int32_t* d_cnt = fgcount;
        if (out_space == CK_HOST) { CK_TRY(ck_ensure(ctx, ctx->fgcbuf, (size_t)n * 361 * sizeof(int32_t))); d_cnt = (int32_t*)ctx->fgcbuf.p; }
        CK_TRY(k_mog2_run(ctx, ctx->mog2[mog2_handle], (const uint8_t*)ctx->goban.p, n, learning_rates, d_cnt, nullptr, 379, 379));
        if (out_space == CK_HOST) CK_TRY(ck_from_device(ctx, fgcount, d_cnt, (size_t)n * 361 * sizeof(int32_t), CK_HOST));
    }
    RegionOut ro; ro.label = region_label; ro.conf = region_conf;
    CK_TRY(cnn_predict_dev(ctx, (const uint8_t*)ctx->goban.p, n, nullptr, labels, conf, out_space, ro));
    return cnn_finish(ctx, (const uint8_t*)ctx->goban.p, n, nullptr, labels, conf, out_space, ro);
    CK_API_END(ctx)
}

int ck_mog2_band_run(ck_ctx* ctx, int handle, const uint8_t* band, int n, int in_space, const double* learning_rates,
                     int last_band, int32_t* counts, int out_space)
{
    CK_API_BEGIN(ctx)
    if (!ctx) return CK_ERR_ARG;
    if (handle < 0 || handle >= (int)ctx->mog2.size() || !ctx->mog2[handle].alive)
        return ck_fail(ctx, CK_ERR_ARG, "bad mog2 handle %d", handle);
    if (!band || !learning_rates || !counts || n <= 0) return ck_fail(ctx, CK_ERR_ARG, "NULL argument or n <= 0");
    CK_HIP(ctx, hipSetDevice(ctx->device));
    Mog2State& st = ctx->mog2[handle];
    const size_t zones = (size_t)((st.h + 19) / 20) * ((st.w + 19) / 20);
    const void* d_in;
    CK_TRY(ck_to_device(ctx, band, (size_t)n * st.h * st.w * 3, in_space, ctx->in_stage, &d_in));
    int32_t* d_cnt = counts;
    if (out_space == CK_HOST) { CK_TRY(ck_ensure(ctx, ctx->fgcbuf, (size_t)n * zones * sizeof(int32_t))); d_cnt = (int32_t*)ctx->fgcbuf.p; }
    CK_TRY(k_mog2_run(ctx, st, (const uint8_t*)d_in, n, learning_rates, d_cnt, nullptr, last_band ? st.h - 1 : -1, st.w - 1));
    if (out_space == CK_HOST) CK_TRY(ck_from_device(ctx, counts, d_cnt, (size_t)n * zones * sizeof(int32_t), CK_HOST));
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_zone_counts(ck_ctx* ctx, const uint8_t* mask, int n, int in_space, int32_t* counts, int out_space)
{
    CK_API_BEGIN(ctx)
    if (!ctx) return CK_ERR_ARG;
    if (!mask || !counts || n <= 0) return ck_fail(ctx, CK_ERR_ARG, "NULL argument or n <= 0");
    CK_HIP(ctx, hipSetDevice(ctx->device));
    const void* d_in;
    CK_TRY(ck_to_device(ctx, mask, (size_t)n * 380 * 380, in_space, ctx->in_stage2, &d_in));
    int32_t* d_cnt = counts;
    if (out_space == CK_HOST) { CK_TRY(ck_ensure(ctx, ctx->fgcbuf, (size_t)n * 361 * sizeof(int32_t))); d_cnt = (int32_t*)ctx->fgcbuf.p; }
    CK_TRY(k_zone_counts(ctx, (const uint8_t*)d_in, n, 380, d_cnt));
    if (out_space == CK_HOST) CK_TRY(ck_from_device(ctx, counts, d_cnt, (size_t)n * 361 * sizeof(int32_t), CK_HOST));
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_contour_stones(ck_ctx* ctx, const uint8_t* goban, const uint8_t* fg, int n, int side, int in_space, const int32_t* rects,
                      int rs, int re, int cs, int ce, uint8_t* stones, int16_t* zones, uint8_t* mask)
{
    CK_API_BEGIN(ctx)
    if (!ctx) return CK_ERR_ARG;
    if (!goban || !fg || !rects || !stones || n <= 0) return ck_fail(ctx, CK_ERR_ARG, "NULL argument or n <= 0");
    if (side < 19 * 4 || side > 4096) return ck_fail(ctx, CK_ERR_ARG, "goban image side %d", side);
    if (rs < 0 || cs < 0 || re > 19 || ce > 19 || re <= rs || ce <= cs)
        return ck_fail(ctx, CK_ERR_ARG, "intersection range rows [%d, %d) columns [%d, %d)", rs, re, cs, ce);
    CK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)n * side * side;
    const void *d_img, *d_fg;
    CK_TRY(ck_to_device(ctx, goban, px * 3, in_space, ctx->in_stage, &d_img));
    CK_TRY(ck_to_device(ctx, fg, px, in_space, ctx->in_stage2, &d_fg));
    CK_TRY(k_contour_stones(ctx, (const uint8_t*)d_img, (const uint8_t*)d_fg, n, side, rects, rs, re, cs, ce, stones, zones, mask));
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_contours_external(ck_ctx* ctx, const uint8_t* edges, int n, int h, int w, int in_space,
                         int32_t* counts, int32_t* table, int table_cap, int32_t* points, int points_cap)
{
    CK_API_BEGIN(ctx)
    CK_TRY(check_img(ctx, edges, n, h, w));
    if (!counts || !table || table_cap <= 0) return ck_fail(ctx, CK_ERR_ARG, "NULL table");
    const void* d_in;
    CK_TRY(ck_to_device(ctx, edges, (size_t)n * h * w, in_space, ctx->in_stage, &d_in));
    std::vector<std::vector<CkContour>> found;
    CK_TRY(k_contour_survey(ctx, (const uint8_t*)d_in, n, h, w, found));
    size_t nt = 0, np = 0;
    for (int f = 0; f < n; f++) {
        counts[f] = (int32_t)found[f].size();
        for (const CkContour& c : found[f]) {
            if ((int)nt >= table_cap) return ck_fail(ctx, CK_ERR_CAPACITY, "more than %d contours in the batch", table_cap);
            int32_t* t = table + nt * 4;
            t[0] = c.root % w; t[1] = c.root / w; t[2] = c.nvert; t[3] = (int32_t)(c.pts.size() / 2);
            nt++;
            if (points) {
                if (np + c.pts.size() / 2 > (size_t)points_cap) return ck_fail(ctx, CK_ERR_CAPACITY, "more than %d border pixels in the batch", points_cap);
                memcpy(points + np * 2, c.pts.data(), c.pts.size() * 4);
                np += c.pts.size() / 2;
            }
        }
    }
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_find_intersections(ck_ctx* ctx, const uint8_t* goban, int n, int side, int in_space, const int16_t* mtx, const int32_t* rects,
                          int16_t* grid, int16_t* lines, int32_t* nlines, uint8_t* edges)
{
    CK_API_BEGIN(ctx)
    if (!ctx) return CK_ERR_ARG;
    if (!goban || !mtx || !rects || !grid || n <= 0) return ck_fail(ctx, CK_ERR_ARG, "NULL argument or n <= 0");
    if (side < 19 * 4 || side > 4096) return ck_fail(ctx, CK_ERR_ARG, "goban image side %d", side);
    CK_HIP(ctx, hipSetDevice(ctx->device));
    const void* d_img;
    CK_TRY(ck_to_device(ctx, goban, (size_t)n * side * side * 3, in_space, ctx->in_stage, &d_img));
    const int16_t* found;
    const int32_t* counts;
    CK_TRY(k_grid_lines(ctx, (const uint8_t*)d_img, n, side, rects, &found, &counts, edges));
    if (lines) memcpy(lines, found, (size_t)n * 361 * CK_ZONE_LINES * 4 * sizeof(int16_t));
    if (nlines) memcpy(nlines, counts, (size_t)n * 361 * sizeof(int32_t));
    // update_grid, zone by zone; images are independent: a few host threads share them
    auto one_image = [&](int f) {
        int32_t seg[CK_ZONE_LINES * 4];
        int16_t* g = grid + (size_t)f * 361 * 2;
        memcpy(g, mtx, 361 * 2 * sizeof(int16_t));
        for (int z = 0; z < 361; z++) {
            const int k = counts[(size_t)f * 361 + z];
            if (!k) continue;
            const int16_t* l = found + ((size_t)f * 361 + z) * CK_ZONE_LINES * 4;
            for (int i = 0; i < k * 4; i++) seg[i] = l[i];
            ck_update_grid_host(seg, k, rects + 4 * z, g + 2 * z);
        }
    };
    ck_parallel_for(n, 8, one_image);
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_update_grid(const int32_t* lines, int k, const int32_t* box, int16_t* slot)
{
    if (!lines || !box || !slot || k < 0) return CK_ERR_ARG;
    for (int i = 0; i < k; i++)
        if (lines[4 * i] == lines[4 * i + 2] && lines[4 * i + 1] == lines[4 * i + 3]) return CK_ERR_ARG;   // zero length: the reference divides by zero
    ck_update_grid_host(lines, k, box, slot);
    return CK_OK;
}

int ck_mog2_create(ck_ctx* ctx, int h, int w, int* handle)
{
    CK_API_BEGIN(ctx)
    if (!ctx || !handle || h <= 0 || w <= 0) return CK_ERR_ARG;
    CK_HIP(ctx, hipSetDevice(ctx->device));
    int idx = -1;
    for (size_t i = 0; i < ctx->mog2.size(); i++) if (!ctx->mog2[i].alive) { idx = (int)i; break; }
    if (idx < 0) { ctx->mog2.emplace_back(); idx = (int)ctx->mog2.size() - 1; }
    Mog2State& st = ctx->mog2[idx];
    st.h = h; st.w = w; st.nframes = 0; st.alive = true;
    const size_t npx = (size_t)h * w;
    CK_TRY(ck_ensure(ctx, st.weight, npx * 5 * sizeof(float)));
    CK_TRY(ck_ensure(ctx, st.variance, npx * 5 * sizeof(float)));
    CK_TRY(ck_ensure(ctx, st.mean, npx * 15 * sizeof(float)));
    CK_TRY(ck_ensure(ctx, st.nmodes, npx));
    CK_HIP(ctx, hipMemsetAsync(st.nmodes.p, 0, npx, ctx->stream));
    *handle = idx;
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_mog2_apply(ck_ctx* ctx, int handle, const uint8_t* img3, int in_space,
                  double learning_rate, uint8_t* fgmask, int out_space)
{
    CK_API_BEGIN(ctx)
    if (!ctx) return CK_ERR_ARG;
    if (handle < 0 || handle >= (int)ctx->mog2.size() || !ctx->mog2[handle].alive)
        return ck_fail(ctx, CK_ERR_ARG, "bad mog2 handle %d", handle);
    if (!img3 || !fgmask) return ck_fail(ctx, CK_ERR_ARG, "NULL image or mask");
    CK_HIP(ctx, hipSetDevice(ctx->device));
    Mog2State& st = ctx->mog2[handle];
    const size_t npx = (size_t)st.h * st.w;
    const void* d_in;
    CK_TRY(ck_to_device(ctx, img3, npx * 3, in_space, ctx->in_stage, &d_in));
    uint8_t* d_fg = fgmask;
    if (out_space == CK_HOST) { CK_TRY(ck_ensure(ctx, ctx->out_stage, npx)); d_fg = (uint8_t*)ctx->out_stage.p; }
    CK_TRY(k_mog2_apply(ctx, st, (const uint8_t*)d_in, learning_rate, d_fg));
    if (out_space == CK_HOST) CK_TRY(ck_from_device(ctx, fgmask, d_fg, npx, CK_HOST));
    return finish(ctx);
    CK_API_END(ctx)
}

int ck_mog2_destroy(ck_ctx* ctx, int handle)
{
    CK_API_BEGIN(ctx)
    if (!ctx || handle < 0 || handle >= (int)ctx->mog2.size()) return CK_ERR_ARG;
    ctx->mog2[handle].alive = false;
    return CK_OK;
    CK_API_END(ctx)
}

}  // extern "C"
