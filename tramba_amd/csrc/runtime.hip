// libtramba_hip: error reporting, launch profiling, host-side scan-order tables.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"

namespace tramba {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------- profiling
struct ProfState {
    bool enabled = false;
    double min_units = 0.0;   // launches accounting for fewer units (bytes / flops) are not timed
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double units = 0.0;
};
static ProfState g_prof[TRAMBA_PROF_COUNT];
static std::mutex g_prof_mu;

ProfScope::ProfScope(int w, hipStream_t s, double units) : which(w), stream(s), on(false), start(nullptr)
{
    if (w < 0 || w >= TRAMBA_PROF_COUNT || !g_prof[w].enabled || units < g_prof[w].min_units) return;
    // never create events while the stream is capturing a graph
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return;
    if (hipEventCreate(&start) != hipSuccess) return;
    on = true;
    g_prof[w].units += units;
    (void)hipEventRecord(start, s);
}

ProfScope::~ProfScope()
{
    if (!on) return;
    hipEvent_t stop;
    if (hipEventCreate(&stop) != hipSuccess) return;
    (void)hipEventRecord(stop, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[which].events.emplace_back(start, stop);
}

// ---------------------------------------------------------------- device error word (common.h)
unsigned *g_deverr_host = nullptr;
static unsigned *g_deverr_dev = nullptr;
static std::once_flag g_deverr_once;

static void deverr_alloc()
{
    unsigned *h = nullptr, *d = nullptr;
    // coherent host memory mapped into every device of the process (one process per GPU is the deployment form)
    if (hipHostMalloc((void **)&h, 64, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    *h = 0u;
    if (hipHostGetDevicePointer((void **)&d, h, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostFree(h);
        return;
    }
    g_deverr_dev = d;
    set_mailbox_err_word(d);
    g_deverr_host = h;
}

MailboxCtl mailbox_ctl(hipStream_t s)
{
    if (!g_deverr_host) {
        // never allocate while the stream records a graph (an allocation is not a stream operation, and under the global
        // capture mode it invalidates the capture): a model's eager warm-up launches come first and allocate
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusNone) std::call_once(g_deverr_once, deverr_alloc);
    }
    const int t = tramba_tune_get(TRAMBA_TUNE_MAILBOX_SKIP);
    return MailboxCtl{t > 0 ? t : -1};
}

int dev_error_report(const char *file, int line)
{
    const unsigned code = __atomic_exchange_n(g_deverr_host, 0u, __ATOMIC_RELAXED);
    if (!code) return TRAMBA_OK;
    set_error("device error word 0x%x%s (seen at %s:%d): a kernel launched EARLIER by this library failed on the device and its "
              "output holds NaN", code, (code & TRAMBA_DEVERR_MAILBOX) ? " [fused scan: carry mailbox timed out]" : "", file, line);
    return TRAMBA_ERR_HIP;
}

}  // namespace tramba

using namespace tramba;

extern "C" const char *tramba_last_error(void) { return g_err; }
extern "C" int tramba_device_error(void)
{
    if (g_deverr_host && *(volatile unsigned *)g_deverr_host) return dev_error_report("tramba_device_error", 0);
    return TRAMBA_OK;
}
// 3: layernorm_bwd_parts(rows, c, dtype), shadow / slab-sum entries
// 4: fused training entries (add_layernorm, layernorm_bwd_res, dwconv_dual, merge_grad, ss2d_bwd_prep / assemble,
//    dw_unpack_grad); a_log / flags arguments of the fused scan forward / backward
// 5: the step's ends and batched launches (sod_loss_*, adam_step, multi_sum / multi_sum_strided, wgrad_parts_cl,
//    dw_pack_multi / dw_unpack_grad_multi, shuffle_norm_head_bwd_cl)
// 6: tramba_device_error, TRAMBA_TUNE_MAILBOX_SKIP (r04)
extern "C" int tramba_abi_version(void) { return 7; }

static int g_tune[TRAMBA_TUNE_COUNT] = {0};
extern "C" int tramba_tune_set(int knob, int value)
{
    TRAMBA_CHECK(knob >= 0 && knob < TRAMBA_TUNE_COUNT, "tune knob %d out of range", knob);
    g_tune[knob] = value;
    return TRAMBA_OK;
}
extern "C" int tramba_tune_get(int knob) { return knob >= 0 && knob < TRAMBA_TUNE_COUNT ? g_tune[knob] : 0; }

extern "C" int tramba_profile_enable(int which, int enable)
{
    TRAMBA_CHECK(which >= 0 && which < TRAMBA_PROF_COUNT, "profile class %d out of range", which);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[which].enabled = enable != 0;
    return TRAMBA_OK;
}

extern "C" int tramba_profile_min_units(int which, double min_units)
{
    TRAMBA_CHECK(which >= 0 && which < TRAMBA_PROF_COUNT, "profile class %d out of range", which);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[which].min_units = min_units;
    return TRAMBA_OK;
}

extern "C" int tramba_profile_read(int which, double *total_ms, double *total_units)
{
    TRAMBA_CHECK(which >= 0 && which < TRAMBA_PROF_COUNT, "profile class %d out of range", which);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfState &p = g_prof[which];
    double ms = 0.0;
    int n = 0;
    for (auto &ev : p.events) {
        (void)hipEventSynchronize(ev.second);
        float t = 0.f;
        if (hipEventElapsedTime(&t, ev.first, ev.second) == hipSuccess) {
            ms += t;
            ++n;
        }
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    p.events.clear();
    if (total_ms) *total_ms = ms;
    if (total_units) *total_units = p.units;
    p.units = 0.0;
    return n;
}

// ---------------------------------------------------------------- scan-order tables (host)
// Replaces the reference's import-time generators:
//   raster    Models/SS2D/csms6s.py:18-22
//   line      Models/SS2D/SpiralLine.py:3-82   (flat index = x + y*H, :103)
//   window    Models/SS2D/Window.py:3-35       (flat index = r*H + c)
//   dilation  Models/SS2D/Dilation.py:3-45
// Tables are built once per (family, size) on the host and cached on the device by the
// Python side; nothing here runs at import time.

extern "C" int tramba_scan_family_k(int family)
{
    switch (family) {
    case TRAMBA_SCAN_RASTER:
    case TRAMBA_SCAN_LINE:
    case TRAMBA_SCAN_WINDOW:
    case TRAMBA_SCAN_DILATION: return 4;
    case TRAMBA_SCAN_HELIX: return 8;
    default: set_error("unknown scan family %d", family); return TRAMBA_ERR_ARG;
    }
}

extern "C" int tramba_default_window(int h)
{
    switch (h) {  // reference constants, csms6s.py:107-108
    case 12: return 4;
    case 24: return 8;
    case 48: return 12;
    case 96: return 16;
    default: break;
    }
    if (h >= 64 && h % 16 == 0) return 16;
    for (int ws = 8; ws >= 2; ws /= 2)
        if (h % ws == 0 && h > ws) return ws;
    return 1;
}

static void raster_rows(int h, int w, int32_t *out)
{
    const int L = h * w;
    for (int i = 0; i < L; ++i) {
        out[i] = i;
        // direction 1 is the row-major walk of the transposed map
        out[L + i] = (i % h) * w + (i / h);
    }
    for (int i = 0; i < L; ++i) {
        out[2 * L + i] = out[L - 1 - i];
        out[3 * L + i] = out[2 * L - 1 - i];
    }
}

// Integer line rasteriser with the error-term convention of SpiralLine.py:3-24.
static void trace_line(int x0, int y0, int x1, int y1, std::vector<std::pair<int, int>> &pts)
{
    pts.clear();
    const int dx = std::abs(x1 - x0), dy = std::abs(y1 - y0);
    const int sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
    int err = dx - dy, x = x0, y = y0;
    for (;;) {
        pts.emplace_back(x, y);
        if (x == x1 && y == y1) break;
        const int e2 = 2 * err;
        const bool stepx = e2 > -dy, stepy = e2 < dx;
        if (stepx) { err -= dy; x += sx; }
        if (stepy) { err += dx; y += sy; }
    }
}

static int line_rows(int h, int w, int32_t *out)
{
    if (h != w) { set_error("line scan needs a square map (got %dx%d)", h, w); return TRAMBA_ERR_ARG; }
    const int L = h * w;
    std::vector<std::pair<int, int>> pts;
    for (int parity = 0; parity < 2; ++parity) {
        std::vector<int32_t> fwd, rev;
        auto emit = [&](int x0, int y0, int x1, int y1) {
            trace_line(x0, y0, x1, y1, pts);
            for (auto &p : pts) fwd.push_back(p.first + p.second * h);
            for (auto it = pts.rbegin(); it != pts.rend(); ++it) rev.push_back(it->first + it->second * h);
        };
        for (int r = parity; r < h; r += 2) emit(0, r, h - 1, w - 1 - r);
        int c0;
        if (parity == 0) {
            c0 = (h % 2 == 0) ? 0 : 2;
        } else {
            c0 = 1;
            if (h % 2 != 0) emit(0, w - 1, h - 1, 0);
        }
        for (int c = c0; c < w; c += 2) emit(c, w - 1, h - 1 - c, 0);
        if ((int)fwd.size() != L) {
            set_error("line scan of size %d yields %zu positions, expected %d", h, fwd.size(), L);
            return TRAMBA_ERR_UNSUPPORTED;
        }
        memcpy(out + (2 * parity) * L, fwd.data(), sizeof(int32_t) * L);
        memcpy(out + (2 * parity + 1) * L, rev.data(), sizeof(int32_t) * L);
    }
    return 4;
}

static int window_rows(int h, int w, int ws, int32_t *out)
{
    if (h != w) { set_error("window scan needs a square map"); return TRAMBA_ERR_ARG; }
    if (ws <= 0) ws = tramba_default_window(h);
    if (h % ws != 0) { set_error("window size %d does not divide %d", ws, h); return TRAMBA_ERR_ARG; }
    const int L = h * w, nw = h / ws;
    int pos = 0;
    for (int wr = 0; wr < nw; ++wr)
        for (int wc = 0; wc < nw; ++wc)
            for (int r = 0; r < ws; ++r)
                for (int c = 0; c < ws; ++c, ++pos) {
                    out[pos] = (wr * ws + r) * w + (wc * ws + c);
                    // dir 2: window grid and pixels both walked column-major
                    out[2 * L + pos] = (wc * ws + c) * w + (wr * ws + r);
                }
    for (int i = 0; i < L; ++i) {
        out[L + i] = out[L - 1 - i];
        out[3 * L + i] = out[3 * L - 1 - i];
    }
    return 4;
}

static int dilation_rows(int h, int w, int rate, int32_t *out)
{
    if (h != w) { set_error("dilation scan needs a square map"); return TRAMBA_ERR_ARG; }
    if (rate <= 0) rate = 4;  // csms6s.py:59
    const int L = h * w;
    std::vector<int32_t> base(4 * (size_t)L);
    raster_rows(h, w, base.data());
    int pos = 0;
    for (int m = 0; m < rate; ++m)
        for (int s = m; s < L; s += rate, ++pos)
            for (int k = 0; k < 4; ++k) out[k * L + pos] = base[k * (size_t)L + s];
    return 4;
}

extern "C" int tramba_scan_table(int family, int h, int w, int param, int32_t *out)
{
    TRAMBA_CHECK(out != nullptr && h > 0 && w > 0, "scan_table: bad arguments");
    const int L = h * w;
    switch (family) {
    case TRAMBA_SCAN_RASTER: raster_rows(h, w, out); return 4;
    case TRAMBA_SCAN_LINE: return line_rows(h, w, out);
    case TRAMBA_SCAN_HELIX: {
        raster_rows(h, w, out);
        int rc = line_rows(h, w, out + 4 * (size_t)L);
        return rc < 0 ? rc : 8;
    }
    case TRAMBA_SCAN_WINDOW: return window_rows(h, w, param, out);
    case TRAMBA_SCAN_DILATION: return dilation_rows(h, w, param, out);
    default: set_error("unknown scan family %d", family); return TRAMBA_ERR_ARG;
    }
}

extern "C" int tramba_scan_table_inverse(const int32_t *table, int k, int l, int32_t *inv_ptr,
                                         int32_t *inv_idx)
{
    TRAMBA_CHECK(table && inv_ptr && inv_idx && k > 0 && l > 0, "scan_table_inverse: bad arguments");
    std::vector<int32_t> count(l + 1, 0);
    for (int64_t e = 0; e < (int64_t)k * l; ++e) {
        const int p = table[e];
        TRAMBA_CHECK(p >= 0 && p < l, "table entry %d out of range [0,%d)", p, l);
        ++count[p + 1];
    }
    inv_ptr[0] = 0;
    for (int p = 0; p < l; ++p) inv_ptr[p + 1] = inv_ptr[p] + count[p + 1];
    std::vector<int32_t> cursor(inv_ptr, inv_ptr + l);
    for (int64_t e = 0; e < (int64_t)k * l; ++e) inv_idx[cursor[table[e]]++] = (int32_t)e;
    return TRAMBA_OK;
}
