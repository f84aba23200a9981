// libfvvdp_hip: FovVideoVDP per-frame visible-difference path for MI355X (gfx950 / CDNA4).
//
// Written for gfx950 only: wave64, single-wave workgroups that stream down the image for the fused pyramid
// kernel, 16-byte-per-lane coalesced HBM access on pixel-interleaved planes, LDS for neighbour exchange, no MFMA
// (5-tap stencils + pointwise + LUT: there is no contraction to feed a matrix core).  See DESIGN.md.
//
// Data layout in HBM (context scratch): Gaussian level i of frame slot s is an array [h_i][w_i][P] fp32 --
// the P temporal-channel planes of one pixel are adjacent (one aligned float4 for video, float2 for images), so
// one lane = one pixel, every load/store is 16 B (8 B) wide and all per-pixel math is thread-local.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>
#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fvvdp_hip.h"

// ------------------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(FVVDP_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char* fvvdp_last_error(void) { return g_err; }

#include "device_common.hpp"
#include "temporal_kernels.hpp"
#include "temporal_launch.hpp"
#include "band_kernel.hpp"
#include "band2_kernel.hpp"
#include "aux_kernels.hpp"
#include "psnr_kernel.hpp"
#include "resize_kernels.hpp"

// ------------------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------------------
// Testing / A-B switches of the library.  Read ONCE, when a context is created (fvvdp_ctx_create), never inside a per-frame call;
// none of them changes results beyond the grouping of fp32 partial sums.
struct CtxEnv {
    bool alloc_malloc = false;    // FVVDP_ALLOC=malloc: every scratch buffer from hipMalloc (default: large levels mapped from chunks)
    size_t vmm_chunk = 0;         // FVVDP_VMM_CHUNK_MB: chunk size of the >= 256 MB levels (default 32 MB)
    int probe_n = -1;             // FVVDP_PLACEMENT_PROBE=n: half-size level-0 candidates compared at creation (0 / 1 = none; default 6)
    int probe_extra = -1;         // FVVDP_PLACEMENT_EXTRA=n: further candidates tried while no pair reaches probe_mixed (default 8)
    float probe_mixed = 6.75f;    // FVVDP_PLACEMENT_MIXED_TBS: a pair written at once at this rate [TB/s] lies in both classes of memory
    bool inrange_off = false;     // FVVDP_BAND_INRANGE=0: always the pyramid kernels with clamps
    int fuse_mode = -1;           // FVVDP_BAND_FUSE=0 / 1: two-level pyramid kernel never / wherever valid (default: large levels)
    int band_cr = 0, band2_kr = 0, band2_kr2 = -1, band2_wpb = 0;   // FVVDP_BAND_CR, FVVDP_BAND2_KR, _KR2, _WPB: work decomposition overrides
    int k1_ticket = -1;           // FVVDP_K1_TICKET=0 / 1: temporal kernel (16-slot ring) without / with its block counter
    int level0_split = -1;        // FVVDP_LEVEL0_SPLIT=1: level 0 in two ranges for ANY context, no choice among candidates (tests)
    int band2_ticket = -1;        // FVVDP_BAND2_TICKET=0 / 1: two-level pyramid kernel with the static split of work / per-XCD counters
    bool temporal_scalar = false; // FVVDP_TEMPORAL_SCALAR=1: the per-pixel temporal kernels (fallbacks for unaligned sizes)
    bool fov_no_rhomap = false;   // FVVDP_FOV_NO_RHOMAP=1: foveated kernels evaluate the rho coordinate per pixel
    bool debug_variant = false;   // FVVDP_DEBUG_VARIANT=1: print which kernel variants are launched (tests)
    bool yuv_general = false;     // FVVDP_YUV_GENERAL_MATRIX=1: YUV ingest with the nine-term colour matrix also where it has the ITU shape (tests)
};
static CtxEnv read_env() {
    CtxEnv e;
    auto num = [](const char* name, int dflt) { const char* v = getenv(name); return (v && v[0]) ? atoi(v) : dflt; };
    if (const char* v = getenv("FVVDP_ALLOC")) e.alloc_malloc = strcmp(v, "malloc") == 0;
    e.vmm_chunk = (size_t)(num("FVVDP_VMM_CHUNK_MB", 0) > 0 ? num("FVVDP_VMM_CHUNK_MB", 0) : 0) << 20;
    e.probe_n = num("FVVDP_PLACEMENT_PROBE", -1);
    e.probe_extra = num("FVVDP_PLACEMENT_EXTRA", -1);
    if (const char* v = getenv("FVVDP_PLACEMENT_MIXED_TBS")) e.probe_mixed = (float)atof(v);
    if (const char* v = getenv("FVVDP_BAND_INRANGE")) e.inrange_off = v[0] == '0';
    if (const char* v = getenv("FVVDP_BAND_FUSE")) e.fuse_mode = (v[0] == '0' || v[0] == '1') ? v[0] - '0' : -1;
    e.band_cr = num("FVVDP_BAND_CR", 0);
    e.band2_kr = num("FVVDP_BAND2_KR", 0);
    e.band2_kr2 = num("FVVDP_BAND2_KR2", -1);
    e.band2_wpb = num("FVVDP_BAND2_WPB", 0);
    if (const char* v = getenv("FVVDP_K1_TICKET")) e.k1_ticket = v[0] != '0' ? 1 : 0;
    if (const char* v = getenv("FVVDP_BAND2_TICKET")) e.band2_ticket = v[0] != '0' ? 1 : 0;
    if (const char* v = getenv("FVVDP_LEVEL0_SPLIT")) e.level0_split = v[0] != '0' ? 1 : 0;
    e.temporal_scalar = getenv("FVVDP_TEMPORAL_SCALAR") != nullptr;
    e.fov_no_rhomap = getenv("FVVDP_FOV_NO_RHOMAP") != nullptr;
    e.debug_variant = getenv("FVVDP_DEBUG_VARIANT") != nullptr;
    e.yuv_general = getenv("FVVDP_YUV_GENERAL_MATRIX") != nullptr;
    return e;
}

struct VmmBlock {               // a buffer that came from the virtual-memory API (see dev_alloc): one reserved range, one or
    void* ptr;                  // more physical allocations mapped into it
    size_t size;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};
struct fvvdp_ctx {
    CtxEnv env;
    int W = 0, H = 0, n_bands = 0, P = 0, max_frames = 0;
    fvvdp_params prm{};
    double rho_band[FVVDP_MAX_BANDS + 1]{};
    int lw[FVVDP_MAX_BANDS + 1]{}, lh[FVVDP_MAX_BANDS + 1]{};
    float* level[FVVDP_MAX_BANDS + 1]{};
    float* level0_hi = nullptr;   // != nullptr: level 0 lives in TWO ranges -- even frame slots in level[0], odd slots here (place_level0)
    // choice of the level-0 buffer at creation (choose_level0): sel_n candidates timed, sel_us[k] = temporal kernel + pyramid pass
    // in us per frame on candidate k, sel_kept = the one in use (-1: no comparison); sel_phase 9 = settled (the only state a
    // caller can observe: the comparison runs inside fvvdp_ctx_create)
    int sel_phase = 9;
    int sel_n = 0;
    int sel_kept = -1;
    float sel_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int level0_kind = 0;          // 0 hipMalloc, 1 mapped from physical chunks
    int level0_kind_hi = -1;      // the same for the range of the odd slots (-1: level 0 is one range)
    // host synchronisations / allocations / frees made INSIDE per-frame entry points since creation (fvvdp_ctx_call_stats): the
    // standard video path makes none (SURVEY 8(b) "allocated once in ctx_create")
    long long n_sync = 0, n_alloc = 0, n_free = 0;
    bool created = false;         // fvvdp_ctx_create has returned: from here on allocations / syncs are counted
    // what is known about the values in level 0 (luminance_range): sustained planes in [lum_lo, lum_hi], no plane's values
    // further apart than lum_width; lum_known = false after fvvdp_load_channels_planar or a source without a display model
    bool lum_known = false;
    int lum_state = 0;            // 0 nothing written yet, 1 range known, 2 unknown (sticky)
    int lum_hold = 0;             // > 0: an outer call has set the range, inner calls leave it alone
    int lum_top = 0;              // highest level-0 slot written since the range bookkeeping last started afresh
    float lum_lo = 0.0f, lum_hi = 0.0f, lum_width = 0.0f;
    float* partial = nullptr;
    long long partial_off[FVVDP_MAX_BANDS]{};
    int max_blk[FVVDP_MAX_BANDS]{};
    int* d_ticket = nullptr;      // block counter of the resident temporal kernel (TemporalArgs::ticket), zeroed before every launch
    size_t partial_floats = 0;
    float4* csf = nullptr;        // [n_bands][32] slope-form records of the 1-D tables
    float4* csf_y = nullptr;      // [32] {Y_log[i],0,0,0}
    bool csf_set = false;
    float y_first = 0, y_inv_step = 0, y_lo = 0, y_hi = 0;
    std::vector<float> h_lut3[2];              // host copies of the full 32^3 LUTs (foveated mode)
    float h_axes[3][FVVDP_LUT_N]{};            // Y_log, rho_log, ecc_sqrt
    float* d_axes = nullptr;                   // [3][32]
    float4* sublut[FVVDP_MAX_BANDS]{};         // per-band rho slices, rebuilt when the geometry changes
    float4* rmap[FVVDP_MAX_BANDS]{};           // per-band rho-axis coordinates of every pixel pair (stock geometry)
    int sub_rw[FVVDP_MAX_BANDS]{}, sub_ilo[FVVDP_MAX_BANDS]{};
    fvvdp_geom sub_geom{};
    const float* map_vx[FVVDP_MAX_BANDS]{};    // user-geometry maps (owned by the caller)
    const float* map_vy[FVVDP_MAX_BANDS]{};
    const float* map_rm[FVVDP_MAX_BANDS]{};
    float map_rm_max[FVVDP_MAX_BANDS]{}, map_rm_min[FVVDP_MAX_BANDS]{};
    bool maps_set = false;
    bool sub_valid = false;
    float rho_lo = 0, rho_hi = 0, ecc_lo = 0, ecc_hi = 0;
    bool lut3_set[2] = {false, false};
    float* d_fix = nullptr;       // [max_frames][2]
    float* d_taps = nullptr;      // [2][FVVDP_MAX_TAPS]
    int* d_idx = nullptr;         // [max_frames + FVVDP_MAX_TAPS]
    float* lum_buf = nullptr;     // two-pass temporal path (33..64 taps): fp32 luminance frames [2][lum_frames][HW]
    size_t lum_floats = 0;
    float* heat[FVVDP_MAX_BANDS + 1]{};   // heat-map accumulation images of levels >= 1, allocated on first use
    unsigned int* colour_ws = nullptr;    // colouring workspace per frame: range[2] + hist[1024] + curve[1024], then lin01[1024]
    size_t scratch = 0;
    std::vector<VmmBlock> vmm;     // buffers that came from the virtual-memory API (freed by vmm_free_all)
    long long wave_capacity = 4096;   // resident single-wave workgroups of the band kernel on the whole chip
    long long wave_capacity2 = 4096;  // ... of the two-level kernel (band2_kernel)
    // timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[FVVDP_MAX_BANDS + 2];
    float t_ms[FVVDP_MAX_BANDS + 2]{};
    int t_cnt[FVVDP_MAX_BANDS + 2]{};
};

// The large pyramid levels are NOT taken from hipMalloc but mapped from physical chunks of 32 MB through the virtual-memory
// API (hipMemCreate / hipMemMap into one reserved range).  Reason, measured (profiles/r04_level0_chunks.md): on a box whose
// memory is free, hipMalloc backs the 8 GB of a 4K x 60 level 0 with one physically contiguous range, and on such a range the
// temporal kernel -- 6 B read, 16 B written per pixel and frame, all of it streaming -- runs at 37.5 us per 4K frame (4.9
// TB/s); on the same buffer mapped from chunks of 2 ... 128 MB it runs at 32.0-32.6 us (5.7 TB/s), reproducibly, and the
// pyramid kernels that read the buffer do not change (34.0 us for levels 0+1 either way).  This was the "placement mode" of
// rounds 2-3: a hipMalloc on a box with fragmented free memory happens to be pieced together the same way.
// FVVDP_ALLOC=malloc goes back to hipMalloc (A/B runs), FVVDP_VMM_CHUNK_MB overrides the chunk size.
// tools/microbench/chunks.hip (profiles/r04_level0_chunks.md): it is the WRITES that a contiguous range slows down -- write only
// 5.8-6.0 -> 6.9-7.1 TB/s, copy 5.3-5.4 -> 6.0-6.1, reads unchanged at 6.1-6.4 -- so every level that a kernel writes is mapped
// this way, the smaller ones (16 ... 256 MB) from 4 MB chunks.
static const size_t VMM_MIN_BYTES = (size_t)16 << 20;      // buffers below this stay with hipMalloc
static const size_t VMM_BIG_BYTES = (size_t)256 << 20;     // from here on: 32 MB chunks (FVVDP_VMM_CHUNK_MB), below: 4 MB

static int vmm_alloc(fvvdp_ctx* c, void** out, size_t bytes);
static void vmm_free_all(fvvdp_ctx* c);

template <typename T>
static int dev_alloc(fvvdp_ctx* c, T** p, size_t count) {
    void* q = nullptr;
    const size_t bytes = count * sizeof(T);
    if (!c->env.alloc_malloc && bytes >= VMM_MIN_BYTES && vmm_alloc(c, &q, bytes) == FVVDP_OK) {
        // mapped
    } else {
        q = nullptr;
        (void)hipGetLastError();                   // (a device without the virtual-memory API: plain allocation)
        hipError_t e = hipMalloc(&q, bytes);
        if (e != hipSuccess) return fail(FVVDP_ENOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    *p = reinterpret_cast<T*>(q);
    c->scratch += bytes;
    if (c->created) c->n_alloc += 1;               // an allocation inside a per-frame call (first use of an optional path)
    return FVVDP_OK;
}

// where frame slot `slot0 + f` of pyramid level L lies (device_common.hpp, L0Addr): level 0 in one or two ranges, the others in one
static L0Addr level_addr(const fvvdp_ctx* c, int L, int slot0) {
    L0Addr a;
    const size_t fr = (size_t)c->lw[L] * c->lh[L] * c->P;
    a.lo = c->level[L];
    a.slot0 = slot0;
    if (L == 0 && c->level0_hi) { a.hi = c->level0_hi; a.half_stride = fr; }
    else { a.hi = c->level[L] + fr; a.half_stride = 2 * fr; }
    return a;
}

struct Timed {
    fvvdp_ctx* c;
    int id;
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    Timed(fvvdp_ctx* c_, int id_, hipStream_t st_) : c(c_), id(id_), st(st_) {
        if (c->timing) {
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, st);
        }
    }
    ~Timed() {
        if (c->timing) {
            (void)hipEventRecord(e1, st);
            c->ev[id].push_back({e0, e1});
        }
    }
};

static int vmm_alloc(fvvdp_ctx* c, void** out, size_t bytes) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran == 0) gran = (size_t)2 << 20;
    size_t chunk = bytes >= VMM_BIG_BYTES ? (size_t)32 << 20 : (size_t)4 << 20;
    if (c->env.vmm_chunk >= gran && bytes >= VMM_BIG_BYTES) chunk = c->env.vmm_chunk;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t size = (bytes + chunk - 1) / chunk * chunk;
    const size_t n = size / chunk;
    VmmBlock b;
    b.size = size;
    b.ptr = nullptr;
    hipError_t e = hipMemAddressReserve(&b.ptr, size, gran, nullptr, 0);
    if (e != hipSuccess) return fail(FVVDP_ENOMEM, "hipMemAddressReserve(%zu) failed: %s", size, hipGetErrorString(e));
    for (size_t i = 0; i < n && e == hipSuccess; ++i) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, chunk, &prop, 0);
        if (e == hipSuccess) b.handles.push_back(h);
    }
    size_t mapped = 0;
    for (size_t i = 0; i < n && e == hipSuccess; ++i) {
        e = hipMemMap(static_cast<char*>(b.ptr) + i * chunk, chunk, 0, b.handles[i], 0);
        if (e == hipSuccess) mapped = (i + 1) * chunk;
    }
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (e == hipSuccess) e = hipMemSetAccess(b.ptr, size, &acc, 1);
    if (e != hipSuccess) {
        if (mapped) (void)hipMemUnmap(b.ptr, mapped);
        for (auto h : b.handles) (void)hipMemRelease(h);
        (void)hipMemAddressFree(b.ptr, size);
        return fail(FVVDP_ENOMEM, "mapping %zu chunks of %zu bytes failed: %s", n, chunk, hipGetErrorString(e));
    }
    c->vmm.push_back(b);
    *out = b.ptr;
    return FVVDP_OK;
}

static bool vmm_owns(const fvvdp_ctx* c, const void* p) {
    for (const VmmBlock& b : c->vmm)
        if (b.ptr == p) return true;
    return false;
}

// buffers of the virtual-memory API are released together by vmm_free_all (context destruction)
static void dev_free(fvvdp_ctx* c, void* p) {
    if (p && !vmm_owns(c, p)) (void)hipFree(p);
}

static void vmm_free_one(fvvdp_ctx* c, void* p) {
    for (size_t i = 0; i < c->vmm.size(); ++i)
        if (c->vmm[i].ptr == p) {
            VmmBlock& b = c->vmm[i];
            (void)hipMemUnmap(b.ptr, b.size);
            for (auto h : b.handles) (void)hipMemRelease(h);
            (void)hipMemAddressFree(b.ptr, b.size);
            c->vmm.erase(c->vmm.begin() + (long)i);
            return;
        }
}

static void vmm_free_all(fvvdp_ctx* c) {
    for (VmmBlock& b : c->vmm) {
        (void)hipMemUnmap(b.ptr, b.size);
        for (auto h : b.handles) (void)hipMemRelease(h);
        (void)hipMemAddressFree(b.ptr, b.size);
    }
    c->vmm.clear();
}

// strips cover coarse columns [0,62), [62,122), ... (see band_kernel)
static int band_strips(int wc) { return wc <= 62 ? 1 : 1 + (wc - 62 + STRIP_J - 1) / STRIP_J; }

static int band2_strips(int wb) { return (wb + F2_PITCH - 1) / F2_PITCH; }

// two-level kernel: a chunk of kr level-C rows costs 2*kr steps of level A plus 7 halo steps and the prologue
static void chunking2(int hc, int n_strips, int n, long long capacity, int kr_override, int& n_chunks, int& kr) {
    double best = 1e300;
    kr = hc;
    for (int cand = 1; cand <= hc; ++cand) {
        const long long chunks = (hc + cand - 1) / cand;
        const long long waves = (long long)n * n_strips * chunks;
        const long long rounds = (waves + capacity - 1) / capacity;
        const double cost = (double)rounds * (2.0 * cand + 9.0) * (1.0 + 1e-4 * (double)chunks);
        if (cost < best) { best = cost; kr = cand; }
    }
    if (kr_override >= 1) kr = kr_override > hc ? hc : kr_override;      // tuning override (FVVDP_BAND2_KR)
    n_chunks = (hc + kr - 1) / kr;
}

static void chunking(int hc, int n_strips, int n, long long capacity, int cr_override, int& n_chunks, int& cr) {
    // Every single-wave workgroup does the same amount of work (cr steps + the prologue for the two halo coarse
    // rows, whose 7 extra fine rows are re-read from HBM: weighted as 8 steps), so the launch proceeds in "rounds"
    // of `capacity` resident waves.  Pick the chunk height that minimises rounds x per-wave cost.
    double best = 1e300;
    cr = hc;
    for (int cand = 2; cand <= hc || cand == 2; ++cand) {
        const int c = cand > hc ? hc : cand;
        const long long chunks = (hc + c - 1) / c;
        const long long waves = (long long)n * n_strips * chunks;
        const long long rounds = (waves + capacity - 1) / capacity;
        const double cost = (double)rounds * ((double)c + 8.0) * (1.0 + 1e-4 * (double)chunks);
        if (cost < best) { best = cost; cr = c; }
        if (cand >= hc) break;
    }
    if (cr_override >= 1) cr = cr_override > hc ? hc : cr_override;      // tuning override (FVVDP_BAND_CR)
    n_chunks = (hc + cr - 1) / cr;
}

static void choose_level0(fvvdp_ctx* c);

extern "C" int fvvdp_ctx_create(fvvdp_ctx** out, int width, int height, int n_bands, int planes, int max_frames,
                                const double* h_rho_band, const fvvdp_params* prm) {
    if (!out || !prm || !h_rho_band) return fail(FVVDP_EINVAL, "null argument");
    if (planes != 2 && planes != 4) return fail(FVVDP_EINVAL, "planes must be 2 (image) or 4 (video), got %d", planes);
    if (n_bands < 1 || n_bands > FVVDP_MAX_BANDS) return fail(FVVDP_EINVAL, "n_bands %d out of range", n_bands);
    if (max_frames < 1) return fail(FVVDP_EINVAL, "max_frames must be >= 1");
    if (width < 4 || height < 4) return fail(FVVDP_EINVAL, "frame %dx%d too small", width, height);
    fvvdp_ctx* c = new fvvdp_ctx();
    c->env = read_env();
    c->W = width;
    c->H = height;
    c->n_bands = n_bands;
    c->P = planes;
    c->max_frames = max_frames;
    c->prm = *prm;
    int w = width, h = height;
    for (int i = 0; i <= n_bands; ++i) {
        c->lw[i] = w;
        c->lh[i] = h;
        c->rho_band[i] = h_rho_band[i];
        if (i < n_bands && (w < 2 || h < 2)) {
            delete c;
            return fail(FVVDP_EINVAL, "pyramid level %d is %dx%d: too many bands for this frame size", i, w, h);
        }
        w = (w + 1) / 2;
        h = (h + 1) / 2;
    }
    int rc = FVVDP_OK;
    for (int i = 0; i <= n_bands && rc == FVVDP_OK; ++i) {
        rc = dev_alloc(c, &c->level[i], (size_t)max_frames * c->lw[i] * c->lh[i] * planes);
    }
    size_t off = 0;
    for (int b = 0; b < n_bands; ++b) {
        const int n_strips = band_strips(c->lw[b + 1]);
        const int max_chunks = (c->lh[b + 1] + 1) / 2;
        c->max_blk[b] = n_strips * max_chunks;
    }
    for (int b = 0; b + 1 < n_bands; ++b) {       // two-level launches (bands b, b+1 share one work decomposition)
        const int blk2 = band2_strips(c->lw[b + 1]) * c->lh[b + 2];
        if (blk2 > c->max_blk[b]) c->max_blk[b] = blk2;
        if (blk2 > c->max_blk[b + 1]) c->max_blk[b + 1] = blk2;
    }
    for (int b = 0; b < n_bands; ++b) {
        c->partial_off[b] = (long long)off;
        off += (size_t)max_frames * c->max_blk[b] * 2;
    }
    c->partial_floats = off;
    if (rc == FVVDP_OK) rc = dev_alloc(c, &c->partial, off);
    if (rc == FVVDP_OK) {
        float* q = nullptr;
        rc = dev_alloc(c, &q, 16);
        c->d_ticket = reinterpret_cast<int*>(q);
    }
    if (rc == FVVDP_OK) rc = dev_alloc(c, &c->csf, (size_t)n_bands * FVVDP_LUT_N);
    if (rc == FVVDP_OK) rc = dev_alloc(c, &c->csf_y, (size_t)FVVDP_LUT_N);
    if (rc == FVVDP_OK) rc = dev_alloc(c, &c->d_fix, (size_t)max_frames * 2);
    if (rc == FVVDP_OK) rc = dev_alloc(c, &c->d_taps, (size_t)2 * FVVDP_MAX_TAPS);
    if (rc == FVVDP_OK) rc = dev_alloc(c, &c->d_idx, (size_t)max_frames + FVVDP_MAX_TAPS);
    if (rc != FVVDP_OK) {
        fvvdp_ctx_destroy(c);
        return rc;
    }
    {
        int dev = 0, cus = 256, per_cu = 16;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        }
        hipError_t e = (planes == 4)
            ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, band_kernel<4, false, 0>, 64, 0)
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, band_kernel<2, false, 0>, 64, 0);
        if (e != hipSuccess || per_cu < 1) per_cu = 16;
        c->wave_capacity = (long long)per_cu * cus;
        int per_cu2 = 16;
        e = (planes == 4) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, band2_kernel<4, false>, 64, 0)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, band2_kernel<2, false>, 64, 0);
        if (e != hipSuccess || per_cu2 < 1) per_cu2 = 16;
        c->wave_capacity2 = (long long)per_cu2 * cus;
    }
    choose_level0(c);
    if (!c->level[0]) {                            // the candidates and the fall-back to one range all failed to allocate
        fvvdp_ctx_destroy(c);
        return fail(FVVDP_ENOMEM, "no device memory for pyramid level 0");
    }
    c->created = true;
    *out = c;
    return FVVDP_OK;
}

extern "C" void fvvdp_ctx_destroy(fvvdp_ctx* c) {
    if (!c) return;
    (void)hipDeviceSynchronize();                  // nothing may still read the scratch
    for (int i = 0; i <= FVVDP_MAX_BANDS; ++i)
        dev_free(c, c->level[i]);
    dev_free(c, c->level0_hi);
    dev_free(c, c->partial);
    dev_free(c, c->d_ticket);
    if (c->csf) (void)hipFree(c->csf);
    if (c->csf_y) (void)hipFree(c->csf_y);
    for (int i = 0; i <= FVVDP_MAX_BANDS; ++i)
        dev_free(c, c->heat[i]);
    if (c->colour_ws) (void)hipFree(c->colour_ws);
    if (c->d_fix) (void)hipFree(c->d_fix);
    if (c->d_taps) (void)hipFree(c->d_taps);
    if (c->d_idx) (void)hipFree(c->d_idx);
    if (c->lum_buf) (void)hipFree(c->lum_buf);
    if (c->d_axes) (void)hipFree(c->d_axes);
    for (int b = 0; b < FVVDP_MAX_BANDS; ++b) {
        dev_free(c, c->sublut[b]);
        dev_free(c, c->rmap[b]);
    }
    vmm_free_all(c);
    for (auto& v : c->ev)
        for (auto& pr : v) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    delete c;
}

extern "C" int fvvdp_ctx_level_size(const fvvdp_ctx* c, int level, int* w, int* h) {
    if (!c || level < 0 || level > c->n_bands) return fail(FVVDP_EINVAL, "bad level %d", level);
    if (w) *w = c->lw[level];
    if (h) *h = c->lh[level];
    return FVVDP_OK;
}

extern "C" size_t fvvdp_ctx_scratch_bytes(const fvvdp_ctx* c) { return c ? c->scratch : 0; }

extern "C" int fvvdp_ctx_set_csf_1d(fvvdp_ctx* c, const float* h_Y_log, const float* h_S_log) {
    if (!c || !h_Y_log || !h_S_log) return fail(FVVDP_EINVAL, "null argument");
    // Record i of a band holds the table value at knot i and the step to knot i+1 for both temporal channels:
    // the kernel evaluates v[i] + f*(v[i+1]-v[i]) with f from the (uniform) knot grid.
    std::vector<float4> rec((size_t)c->n_bands * FVVDP_LUT_N);
    for (int b = 0; b < c->n_bands; ++b)
        for (int i = 0; i < FVVDP_LUT_N; ++i) {
            const float* v0 = h_S_log + ((size_t)b * 2 + 0) * FVVDP_LUT_N;
            const float* v1 = h_S_log + ((size_t)b * 2 + 1) * FVVDP_LUT_N;
            const int j = i + 1 < FVVDP_LUT_N ? i + 1 : i;
            rec[(size_t)b * FVVDP_LUT_N + i] = make_float4(v0[i], v1[i], v0[j] - v0[i], v1[j] - v1[i]);
        }
    HIP_TRY(hipMemcpy(c->csf, rec.data(), rec.size() * sizeof(float4), hipMemcpyHostToDevice));
    c->y_first = h_Y_log[0];
    c->y_inv_step = (float)(FVVDP_LUT_N - 1) / (h_Y_log[FVVDP_LUT_N - 1] - h_Y_log[0]);
    c->y_lo = exp2f(h_Y_log[0]);
    c->y_hi = exp2f(h_Y_log[FVVDP_LUT_N - 1]);
    c->csf_set = true;
    return FVVDP_OK;
}

extern "C" int fvvdp_ctx_set_csf_3d(fvvdp_ctx* c, int tc, const float* h_S_log, const float* h_Y_log,
                                    const float* h_rho_log, const float* h_ecc_sqrt) {
    if (!c || !h_S_log || !h_Y_log || !h_rho_log || !h_ecc_sqrt) return fail(FVVDP_EINVAL, "null argument");
    if (tc < 0 || tc > 1) return fail(FVVDP_EINVAL, "temporal_channel must be 0 or 1");
    const size_t n3 = (size_t)FVVDP_LUT_N * FVVDP_LUT_N * FVVDP_LUT_N;
    int rc = FVVDP_OK;
    if (!c->d_axes) rc = dev_alloc(c, &c->d_axes, (size_t)3 * FVVDP_LUT_N);
    if (rc != FVVDP_OK) return rc;
    c->h_lut3[tc].assign(h_S_log, h_S_log + n3);
    for (int i = 0; i < FVVDP_LUT_N; ++i) {
        c->h_axes[0][i] = h_Y_log[i];
        c->h_axes[1][i] = h_rho_log[i];
        c->h_axes[2][i] = h_ecc_sqrt[i];
    }
    HIP_TRY(hipMemcpy(c->d_axes, c->h_axes, sizeof(c->h_axes), hipMemcpyHostToDevice));
    c->y_lo = exp2f(h_Y_log[0]);
    c->y_hi = exp2f(h_Y_log[FVVDP_LUT_N - 1]);
    c->sub_valid = false;
    c->rho_lo = exp2f(h_rho_log[0]);
    c->rho_hi = exp2f(h_rho_log[FVVDP_LUT_N - 1]);
    c->ecc_lo = h_ecc_sqrt[0] * h_ecc_sqrt[0];
    c->ecc_hi = h_ecc_sqrt[FVVDP_LUT_N - 1] * h_ecc_sqrt[FVVDP_LUT_N - 1];
    c->lut3_set[tc] = true;
    return FVVDP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// stage 1 launchers
// ------------------------------------------------------------------------------------------------------------
// A caller-built table states the range of its entries in L_min / L_max (optional).  The kernels ENFORCE a stated range on every
// entry they read (lut_entry in temporal_kernels.hpp), so what clamps_never_bind() derives from it holds for any table contents.
static bool eotf_lut_range_stated(const fvvdp_eotf* e) {
    return e->L_max > e->L_min && e->L_min >= 0.0f && std::isfinite(e->L_max);
}
static EotfDev make_eotf(const fvvdp_eotf* e) {
    EotfDev d;
    d.kind = e->kind;
    d.scale = e->Y_peak - e->Y_black;
    d.y_black = e->Y_black;
    d.y_peak = e->Y_peak;
    d.gamma = e->gamma;
    d.l_min = e->L_min;
    d.l_max = e->L_max;
    d.lut = e->d_lut;
    if (e->kind == FVVDP_EOTF_LUT && !eotf_lut_range_stated(e)) {
        // no (usable) range stated for the table: its entries pass as they are, and luminance_range() treats level 0 as unknown
        d.l_min = -FLT_MAX;
        d.l_max = FLT_MAX;
    }
    return d;
}

// context-owned buffer of fp32 luminance frames for the two-pass temporal paths (33..64 taps), grown on demand
static int grow_lum_buf(fvvdp_ctx* c, size_t need_floats, int fl, hipStream_t st) {
    if (c->lum_floats >= need_floats) return FVVDP_OK;
    HIP_TRY(hipStreamSynchronize(st));                       // earlier calls may still read the old buffer
    c->n_sync += 1;
    if (c->lum_buf) { (void)hipFree(c->lum_buf); c->n_free += 1; }
    c->lum_buf = nullptr;
    c->lum_floats = 0;
    void* q = nullptr;
    if (hipMalloc(&q, need_floats * sizeof(float)) != hipSuccess)
        return fail(FVVDP_ENOMEM, "hipMalloc(%zu bytes) for the luminance frames of a %d-tap filter failed", need_floats * sizeof(float), fl);
    c->lum_buf = reinterpret_cast<float*>(q);
    c->n_alloc += 1;
    c->lum_floats = need_floats;
    c->scratch += need_floats * sizeof(float);
    return FVVDP_OK;
}

// Range of the values the temporal kernels write into level 0, from the display model (every closed-form model clamps its
// output, eotf_one in temporal_kernels.hpp; a code-value table carries the range of its entries in L_min / L_max), the RGB->Y
// weights and the filter taps: a filter output lies in [s+ lo + s- hi, s+ hi + s- lo] with s+ / s- the sums of its positive /
// negative taps.  The pyramid kernels use it to drop clamps that provably never bind (band2_kernel<P, true>).
// Slots of level 0 filled by different calls may be evaluated together, so the range of a context only WIDENS, and one call of
// unknown range makes it unknown -- until a call rewrites every slot written since the last such call (slots [0, n) with n >= the
// highest slot in use: what every predict() does), which starts the bookkeeping afresh.
static void luminance_slots(fvvdp_ctx* c, int slot0, int n_out) {
    if (c->lum_hold) return;
    if (slot0 == 0 && n_out >= c->lum_top) {
        c->lum_state = 0;
        c->lum_known = false;
        c->lum_top = 0;
    }
    if (slot0 + n_out > c->lum_top) c->lum_top = slot0 + n_out;
}
static void luminance_unknown(fvvdp_ctx* c) {
    c->lum_state = 2;
}
static void luminance_range(fvvdp_ctx* c, const fvvdp_eotf* e, int C, const float* h_rgb2y, const float* h_taps, int fl) {
    if (c->lum_hold || c->lum_state == 2) return;            // (inner pass of the two-pass temporal path: range set by the outer call)
    luminance_unknown(c);                                    // unless everything below works out
    const int was = c->lum_known ? 1 : 0;
    float lo = 0.0f, hi = 0.0f;
    switch (e->kind) {
        case FVVDP_EOTF_SRGB: case FVVDP_EOTF_GAMMA: lo = e->Y_black; hi = e->Y_peak; break;
        case FVVDP_EOTF_PQ: case FVVDP_EOTF_LINEAR: lo = 0.005f + e->Y_black; hi = e->Y_peak + e->Y_black; break;
        case FVVDP_EOTF_ABSOLUTE: case FVVDP_EOTF_LUT: lo = e->L_min; hi = e->L_max; break;
        default: return;                                     // FVVDP_EOTF_NONE: the caller's luminance, unknown
    }
    if (!(hi > lo) || !(lo >= 0.0f) || !std::isfinite(hi)) return;
    if (C == 3) {
        double sw = 0.0;
        for (int k = 0; k < 3; ++k) {
            if (!(h_rgb2y[k] >= 0.0f)) return;
            sw += h_rgb2y[k];
        }
        lo = (float)(lo * sw);
        hi = (float)(hi * sw);
    }
    double width = 0.0, lo0 = lo, hi0 = hi;
    const int n_ch = c->P / 2;
    for (int ch = 0; ch < n_ch; ++ch) {
        double sp = 0.0, sn = 0.0;
        for (int k = 0; k < fl; ++k) {
            const double t = h_taps[ch * fl + k];
            if (t > 0) sp += t; else sn += t;
        }
        if (c->P == 2) { sp = 1.0; sn = 0.0; }               // still images: the frame itself
        const double mn = sp * lo + sn * hi, mx = sp * hi + sn * lo;
        if (ch == 0) { lo0 = mn; hi0 = mx; }
        if (mx - mn > width) width = mx - mn;
    }
    // 1e-3 of slack for the rounding of the table products, the fast log2 / exp2 of the closed-form models and the filters
    const float nlo = (float)(lo0 * (1.0 - 1e-3)), nhi = (float)(hi0 * (1.0 + 1e-3)), nw = (float)(width * (1.0 + 1e-3));
    if (!(std::isfinite(nhi) && nlo > 0.0f)) return;
    c->lum_lo = was ? fminf(c->lum_lo, nlo) : nlo;
    c->lum_hi = was ? fmaxf(c->lum_hi, nhi) : nhi;
    c->lum_width = was ? fmaxf(c->lum_width, nw) : nw;
    c->lum_known = true;
    c->lum_state = 1;
}

// May the pyramid pass over levels [b, b + n_levels] drop its clamps (band2_kernel INRANGE)?  Every reduce step on the way
// must be a convex combination (the reference's right-edge fix-up is not when the row and column parities of a level
// differ, fvvdp_lpyr_dec.py:198-205: weights summing to 1.25 or 0.75), so that every level and every expanded level stays
// inside the range of level 0.
static bool clamps_never_bind(const fvvdp_ctx* c, int b, int n_levels) {
    if (!c->lum_known || c->lum_state != 1) return false;
    if (c->env.inrange_off) return false;                                  // A/B runs, tests
    for (int i = 0; i < b + n_levels; ++i)
        if ((c->lw[i] & 1) != (c->lh[i] & 1)) return false;
    const float lo = c->lum_lo, hi = c->lum_hi;
    return lo >= c->prm.lbkg_min &&                                   // (1) L_bkg = max(., lbkg_min) is the identity
           c->prm.contrast_max * lo > c->lum_width &&                 // (2) contrast <= contrast_max never binds
           lo >= 2.0f * c->y_lo && hi <= 0.5f * c->y_hi;             // (3) well inside the Y axis of the CSF table
}

// h_frame_idx1: per-stream frame indices of the reference stream, or nullptr = the same as h_frame_idx
static int temporal_channels_core(fvvdp_ctx* c, const void* d_test, const void* d_ref, int dtype, int C,
                                  size_t chan_stride, size_t frame_stride, const fvvdp_eotf* eotf,
                                  const float* h_rgb2y, const int32_t* h_frame_idx, const int32_t* h_frame_idx1,
                                  const float* h_taps, int fl, int n_out, int slot0, int32_t* d_oob_flag, void* stream) {
    if (!c || !d_test || !d_ref || !eotf || !h_frame_idx || !h_taps) return fail(FVVDP_EINVAL, "null argument");
    if (dtype < FVVDP_U8 || dtype > FVVDP_F32) return fail(FVVDP_EINVAL, "Only uint8, uint16 and float32 is currently supported");
    if (C != 1 && C != 3) return fail(FVVDP_EINVAL, "The content must have either 1 or 3 colour channels.");
    if (C == 3 && !h_rgb2y) return fail(FVVDP_EINVAL, "rgb2y weights required for C == 3");
    if (fl < 1 || fl > FVVDP_MAX_TAPS) return fail(FVVDP_EINVAL, "filter length %d out of range", fl);
    if (c->P == 2 && fl != 1) return fail(FVVDP_EINVAL, "still-image context (planes == 2) needs fl == 1");
    if (n_out < 1 || slot0 < 0 || slot0 + n_out > c->max_frames) return fail(FVVDP_EINVAL, "slots [%d,%d) exceed max_frames %d", slot0, slot0 + n_out, c->max_frames);
    if (eotf->kind == FVVDP_EOTF_LUT && (dtype == FVVDP_F32 || !eotf->d_lut)) return fail(FVVDP_EINVAL, "FVVDP_EOTF_LUT needs an integer source and a table");
    if (eotf->kind != FVVDP_EOTF_LUT && dtype == FVVDP_U8) return fail(FVVDP_EINVAL, "uint8 sources need FVVDP_EOTF_LUT (uint16: table or closed form)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int HW = c->W * c->H;
    luminance_slots(c, slot0, n_out);
    luminance_range(c, eotf, C, h_rgb2y, h_taps, fl);
    Timed tm(c, 0, st);
    // register-ring kernels: up to 32 taps for every sample type, up to 64 taps (129-256 fps) for the cases of k1_ring64_ok()
    // (uint8; 16-bit / float RGB behind an sRGB or PQ display; float luminance frames); the 1-pixel-per-lane ring needs no alignment.  The 64-slot ring is not instantiated
    // for the other cases; they take the generic kernel, which re-reads the window for every output frame
    const bool ring64 = (c->P == 4) && fl > 32 && fl <= 64 && !c->env.temporal_scalar &&
                        k1_ring64_ok(dtype, C, eotf->kind);
    const bool ring_ok = ((c->P == 4) && (fl <= 32)) || ring64;
    if (ring_ok) {
        const int FL = fl <= 8 ? 8 : (fl <= 16 ? 16 : (fl <= 32 ? 32 : 64));
        const int max_out = T_MAX_IDX - (FL - 1);
        // output frames [t_begin, t_end) of this call into the level-0 buffer `level0`
        auto run = [&](int t_begin, int t_end) -> int {
            for (int t0 = t_begin; t0 < t_end; t0 += max_out) {
                const int nn = (t_end - t0) < max_out ? (t_end - t0) : max_out;
                TemporalArgs a;
                memset(&a, 0, sizeof(a));
                a.src[0] = d_test;
                a.src[1] = d_ref;
                a.chan_stride = chan_stride;
                a.frame_stride = frame_stride;
                a.C = C;
                a.HW = HW;
                a.e = make_eotf(eotf);
                if (C == 3) { a.w[0] = h_rgb2y[0]; a.w[1] = h_rgb2y[1]; a.w[2] = h_rgb2y[2]; } else { a.w[0] = 1.0f; }
                a.n_out = nn;
                a.fl = fl;
                a.out = level_addr(c, 0, slot0 + t0);
                a.oob = d_oob_flag;
                for (int k = 0; k < fl; ++k) { a.taps2[k][0] = h_taps[k]; a.taps2[k][1] = h_taps[fl + k]; }
                // virtual time of h_frame_idx: entry (fl-1+t) is the newest frame of output t; pad older history
                const int pad = FL - fl;
                for (int u = 0; u < FL - 1 + nn; ++u) {
                    const int src = t0 + u - pad;         // index into h_frame_idx
                    a.idx[u] = h_frame_idx[src < 0 ? 0 : src];
                    a.idx1[u] = h_frame_idx1 ? h_frame_idx1[src < 0 ? 0 : src] : a.idx[u];
                }
                // vector path needs the lane's PX consecutive samples to be naturally aligned
                const int PXv = k1_px(FL, dtype);        // temporal_launch.hpp
                const int es = dtype == FVVDP_U8 ? 1 : (dtype == FVVDP_U16 ? 2 : 4);
                const bool vec_ok = !c->env.temporal_scalar && (HW % PXv == 0) && (HW >= PXv) &&
                                    (chan_stride % PXv == 0) && (frame_stride % PXv == 0) &&
                                    (reinterpret_cast<uintptr_t>(d_test) % (size_t)(es * PXv) == 0) &&
                                    (reinterpret_cast<uintptr_t>(d_ref) % (size_t)(es * PXv) == 0);
                if (vec_ok) {
                    // uint8, 9..16 taps (33-64 fps): resident workgroups that take their pixel blocks from a counter
                    // (temporal_vec_kernel; FVVDP_K1_TICKET=0: one workgroup per block).  Same blocks, same arithmetic.
                    const bool tickets = FL == 16 && c->env.k1_ticket != 0;
                    if (tickets && dtype == FVVDP_U8 && c->d_ticket) {
                        if (hipMemsetAsync(c->d_ticket, 0, sizeof(int), st) == hipSuccess) a.ticket = c->d_ticket;
                        else (void)hipGetLastError();
                    }
                    k1_launch_vec(FL, dtype, a, st);
                } else if (FL == 64) {
                    // the 64-slot ring exists as the 1-pixel-per-lane vector kernel only; it needs nothing but element alignment
                    return fail(FVVDP_EINVAL, "source pointers must be aligned to their element size");
                } else {
                    k1_launch_ring(FL, dtype, a, st);
                }
            }
            return FVVDP_OK;
        };
        {
            const int rc = run(0, n_out);
            if (rc != FVVDP_OK) return rc;
        }
    } else if (c->P == 4 && fl > 32 && fl <= 64 && !h_frame_idx1 && !c->env.temporal_scalar &&
               fl - 1 + n_out <= T_MAX_IDX && k1_ring64_ok(FVVDP_F32, 1, FVVDP_EOTF_NONE)) {
        // 33..64 taps (129-256 fps) for a sample type / display model the 64-slot ring is not instantiated for: two passes.
        // Every source frame of the window -> fp32 luminance once (luminance_frames_kernel), then the 64-slot ring on those
        // frames.  The generic kernel below would evaluate the display model fl times per pixel and output frame.
        const int total = fl - 1 + n_out;
        std::vector<int> uniq(h_frame_idx, h_frame_idx + total);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        const int nu = (int)uniq.size();
        {
            const int rc = grow_lum_buf(c, (size_t)2 * nu * HW, fl, st);
            if (rc != FVVDP_OK) return rc;
        }
        LumArgs la;
        memset(&la, 0, sizeof(la));
        la.src[0] = d_test;
        la.src[1] = d_ref;
        la.chan_stride = chan_stride;
        la.frame_stride = frame_stride;
        la.C = C;
        la.HW = HW;
        la.e = make_eotf(eotf);
        if (C == 3) { la.w[0] = h_rgb2y[0]; la.w[1] = h_rgb2y[1]; la.w[2] = h_rgb2y[2]; } else { la.w[0] = 1.0f; }
        la.n_frames = nu;
        la.out = c->lum_buf;
        la.oob = d_oob_flag;
        for (int k = 0; k < nu; ++k) la.fr[k] = uniq[k];
        k1_launch_luminance(dtype, la, st);
        std::vector<int32_t> pos(total);
        for (int u = 0; u < total; ++u) pos[u] = (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), h_frame_idx[u]) - uniq.begin());
        fvvdp_eotf none;
        memset(&none, 0, sizeof(none));
        none.kind = FVVDP_EOTF_NONE;
        HIP_TRY(hipGetLastError());
        const bool timing = c->timing;
        c->timing = false;                  // the events of this call (both passes) belong to the outer timer
        c->lum_hold += 1;                   // ... and so does the luminance range (the inner pass sees plain luminance frames)
        const int rc2 = temporal_channels_core(c, c->lum_buf, c->lum_buf + (size_t)nu * HW, FVVDP_F32, 1, 0, (size_t)HW, &none, nullptr,
                                               pos.data(), nullptr, h_taps, fl, n_out, slot0, d_oob_flag, stream);
        c->timing = timing;
        c->lum_hold -= 1;
        return rc2;
    } else {
        // images, more than 64 taps and frame sizes without 4-sample alignment: one thread per pixel and output frame
        if (h_frame_idx1) return fail(FVVDP_EINVAL, "per-frame source pointers are supported for video with fl <= 32 (uint8: 64) only");
        if (fl - 1 + n_out > c->max_frames + FVVDP_MAX_TAPS) return fail(FVVDP_EINVAL, "too many frames for one call");
        GenericArgs a;
        memset(&a, 0, sizeof(a));
        if (fl <= 32 && fl - 1 + n_out <= T_MAX_IDX) {
            // still images and short clips: the tables ride in the kernel arguments
            a.inline_tables = 1;
            for (int k = 0; k < 2 * fl; ++k) a.taps_i[k] = h_taps[k];
            for (int k = 0; k < fl - 1 + n_out; ++k) a.idx_i[k] = h_frame_idx[k];
        } else {
            HIP_TRY(hipStreamSynchronize(st));
            c->n_sync += 1;                                      // (more than 32 taps without a ring kernel: tables through device memory)
            HIP_TRY(hipMemcpy(c->d_taps, h_taps, sizeof(float) * 2 * fl, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(c->d_idx, h_frame_idx, sizeof(int) * (fl - 1 + n_out), hipMemcpyHostToDevice));
        }
        a.src[0] = d_test;
        a.src[1] = d_ref;
        a.chan_stride = chan_stride;
        a.frame_stride = frame_stride;
        a.C = C;
        a.HW = HW;
        a.e = make_eotf(eotf);
        if (C == 3) { a.w[0] = h_rgb2y[0]; a.w[1] = h_rgb2y[1]; a.w[2] = h_rgb2y[2]; } else { a.w[0] = 1.0f; }
        a.n_out = n_out;
        a.fl = fl;
        a.out = level_addr(c, 0, slot0);
        a.oob = d_oob_flag;
        a.taps = c->d_taps;
        a.idx = c->d_idx;
        k1_launch_generic(c->P, dtype, a, st);
    }
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

extern "C" int fvvdp_temporal_channels(fvvdp_ctx* c, const void* d_test, const void* d_ref, int dtype, int C,
                                       size_t chan_stride, size_t frame_stride, const fvvdp_eotf* eotf,
                                       const float* h_rgb2y, const int32_t* h_frame_idx, const float* h_taps, int fl,
                                       int n_out, int slot0, int32_t* d_oob_flag, void* stream) {
    return temporal_channels_core(c, d_test, d_ref, dtype, C, chan_stride, frame_stride, eotf, h_rgb2y, h_frame_idx, nullptr,
                                  h_taps, fl, n_out, slot0, d_oob_flag, stream);
}

extern "C" int fvvdp_temporal_channels_frames(fvvdp_ctx* c, const void* const* h_test_frames, const void* const* h_ref_frames,
                                              int n_frames, int dtype, int C, size_t chan_stride, const fvvdp_eotf* eotf,
                                              const float* h_rgb2y, const int32_t* h_frame_idx, const float* h_taps, int fl,
                                              int n_out, int slot0, int32_t* d_oob_flag, void* stream) {
    if (!c || !h_test_frames || !h_ref_frames || !h_frame_idx) return fail(FVVDP_EINVAL, "null argument");
    if (n_frames < 1 || n_frames > 65536) return fail(FVVDP_EINVAL, "n_frames out of range");
    if (dtype < FVVDP_U8 || dtype > FVVDP_F32) return fail(FVVDP_EINVAL, "Only uint8, uint16 and float32 is currently supported");
    if (fl < 1 || fl > ((eotf && k1_ring64_ok(dtype, C, eotf->kind)) ? 64 : 32) || c->P != 4)
        return fail(FVVDP_EUNSUPPORTED, "per-frame source pointers: video contexts with fl <= 32 (uint8, float luminance: 64) only");
    const size_t es = dtype == FVVDP_U8 ? 1 : (dtype == FVVDP_U16 ? 2 : 4);
    // every frame is addressed as base + index * frame_stride with ONE base per stream: base = lowest frame address,
    // frame_stride = the coarsest granule (4 elements, else 1) that divides every distance; indices are 31-bit
    const void* const* fr[2] = {h_test_frames, h_ref_frames};
    uintptr_t base[2];
    size_t gran = 4;
    for (int s = 0; s < 2; ++s) {
        base[s] = UINTPTR_MAX;
        for (int i = 0; i < n_frames; ++i) {
            if (!fr[s][i]) return fail(FVVDP_EINVAL, "null frame pointer");
            const uintptr_t p = reinterpret_cast<uintptr_t>(fr[s][i]);
            if (p % es) return fail(FVVDP_EINVAL, "frame pointer not aligned to its element size");
            if (p < base[s]) base[s] = p;
        }
        for (int i = 0; i < n_frames; ++i)
            if ((reinterpret_cast<uintptr_t>(fr[s][i]) - base[s]) % (es * 4)) gran = 1;
    }
    std::vector<int32_t> off[2];
    for (int s = 0; s < 2; ++s) {
        off[s].resize(n_frames);
        for (int i = 0; i < n_frames; ++i) {
            const uintptr_t d = (reinterpret_cast<uintptr_t>(fr[s][i]) - base[s]) / (es * gran);
            if (d > 0x7FFFFFFFu) return fail(FVVDP_EUNSUPPORTED, "frames are too far apart in memory for 31-bit frame indices");
            off[s][i] = (int32_t)d;
        }
    }
    const int total = fl - 1 + n_out;
    std::vector<int32_t> i0(total), i1(total);
    for (int u = 0; u < total; ++u) {
        const int f = h_frame_idx[u];
        if (f < 0 || f >= n_frames) return fail(FVVDP_EINVAL, "frame index %d out of range", f);
        i0[u] = off[0][f];
        i1[u] = off[1][f];
    }
    return temporal_channels_core(c, reinterpret_cast<const void*>(base[0]), reinterpret_cast<const void*>(base[1]), dtype, C,
                                  chan_stride, gran, eotf, h_rgb2y, i0.data(), i1.data(), h_taps, fl, n_out, slot0, d_oob_flag,
                                  stream);
}

extern "C" int fvvdp_temporal_channels_yuv(fvvdp_ctx* c, const void* d_test, const void* d_ref, const fvvdp_yuv_format* fmt,
                                           size_t frame_stride, const fvvdp_eotf* eotf, const float* h_rgb2y,
                                           const int32_t* h_frame_idx, const float* h_taps, int fl, int n_out, int slot0,
                                           int32_t* d_oob_flag, void* stream) {
    if (!c || !d_test || !d_ref || !fmt || !eotf || !h_rgb2y || !h_frame_idx || !h_taps) return fail(FVVDP_EINVAL, "null argument");
    if (c->P != 4) return fail(FVVDP_EINVAL, "YUV ingest is for video contexts (planes == 4)");
    if (fmt->bit_depth < 8 || fmt->bit_depth > 16) return fail(FVVDP_EINVAL, "bit depth %d not supported", fmt->bit_depth);
    if (fmt->chroma_420 && ((c->W | c->H) & 1)) return fail(FVVDP_EINVAL, "4:2:0 needs even frame dimensions");
    if (fl < 1 || fl > 64) return fail(FVVDP_EINVAL, "filter length %d out of range for the YUV path (1..64)", fl);
    if (n_out < 1 || slot0 < 0 || slot0 + n_out > c->max_frames) return fail(FVVDP_EINVAL, "slots out of range");
    if (eotf->kind == FVVDP_EOTF_LUT) return fail(FVVDP_EINVAL, "YUV sources need a closed-form display model (RGB is fractional after the matrix)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    luminance_slots(c, slot0, n_out);
    luminance_range(c, eotf, 3, h_rgb2y, h_taps, fl);        // (RGB is clamped to [0,1] before the display model)
    Timed tm(c, 0, st);
    if (fl > 32) {
        // 33..64 taps (129-256 fps): every source frame of the window -> fp32 luminance once, then the 64-slot ring on the
        // luminance frames (see temporal_channels_core)
        const int total = fl - 1 + n_out;
        if (total > T_MAX_IDX) return fail(FVVDP_EINVAL, "too many frames for one call");
        const int HW = c->W * c->H;
        std::vector<int> uniq(h_frame_idx, h_frame_idx + total);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        const int nu = (int)uniq.size();
        {
            const int rc = grow_lum_buf(c, (size_t)2 * nu * HW, fl, st);
            if (rc != FVVDP_OK) return rc;
        }
        YuvLumArgs la;
        memset(&la, 0, sizeof(la));
        YuvArgs& a = la.y;
        a.src[0] = d_test;
        a.src[1] = d_ref;
        a.frame_stride = frame_stride;
        a.W = c->W;
        a.H = c->H;
        a.chroma420 = fmt->chroma_420 ? 1 : 0;
        a.uvw = fmt->chroma_420 ? c->W / 2 : c->W;
        a.uvh = fmt->chroma_420 ? c->H / 2 : c->H;
        const float scale = (float)(1 << (fmt->bit_depth - 8));
        a.wy = 1.0f / (scale * 219.0f);
        a.wc = 1.0f / (scale * 224.0f);
        for (int i = 0; i < 9; ++i) a.m[i] = fmt->ycbcr2rgb[i];
        a.e = make_eotf(eotf);
        a.w[0] = h_rgb2y[0]; a.w[1] = h_rgb2y[1]; a.w[2] = h_rgb2y[2];
        a.oob = d_oob_flag;
        la.n_frames = nu;
        la.out = c->lum_buf;
        for (int k = 0; k < nu; ++k) la.fr[k] = uniq[k];
        k1_launch_yuv_luminance(fmt->bit_depth > 8 ? 2 : 1, la, st);
        HIP_TRY(hipGetLastError());
        std::vector<int32_t> pos(total);
        for (int u = 0; u < total; ++u) pos[u] = (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), h_frame_idx[u]) - uniq.begin());
        fvvdp_eotf none;
        memset(&none, 0, sizeof(none));
        none.kind = FVVDP_EOTF_NONE;
        const bool timing = c->timing;
        c->timing = false;
        c->lum_hold += 1;
        const int rc2 = temporal_channels_core(c, c->lum_buf, c->lum_buf + (size_t)nu * HW, FVVDP_F32, 1, 0, (size_t)HW, &none, nullptr,
                                               pos.data(), nullptr, h_taps, fl, n_out, slot0, d_oob_flag, stream);
        c->timing = timing;
        c->lum_hold -= 1;
        return rc2;
    }
    const int FL = fl <= 8 ? 8 : (fl <= 16 ? 16 : 32);
    const int max_out = T_MAX_IDX - (FL - 1);
    for (int t0 = 0; t0 < n_out; t0 += max_out) {
        const int nn = (n_out - t0) < max_out ? (n_out - t0) : max_out;
        YuvArgs a;
        memset(&a, 0, sizeof(a));
        a.src[0] = d_test;
        a.src[1] = d_ref;
        a.frame_stride = frame_stride;
        a.W = c->W;
        a.H = c->H;
        a.chroma420 = fmt->chroma_420 ? 1 : 0;
        a.uvw = fmt->chroma_420 ? c->W / 2 : c->W;
        a.uvh = fmt->chroma_420 ? c->H / 2 : c->H;
        const float scale = (float)(1 << (fmt->bit_depth - 8));
        a.wy = 1.0f / (scale * 219.0f);
        a.wc = 1.0f / (scale * 224.0f);
        for (int i = 0; i < 9; ++i) a.m[i] = fmt->ycbcr2rgb[i];
        a.e = make_eotf(eotf);
        a.w[0] = h_rgb2y[0]; a.w[1] = h_rgb2y[1]; a.w[2] = h_rgb2y[2];
        a.n_out = nn;
        a.fl = fl;
        a.out = level_addr(c, 0, slot0 + t0);
        a.oob = d_oob_flag;
        for (int k = 0; k < fl; ++k) { a.taps2[k][0] = h_taps[k]; a.taps2[k][1] = h_taps[fl + k]; }
        const int pad = FL - fl;
        for (int u = 0; u < FL - 1 + nn; ++u) {
            const int src = t0 + u - pad;
            a.idx[u] = h_frame_idx[src < 0 ? 0 : src];
        }
        const int bytes = fmt->bit_depth > 8 ? 2 : 1;
        // vector kernel: 4 consecutive pixels per lane -> rows, planes and frames must keep 4-sample alignment
        const size_t al = (size_t)bytes * 4;
        const bool vec_ok = !c->env.temporal_scalar && FL <= 16 && (c->W % 4 == 0) && (frame_stride % 4 == 0) &&
                            (reinterpret_cast<uintptr_t>(d_test) % al == 0) && (reinterpret_cast<uintptr_t>(d_ref) % al == 0);
        if (vec_ok) {
            k1_launch_yuv_vec(FL, bytes, a.chroma420 != 0, c->env.yuv_general, a, st);
        } else {
            k1_launch_yuv(FL, bytes, a, st);
        }
    }
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

extern "C" int fvvdp_load_channels_planar(fvvdp_ctx* c, const float* d_R, int n, int slot0, void* stream) {
    if (!c || !d_R) return fail(FVVDP_EINVAL, "null argument");
    if (n < 1 || slot0 < 0 || slot0 + n > c->max_frames) return fail(FVVDP_EINVAL, "slots out of range");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int HW = c->W * c->H;
    dim3 grid((HW + 255) / 256, n), block(256);
    const L0Addr out = level_addr(c, 0, slot0);
    luminance_slots(c, slot0, n);
    luminance_unknown(c);                            // the caller's own temporal channels
    if (c->P == 4) hipLaunchKernelGGL((interleave_kernel<4>), grid, block, 0, st, const_cast<float*>(d_R), out, HW, 1);
    else hipLaunchKernelGGL((interleave_kernel<2>), grid, block, 0, st, const_cast<float*>(d_R), out, HW, 1);
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

template <int PX>
static void launch_pu21(int dtype, const Pu21Args& a, int n_frames, hipStream_t st) {
    dim3 grid(FVVDP_PSNR_SLICES, n_frames), block(256);
    if (dtype == FVVDP_U8) hipLaunchKernelGGL((pu21_sse_kernel<SRC_U8, PX>), grid, block, 0, st, a);
    else if (dtype == FVVDP_U16) hipLaunchKernelGGL((pu21_sse_kernel<SRC_U16, PX>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((pu21_sse_kernel<SRC_F32, PX>), grid, block, 0, st, a);
}

extern "C" int fvvdp_yuv_frame_resized(const void* d_frame, const fvvdp_yuv_format* fmt, int W, int H, float* d_rgb_scratch,
                                       int out_w, int out_h, int mode, const fvvdp_eotf* eotf, const float* h_rgb2y,
                                       float* d_lum, float* d_rgb_out, void* stream) {
    if (!d_frame || !fmt || !d_rgb_scratch || !eotf || !h_rgb2y || !d_lum) return fail(FVVDP_EINVAL, "null argument");
    if (W < 1 || H < 1 || out_w < 1 || out_h < 1 || W > 32768 || H > 32768 || out_w > 32768 || out_h > 32768)
        return fail(FVVDP_EINVAL, "frame size out of range (1..32768 per axis)");        // index products stay inside 31 bits
    if (fmt->bit_depth < 8 || fmt->bit_depth > 16) return fail(FVVDP_EINVAL, "bit depth %d not supported", fmt->bit_depth);
    if (fmt->chroma_420 && ((W | H) & 1)) return fail(FVVDP_EINVAL, "4:2:0 needs even frame dimensions");
    if (mode < FVVDP_RESIZE_NEAREST || mode > FVVDP_RESIZE_AREA) return fail(FVVDP_EINVAL, "unknown resize mode %d", mode);
    if (eotf->kind == FVVDP_EOTF_LUT) return fail(FVVDP_EINVAL, "YUV sources need a closed-form display model (RGB is fractional after the matrix)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    YuvRgbArgs y;
    memset(&y, 0, sizeof(y));
    y.src = d_frame;
    y.W = W; y.H = H;
    y.chroma420 = fmt->chroma_420 ? 1 : 0;
    y.uvw = fmt->chroma_420 ? W / 2 : W;
    y.uvh = fmt->chroma_420 ? H / 2 : H;
    const float scale = (float)(1 << (fmt->bit_depth - 8));
    y.wy = 1.0f / (scale * 219.0f);
    y.wc = 1.0f / (scale * 224.0f);
    for (int i = 0; i < 9; ++i) y.m[i] = fmt->ycbcr2rgb[i];
    y.out = d_rgb_scratch;
    const dim3 block(256), g1((unsigned)(((long long)W * H + 255) / 256));
    if (fmt->bit_depth > 8) hipLaunchKernelGGL((yuv_rgb_planar_kernel<unsigned short>), g1, block, 0, st, y);
    else hipLaunchKernelGGL((yuv_rgb_planar_kernel<unsigned char>), g1, block, 0, st, y);
    ResizeArgs r;
    memset(&r, 0, sizeof(r));
    r.rgb = d_rgb_scratch;
    r.W = W; r.H = H; r.Wo = out_w; r.Ho = out_h;
    r.sx = (float)W / (float)out_w;
    r.sy = (float)H / (float)out_h;
    r.e = make_eotf(eotf);
    r.w[0] = h_rgb2y[0]; r.w[1] = h_rgb2y[1]; r.w[2] = h_rgb2y[2];
    r.lum = d_lum;
    r.rgb_out = d_rgb_out;
    const dim3 g2((unsigned)(((long long)out_w * out_h + 255) / 256));
    switch (mode) {
        case FVVDP_RESIZE_NEAREST: hipLaunchKernelGGL((resize_lum_kernel<FVVDP_RESIZE_NEAREST>), g2, block, 0, st, r); break;
        case FVVDP_RESIZE_BILINEAR: hipLaunchKernelGGL((resize_lum_kernel<FVVDP_RESIZE_BILINEAR>), g2, block, 0, st, r); break;
        case FVVDP_RESIZE_BICUBIC: hipLaunchKernelGGL((resize_lum_kernel<FVVDP_RESIZE_BICUBIC>), g2, block, 0, st, r); break;
        default: hipLaunchKernelGGL((resize_lum_kernel<FVVDP_RESIZE_AREA>), g2, block, 0, st, r); break;
    }
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

extern "C" int fvvdp_pu21_sse(const void* d_test, const void* d_ref, int dtype, int C, size_t chan_stride,
                              size_t frame_stride, size_t n_pixels, const fvvdp_eotf* eotf, const float* h_rgb2y,
                              const fvvdp_pu21* pu, int n_frames, double* d_partial, double* d_sse, int32_t* d_oob_flag,
                              void* stream) {
    if (!d_test || !d_ref || !eotf || !pu || !d_partial || !d_sse) return fail(FVVDP_EINVAL, "null argument");
    if (dtype < FVVDP_U8 || dtype > FVVDP_F32) return fail(FVVDP_EINVAL, "Only uint8, uint16 and float32 is currently supported");
    if (C != 1 && C != 3) return fail(FVVDP_EINVAL, "The content must have either 1 or 3 colour channels.");
    if (C == 3 && !h_rgb2y) return fail(FVVDP_EINVAL, "rgb2y weights required for C == 3");
    if (n_frames < 1 || n_frames > 65535) return fail(FVVDP_EINVAL, "n_frames %d out of range (1..65535 per call)", n_frames);
    if (n_pixels < 1 || n_pixels > 0x7FFFFFFFu) return fail(FVVDP_EINVAL, "n_pixels out of range");
    if (eotf->kind == FVVDP_EOTF_LUT && (dtype == FVVDP_F32 || !eotf->d_lut)) return fail(FVVDP_EINVAL, "FVVDP_EOTF_LUT needs an integer source and a table");
    if (eotf->kind != FVVDP_EOTF_LUT && dtype == FVVDP_U8) return fail(FVVDP_EINVAL, "uint8 sources need FVVDP_EOTF_LUT (uint16: table or closed form)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    Pu21Args a;
    memset(&a, 0, sizeof(a));
    a.src[0] = d_test;
    a.src[1] = d_ref;
    a.chan_stride = chan_stride;
    a.frame_stride = frame_stride;
    a.C = C;
    a.HW = (unsigned int)n_pixels;
    const unsigned int per = (unsigned int)((n_pixels + FVVDP_PSNR_SLICES - 1) / FVVDP_PSNR_SLICES);
    a.chunk = (per + 3u) & ~3u;
    a.e = make_eotf(eotf);
    if (C == 3) { a.w[0] = h_rgb2y[0]; a.w[1] = h_rgb2y[1]; a.w[2] = h_rgb2y[2]; } else { a.w[0] = 1.0f; }
    for (int i = 0; i < 7; ++i) a.p[i] = pu->p[i];
    a.l_min = pu->L_min;
    a.l_max = pu->L_max;
    a.partial = d_partial;
    a.oob = d_oob_flag;
    // 4 consecutive samples per load where every plane and frame keeps 4-sample alignment
    const int es = dtype == FVVDP_U8 ? 1 : (dtype == FVVDP_U16 ? 2 : 4);
    const bool vec_ok = (n_pixels % 4 == 0) && (chan_stride % 4 == 0) && (frame_stride % 4 == 0) &&
                        (reinterpret_cast<uintptr_t>(d_test) % (size_t)(es * 4) == 0) &&
                        (reinterpret_cast<uintptr_t>(d_ref) % (size_t)(es * 4) == 0);
    if (vec_ok) launch_pu21<4>(dtype, a, n_frames, st);
    else launch_pu21<1>(dtype, a, n_frames, st);
    hipLaunchKernelGGL(pu21_finalize_kernel, dim3(n_frames), dim3(64), 0, st, d_partial, d_sse);
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

extern "C" int fvvdp_export_level(fvvdp_ctx* c, int level, int n, float* d_out, void* stream) {
    if (!c || !d_out) return fail(FVVDP_EINVAL, "null argument");
    if (level < 0 || level > c->n_bands || n < 1 || n > c->max_frames) return fail(FVVDP_EINVAL, "bad level/n");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int HW = c->lw[level] * c->lh[level];
    dim3 grid((HW + 255) / 256, n), block(256);
    if (c->P == 4) hipLaunchKernelGGL((interleave_kernel<4>), grid, block, 0, st, d_out, level_addr(c, level, 0), HW, 0);
    else hipLaunchKernelGGL((interleave_kernel<2>), grid, block, 0, st, d_out, level_addr(c, level, 0), HW, 0);
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// stage 2 launcher
// ------------------------------------------------------------------------------------------------------------
// Foveated mode: slice of the 32^3 LUT that band b can reach.  rho = rho_band*res_mag with res_mag in
// [1, res_mag(corner of the screen)], so only a few rho knots are live per band; the slice is stored as
// [ecc][Y][rho] of float4 {S0[i], S1[i], S0[i+1], S1[i+1]} so that one aligned 16-byte load returns the two rho
// corners of both temporal channels (value at knot i and the step to knot i+1: v[i] + f*(v[i+1]-v[i])) and neighbouring
// pixels (similar ecc, Y) share cache lines.
static int build_sublut(fvvdp_ctx* c, const fvvdp_geom* g, hipStream_t st) {
    fvvdp_geom key;
    memset(&key, 0, sizeof(key));
    if (g) key = *g;
    if (c->sub_valid && memcmp(&c->sub_geom, &key, sizeof(key)) == 0) return FVVDP_OK;
    HIP_TRY(hipStreamSynchronize(st));            // geometry changed: kernels of earlier calls may still read the old slices
    c->n_sync += 1;
    double mag_max = 1.0, mag_min = 1.0;
    if (g) {
        const double delta = (1.0 / (double)g->ppd_centre) / 2.0 * M_PI / 180.0;
        const double ax = atan(0.5 * g->display_size_m[0] / g->distance_m) * 180.0 / M_PI;
        const double ay = atan(0.5 * g->display_size_m[1] / g->distance_m) * 180.0 / M_PI;
        double va = sqrt(ax * ax + ay * ay);
        if (va > 89.9) va = 89.9;
        va *= M_PI / 180.0;
        mag_max = cos(delta) / (cos(va) * cos(va + delta)) * 1.01;   // 1 % margin over the corner pixel
    }
    const float* xr = c->h_axes[1];
    for (int b = 0; b < c->n_bands; ++b) {
        if (!g) {                // user geometry: the caller states the range of its magnification map
            mag_max = c->map_rm_max[b] > 0 ? (double)c->map_rm_max[b] * 1.01 : 1e9;
            mag_min = c->map_rm_min[b] > 0 ? (double)c->map_rm_min[b] * 0.99 : 0.0;
        }
        double r_lo = c->rho_band[b] * mag_min, r_hi = c->rho_band[b] * mag_max;
        const double lo_c = exp2((double)xr[0]), hi_c = exp2((double)xr[FVVDP_LUT_N - 1]);
        r_lo = fmin(fmax(r_lo, lo_c), hi_c);
        r_hi = fmin(fmax(r_hi, lo_c), hi_c);
        int i_lo = 0, i_hi = FVVDP_LUT_N - 1;
        while (i_lo + 1 < FVVDP_LUT_N - 1 && xr[i_lo + 1] < log2(r_lo) - 1e-3) ++i_lo;   // last knot <= log2(r_lo)
        while (i_hi - 1 > i_lo && xr[i_hi - 1] > log2(r_hi) + 1e-3) --i_hi;              // first knot >= log2(r_hi)
        const int rw = i_hi - i_lo;                                                       // intervals covered
        if (c->sublut[b] && c->sub_rw[b] != rw) {
            (void)hipFree(c->sublut[b]);
            c->n_free += 1;
            c->sublut[b] = nullptr;
        }
        const size_t n = (size_t)FOV_PLANE * rw;                 // padded layout, see FOV_ROW / FOV_PLANE (band_kernel.hpp)
        if (!c->sublut[b]) {
            int rc = dev_alloc(c, &c->sublut[b], n);
            if (rc != FVVDP_OK) return rc;
        }
        std::vector<float4> h(n, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        const float* L0 = c->h_lut3[0].data();
        const float* L1 = c->h_lut3[1].empty() ? L0 : c->h_lut3[1].data();
        for (int k = 0; k < FVVDP_LUT_N; ++k)
            for (int j = 0; j < FVVDP_LUT_N; ++j)
                for (int i = 0; i < rw; ++i) {
                    const size_t s0 = ((size_t)j * FVVDP_LUT_N + (i_lo + i)) * FVVDP_LUT_N + k;       // [Y][rho][ecc]
                    const size_t s1 = ((size_t)j * FVVDP_LUT_N + (i_lo + i + 1)) * FVVDP_LUT_N + k;
                    // slice layout [rho interval i][ecc k][Y j]; rho blend in slope form {v[i], v[i+1] - v[i]}
                    h[(size_t)i * FOV_PLANE + (size_t)k * FOV_ROW + j] = make_float4(L0[s0], L1[s0], L0[s1] - L0[s0], L1[s1] - L1[s0]);
                }
        HIP_TRY(hipMemcpy(c->sublut[b], h.data(), n * sizeof(float4), hipMemcpyHostToDevice));
        c->sub_rw[b] = rw;
        c->sub_ilo[b] = i_lo;
        if (g && !c->env.fov_no_rhomap) {   // frame-invariant rho-axis coordinates of every pixel (stock geometry)
            const int pw = (c->lw[b] + 1) / 2;
            if (!c->rmap[b]) {
                int rc = dev_alloc(c, &c->rmap[b], (size_t)pw * c->lh[b]);
                if (rc != FVVDP_OK) return rc;
            }
            RhoMapArgs ra;
            memset(&ra, 0, sizeof(ra));
            ra.out = c->rmap[b];
            ra.w = c->lw[b];
            ra.h = c->lh[b];
            ra.size_m0 = g->display_size_m[0];
            ra.size_m1 = g->display_size_m[1];
            ra.dist_m = g->distance_m;
            const double delta = (1.0 / (double)g->ppd_centre) / 2.0 * M_PI / 180.0;
            ra.delta_rad = (float)delta;
            ra.cos_delta = (float)cos(delta);
            ra.rho_band = (float)c->rho_band[b];
            ra.rho_lo = c->rho_lo;
            ra.rho_hi = c->rho_hi;
            ra.first = c->h_axes[1][0];
            ra.inv_step = (float)(FVVDP_LUT_N - 1) / (c->h_axes[1][FVVDP_LUT_N - 1] - c->h_axes[1][0]);
            ra.i_lo = i_lo;
            ra.rw = rw;
            ra.axis = c->d_axes + FVVDP_LUT_N;
            ra.plane_bytes = FOV_PLANE * 16;
            hipLaunchKernelGGL(fov_rho_map_kernel, dim3((pw + 255) / 256, c->lh[b]), dim3(256), 0, st, ra);
        }
    }
    c->sub_geom = key;
    c->sub_valid = true;
    return FVVDP_OK;
}

template <int P>
static void launch_band(const BandArgs& a, int nblocks, bool dbg, bool fov, hipStream_t st) {
    dim3 grid(nblocks), block(64);
    if (fov) {
        const dim3 gridf((nblocks + FOV_WPB - 1) / FOV_WPB), blockf(64 * FOV_WPB);
        // dynamic LDS: [LUT slice of the band (mode 1)] + vertical view angle of every band row
        const size_t lds_vy = (size_t)a.h * sizeof(float);
        const size_t lds_lut = (size_t)FOV_PLANE * a.rw * sizeof(float4);
        if (dbg) hipLaunchKernelGGL((band_kernel<P, true, 2>), gridf, blockf, lds_vy, st, a);
        else if (a.lut_lds && a.rmap && !a.mvx) hipLaunchKernelGGL((band_kernel<P, false, 1>), gridf, blockf, lds_lut + lds_vy, st, a);
        else if (a.lut_lds) hipLaunchKernelGGL((band_kernel<P, false, 3>), gridf, blockf, lds_lut + lds_vy, st, a);
        else hipLaunchKernelGGL((band_kernel<P, false, 2>), gridf, blockf, lds_vy, st, a);
    } else {
        if (dbg) hipLaunchKernelGGL((band_kernel<P, true, 0>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((band_kernel<P, false, 0>), grid, block, 0, st, a);
    }
}

static void fill_pool_args(PoolArgs& a, const float* d_Q, int n_bands, int n_channels, int n_frames, int q_stride,
                           const fvvdp_pool_params* prm, float* d_jod) {
    memset(&a, 0, sizeof(a));
    a.Q = d_Q;
    a.n_bands = n_bands;
    a.n_ch = n_channels;
    a.n_frames = n_frames;
    a.q_stride = q_stride;
    a.beta_sch = prm->beta_sch;
    a.beta_tch = prm->beta_tch;
    a.beta_t = prm->beta_t;
    a.w_transient = prm->w_transient;
    a.jod_a = prm->jod_a;
    a.beta_jod = prm->beta_jod;
    a.out = d_jod;
}

static int check_pool_params(const fvvdp_pool_params* prm) {
    if (!(prm->beta_sch > 0.0f) || !(prm->beta_tch > 0.0f) || !(prm->beta_t > 0.0f) || !(prm->beta_jod > 0.0f))
        return fail(FVVDP_EINVAL, "pooling exponents must be positive");
    return FVVDP_OK;
}

// slot0: first level-0 frame slot of the batch (fvvdp_temporal_channels wrote slots [slot0, slot0 + n)); the levels below are
// scratch of the pass itself and always use slots [0, n)
static int bands_forward_core(fvvdp_ctx* c, int slot0, int n, float* d_Q, int q_stride, int q_col0, const float* h_fixation,
                              const fvvdp_geom* geom, const fvvdp_band_maps* maps, const fvvdp_pool_params* pool,
                              float* d_jod, void* stream) {
    if (!c || !d_Q) return fail(FVVDP_EINVAL, "null argument");
    if (slot0 < 0 || n < 1 || slot0 + n > c->max_frames) return fail(FVVDP_EINVAL, "slots [%d,%d) exceed max_frames %d", slot0, slot0 + n, c->max_frames);
    if (pool) {
        if (!d_jod) return fail(FVVDP_EINVAL, "null argument");
        int rc = check_pool_params(pool);
        if (rc != FVVDP_OK) return rc;
    }
    const bool last_batch = (q_col0 + n == q_stride);            // this call completes the clip
    const bool pool_now = pool && last_batch;
    if (n < 1 || n > c->max_frames) return fail(FVVDP_EINVAL, "n=%d exceeds max_frames=%d", n, c->max_frames);
    if (q_col0 < 0 || q_col0 + n > q_stride) return fail(FVVDP_EINVAL, "Q columns out of range");
    const bool fov = h_fixation != nullptr;
    if (fov && !geom && !c->maps_set) return fail(FVVDP_EINVAL, "foveated mode needs the display geometry (or fvvdp_ctx_set_view_maps)");
    if (fov && !(c->lut3_set[0] && (c->P == 2 || c->lut3_set[1]))) return fail(FVVDP_ESTATE, "fvvdp_ctx_set_csf_3d not called");
    if (!fov && !c->csf_set) return fail(FVVDP_ESTATE, "fvvdp_ctx_set_csf_1d not called");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (fov) {
        // stream-ordered after the kernels of the previous call that still read d_fix; the source is pageable host
        // memory, which the runtime stages before the call returns
        HIP_TRY(hipMemcpyAsync(c->d_fix, h_fixation, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
        int rc = build_sublut(c, geom, st);
        if (rc != FVVDP_OK) return rc;
    }
    FinalizeArgs fa;
    memset(&fa, 0, sizeof(fa));
    // Two pyramid levels per pass where possible (band2_kernel): plain evaluation only -- the map-writing and the
    // foveated variants, and levels too small for the two-level border logic, take one level per launch.
    bool any_maps = false;
    if (maps)
        for (int b = 0; b < c->n_bands; ++b) any_maps = any_maps || maps[b].d_D || maps[b].d_contrast || maps[b].d_lbkg || maps[b].d_S;
    // It pays where the one-level kernel is bound by HBM (large levels: 4K levels 0+1 39 vs 46 us per frame); the
    // two-level kernel's arithmetic takes as long as its data flow (its strips own 54 of 64 lanes), so small levels stay on the one-level kernel
    // (960x540 + 480x270: 3.1 vs 2.95 us).  FVVDP_BAND_FUSE=0 / 1 forces never / wherever valid (tests, A/B runs).
    const int fuse_mode = c->env.fuse_mode;                      // 0 / 1, anything else: automatic
    const bool fuse_ok = !fov && !any_maps && fuse_mode != 0;
    for (int b = 0; b < c->n_bands; ++b) {
        const bool big = (long long)c->lw[b] * c->lh[b] >= 500000;      // (4K: levels 0+1 and 2+3; 2+3 in one launch: 3.06 -> 2.67 us per frame)
        if (fuse_ok && (big || fuse_mode == 1) && b + 1 < c->n_bands && c->lw[b + 1] >= 4 && c->lh[b + 1] >= 4 &&
            c->lw[b + 2] >= 2 && c->lh[b + 2] >= 2) {
            Band2Args a;
            memset(&a, 0, sizeof(a));
            a.A = level_addr(c, b, b == 0 ? slot0 : 0);
            a.Gc = c->level[b + 2];
            a.w = c->lw[b];
            a.h = c->lh[b];
            a.wb = c->lw[b + 1];
            a.hb = c->lh[b + 1];
            a.wc = c->lw[b + 2];
            a.hc = c->lh[b + 2];
            a.n_strips = band2_strips(a.wb);
            chunking2(a.hc, a.n_strips, n, c->wave_capacity2, c->env.band2_kr, a.n_chunks, a.kr);
            a.n_big = a.n_chunks;
            a.kr2 = a.kr;
            a.n_frames = n;
            {
                // a launch of several rounds of tall chunks: the last chunk of every frame is cut into four and dispatched after
                // all tall ones (band2_kernel's two phases)
                const long long waves = (long long)n * a.n_strips * a.n_chunks;
                int kr2 = (waves >= 2 * c->wave_capacity2 && a.n_chunks >= 3 && a.kr >= 16) ? (a.kr + 3) / 4 : 0;
                if (c->env.band2_kr2 >= 0) kr2 = c->env.band2_kr2;                  // tuning override (FVVDP_BAND2_KR2); 0 = uniform chunks
                if (kr2 >= 1 && kr2 < a.kr && a.n_chunks >= 2) {
                    a.n_big = a.n_chunks - 1;
                    a.kr2 = kr2;
                    a.n_chunks = a.n_big + (a.hc - a.n_big * a.kr + kr2 - 1) / kr2;
                }
            }
            a.mulA = (b == 0) ? 1.0f : 2.0f;
            a.mulB = 2.0f;
            a.csfA = c->csf + (size_t)b * FVVDP_LUT_N;
            a.csfB = c->csf + (size_t)(b + 1) * FVVDP_LUT_N;
            a.y_first = c->y_first;
            a.y_inv_step = c->y_inv_step;
            a.ly_lo = log2f(c->y_lo);
            a.ly_hi = log2f(c->y_hi);
            a.lg_gain = log2f(c->prm.sens_gain);
            a.lg_k = log2f(c->prm.mask_k);
            a.p = c->prm.mask_p;
            a.q0 = c->prm.mask_q[0];
            a.q1 = c->prm.mask_q[1];
            a.beta = c->prm.beta;
            a.lbkg_min = c->prm.lbkg_min;
            a.cmax = c->prm.contrast_max;
            a.lg_dmax = log2f(c->prm.d_max);
            a.partialA = c->partial + c->partial_off[b];
            a.partialB = c->partial + c->partial_off[b + 1];
            const int nblk = a.n_strips * a.n_chunks;
            if (nblk > c->max_blk[b] || nblk > c->max_blk[b + 1]) return fail(FVVDP_ESTATE, "internal: partial buffer too small");
            {
                Timed tm(c, 1 + b, st);
                const bool inrange = clamps_never_bind(c, b, 2);
                // Waves per workgroup: 4 adjacent strips walked in step (one barrier per stage).  Free-running single waves drift
                // apart over a long launch (items started in the first round of a 4K x 60 launch, all in step, take 316 us, later
                // ones 390 us); neighbours kept in step meet in the CU's vector cache on their shared halo columns and ask for one
                // contiguous piece of each row.  Measured (profiles/r04_lockstep.md): -1.5 ... -6 % at 4K on five boxes, -2.6 % at
                // 8K, 0 ... -3 % at 2560x1440; slower where the strips do not fill whole groups (a workgroup holds its four wave
                // slots) and below 2560 columns (short launches stay in step by themselves).  Same work items, same partial sums.
                int wpb = (a.w >= 2560 && a.n_strips % BAND2_WPB_MAX == 0) ? BAND2_WPB_MAX : 1;
                { const int v = c->env.band2_wpb; if (v >= 1 && v <= BAND2_WPB_MAX && a.n_strips % v == 0) wpb = v; }      // FVVDP_BAND2_WPB
                // work items handed out per XCD at run time (band2_kernel, `tickets`): launches of several rounds only;
                // FVVDP_BAND2_TICKET=0 / 1 forces the static split / the counters
                const int n_items = (a.n_strips / wpb) * a.n_chunks * n;
                const long long waves2 = (long long)n * a.n_strips * a.n_chunks;
                // Measured (profiles/r05_band2_tickets.md): 4K levels 0+1 (6000 items of ~300 us, 4 waves each) -0.4 ... -1.9 %; launches of
                // many short items lose -- every workgroup's atomic is served at the memory side, ~0.3 us each and one after the other
                // per counter (1080p levels 0+1, 8640 single-wave items: 8.8 -> 14.3 us per frame; 4K levels 2+3: 2.8 -> 3.7) -- so the
                // counters are on only for the wide launches whose workgroups carry 4 strips
                bool tickets = c->d_ticket && (c->env.band2_ticket == 1 ||
                                               (c->env.band2_ticket != 0 && waves2 >= 2 * c->wave_capacity2 && wpb == BAND2_WPB_MAX && a.w >= 2560));
                if (tickets && hipMemsetAsync(c->d_ticket, 0, 16 * sizeof(int), st) != hipSuccess) { (void)hipGetLastError(); tickets = false; }
                a.n_items = n_items;
                a.tickets = tickets ? c->d_ticket : nullptr;
                const int n_wg = tickets ? ((n_items + n_items / 8 + 7) / 8 * 8) : n_items;
                const dim3 grid2((unsigned int)n_wg), block2(64 * wpb);
                if (c->env.debug_variant)                // tests: which variant was launched
                    fprintf(stderr, "fvvdp: levels %d+%d: band2_kernel<%d, %s>, luminance range %s [%g, %g], widest plane range %g, %d waves per workgroup\n",
                                 b, b + 1, c->P, inrange ? "true" : "false", c->lum_state == 1 ? "known" : "unknown", c->lum_lo, c->lum_hi, c->lum_width, wpb);
                if (c->P == 4) {
                    if (inrange) hipLaunchKernelGGL((band2_kernel<4, true>), grid2, block2, 0, st, a);
                    else hipLaunchKernelGGL((band2_kernel<4, false>), grid2, block2, 0, st, a);
                } else {
                    if (inrange) hipLaunchKernelGGL((band2_kernel<2, true>), grid2, block2, 0, st, a);
                    else hipLaunchKernelGGL((band2_kernel<2, false>), grid2, block2, 0, st, a);
                }
            }
            for (int bb = b; bb <= b + 1; ++bb) {
                fa.nblk[bb] = nblk;
                fa.off[bb] = c->partial_off[bb];
                fa.npx[bb] = (float)c->lw[bb] * (float)c->lh[bb];
            }
            ++b;                                             // band b+1 is done as well
            continue;
        }
        BandArgs a;
        memset(&a, 0, sizeof(a));
        a.F = level_addr(c, b, b == 0 ? slot0 : 0);
        a.Gc = c->level[b + 1];
        a.w = c->lw[b];
        a.h = c->lh[b];
        a.wc = c->lw[b + 1];
        a.hc = c->lh[b + 1];
        a.n_strips = band_strips(a.wc);
        chunking(a.hc, a.n_strips, n, c->wave_capacity, c->env.band_cr, a.n_chunks, a.cr);
        a.band_mul = (b == 0) ? 1.0f : 2.0f;                 // lpyr.get_band, fvvdp_lpyr_dec.py:57-63
        a.csf = c->csf + (size_t)b * FVVDP_LUT_N;
        a.csf_y = c->csf_y;
        a.y_first = c->y_first;
        a.y_inv_step = c->y_inv_step;
        a.y_lo = c->y_lo;
        a.y_hi = c->y_hi;
        a.ly_lo = log2f(c->y_lo);
        a.ly_hi = log2f(c->y_hi);
        a.lg_gain = log2f(c->prm.sens_gain);
        a.lg_k = log2f(c->prm.mask_k);
        a.p = c->prm.mask_p;
        a.q0 = c->prm.mask_q[0];
        a.q1 = c->prm.mask_q[1];
        a.beta = c->prm.beta;
        a.lbkg_min = c->prm.lbkg_min;
        a.cmax = c->prm.contrast_max;
        a.lg_dmax = log2f(c->prm.d_max);
        a.partial = c->partial + c->partial_off[b];
        bool dbg = false;
        if (maps) {
            a.dD = maps[b].d_D;
            a.dC = maps[b].d_contrast;
            a.dL = maps[b].d_lbkg;
            a.dS = maps[b].d_S;
            dbg = a.dD || a.dC || a.dL || a.dS;
        }
        if (fov) {
            a.sublut = c->sublut[b];
            a.axes = c->d_axes;
            a.rw = c->sub_rw[b];
            a.i_lo = c->sub_ilo[b];
            a.fix = c->d_fix;
            if (geom) {
                a.size_m0 = geom->display_size_m[0];
                a.size_m1 = geom->display_size_m[1];
                a.dist_m = geom->distance_m;
                const double delta = (1.0 / (double)geom->ppd_centre) / 2.0 * M_PI / 180.0;
                a.delta_rad = (float)delta;
                a.cos_delta = (float)cos(delta);
            } else {
                a.size_m0 = a.size_m1 = a.dist_m = 1.0f;
                a.mvx = c->map_vx[b];
                a.mvy = c->map_vy[b];
                a.mrm = c->map_rm[b];
            }
            a.rho_band = (float)c->rho_band[b];
            a.rho_lo = c->rho_lo;
            a.rho_hi = c->rho_hi;
            a.ecc_lo = c->ecc_lo;
            a.ecc_hi = c->ecc_hi;
            for (int ax = 0; ax < 3; ++ax) {
                a.first[ax] = c->h_axes[ax][0];
                a.inv_step[ax] = (float)(FVVDP_LUT_N - 1) / (c->h_axes[ax][FVVDP_LUT_N - 1] - c->h_axes[ax][0]);
                const double step = ((double)c->h_axes[ax][FVVDP_LUT_N - 1] - (double)c->h_axes[ax][0]) / (FVVDP_LUT_N - 1);
                a.frac_scale[ax] = (float)(step / (step + 1e-6));
                a.grid_off[ax] = -a.first[ax] * a.inv_step[ax];
            }
            a.frame_w = c->W;
            a.frame_h = c->H;
            a.rmap = (geom && !c->env.fov_no_rhomap) ? c->rmap[b] : nullptr;
            a.rmap_w = (c->lw[b] + 1) / 2;
        }
        const int nblk = a.n_strips * a.n_chunks;
        if (nblk > c->max_blk[b]) return fail(FVVDP_ESTATE, "internal: partial buffer too small");
        a.n_items = nblk * n;
        {   // LUT slice in LDS if it fits next to the row table (64 KB of dynamic LDS without opting in to more)
            const size_t lut_b = (size_t)FOV_PLANE * c->sub_rw[b] * sizeof(float4);       // 18.6 KB per rho interval
            // the kernel's static tables (s_csf 512 B + s_ax 768 B) share the 64 KB with the dynamic part
            const size_t lds_static = sizeof(float4) * FVVDP_LUT_N + sizeof(float2) * 3 * FVVDP_LUT_N;
            a.lut_lds = (fov && lut_b <= 56 * 1024 && lut_b + (size_t)c->lh[b] * sizeof(float) + lds_static <= 64 * 1024) ? 1 : 0;
        }
        {
            Timed tm(c, 1 + b, st);
            if (c->P == 4) launch_band<4>(a, nblk * n, dbg, fov, st);
            else launch_band<2>(a, nblk * n, dbg, fov, st);
        }
        fa.nblk[b] = nblk;
        fa.off[b] = c->partial_off[b];
        fa.npx[b] = (float)a.w * (float)a.h;
    }
    fa.partial = c->partial;
    fa.Q = d_Q;
    fa.n_bands = c->n_bands;
    fa.n = n;
    fa.q_stride = q_stride;
    fa.q_col0 = q_col0;
    fa.tc = c->P / 2;
    fa.inv_beta = 1.0f / c->prm.beta;
    {
        Timed tm(c, 1 + c->n_bands, st);
        const int total = c->n_bands * 2 * n;
        hipLaunchKernelGGL(finalize_kernel, dim3(total), dim3(64), 0, st, fa);
    }
    if (pool_now) {
        PoolArgs pa;
        fill_pool_args(pa, d_Q, c->n_bands, 2, q_stride, q_stride, pool, d_jod);
        hipLaunchKernelGGL(pool_jod_kernel, dim3(1), dim3(256), 0, st, pa);
    }
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

extern "C" int fvvdp_bands_forward(fvvdp_ctx* c, int n, float* d_Q, int q_stride, int q_col0, const float* h_fixation,
                                   const fvvdp_geom* geom, const fvvdp_band_maps* maps, void* stream) {
    return bands_forward_core(c, 0, n, d_Q, q_stride, q_col0, h_fixation, geom, maps, nullptr, nullptr, stream);
}

extern "C" int fvvdp_bands_forward_pool(fvvdp_ctx* c, int n, float* d_Q, int q_stride, int q_col0, const float* h_fixation,
                                        const fvvdp_geom* geom, const fvvdp_band_maps* maps, const fvvdp_pool_params* pool,
                                        float* d_jod, void* stream) {
    if (!pool) return fail(FVVDP_EINVAL, "null argument");
    return bands_forward_core(c, 0, n, d_Q, q_stride, q_col0, h_fixation, geom, maps, pool, d_jod, stream);
}

// ---- placement of level 0, at context creation ------------------------------------------------------------------------------
// profiles/r05_k1_mode.md: the memory of a box comes in two classes.  Streaming writes into ONE class top out at ~5.5 TB/s, into both at
// once at 7.0 -- and that, not anything in the kernel, is the temporal kernel's slow and fast
// mode (35-36 against 30-31 us per 4K frame).  An allocation of several GB lies in one class or across both as the driver's allocator
// happens to stand, whatever the API; but among a handful of allocations of different kinds both classes turn up (12 buffers of one
// process split cleanly in two groups, in each of three processes: section 6 of the file).  So level 0 of a large video context lives in
// TWO ranges -- even frame slots in one, odd slots in the other (L0Addr, device_common.hpp: one address formula in every kernel that
// touches level 0) -- and the two are CHOSEN: N half-size candidates (default 6: chunk-mapped and hipMalloc in
// turn), every pair written at once by a streaming-write probe (4 ms per pair), the pair with the highest rate is kept, the
// rest freed.  Two halves of different classes are the fast mode by construction; if every candidate lies in one class (no pair above
// 6.75 TB/s), up to 8 further candidates are taken while the first ones are held, each written together with one range of the best pair,
// until a pair of different classes turns up; failing that the best pair is as good as any single buffer.  All of it happens here, before the first user call (~0.15 s, N x half a level 0 held for the moment);
// per-frame calls never allocate, free or synchronise for it.  Results never depend on it.  Video contexts whose level 0 holds >= 1 GiB
// (FVVDP_PLACEMENT_PROBE=n candidates, 0 / 1 = level 0 stays the single range it was allocated as; FVVDP_LEVEL0_SPLIT=1 splits any
// context without probing: tests).
static void choose_level0(fvvdp_ctx* c) {
    c->sel_phase = 9;
    c->level0_kind = vmm_owns(c, c->level[0]) ? 1 : 0;
    const int HW = c->W * c->H;
    const size_t frame_floats = (size_t)HW * c->P;
    const size_t bytes = (size_t)c->max_frames * frame_floats * sizeof(float);
    const int half_frames = (c->max_frames + 1) / 2;
    const size_t half_bytes = (size_t)half_frames * frame_floats * sizeof(float);
    int n_cand = c->env.probe_n < 0 ? 6 : c->env.probe_n;
    if (n_cand > 8) n_cand = 8;
    const bool eligible = c->P == 4 && bytes >= ((size_t)1 << 30) && c->max_frames >= 16 && (HW % 4) == 0;
    auto alloc_kind = [&](int kind, size_t nbytes) -> float* {
        void* q = nullptr;
        const bool got = kind == 1 ? vmm_alloc(c, &q, nbytes) == FVVDP_OK : hipMalloc(&q, nbytes) == hipSuccess;
        if (!got) { (void)hipGetLastError(); return nullptr; }
        return reinterpret_cast<float*>(q);
    };
    auto free_any = [&](float* p) {
        if (!p) return;
        if (vmm_owns(c, p)) vmm_free_one(c, p); else (void)hipFree(p);
    };
    if (c->env.level0_split == 1 && !(eligible && n_cand >= 2)) {
        // tests: two ranges without a choice (any context; the range the context came with keeps the even slots)
        c->level0_hi = alloc_kind(0, half_bytes);
        c->level0_kind_hi = c->level0_hi ? 0 : -1;
        return;
    }
    if (!eligible || n_cand < 2) return;
    {   // room for the candidates next to a margin: the first two are taken WHILE the full-size range the context came with is still
        // held (it is given back only once two halves are in hand, so a failed probe leaves the context as it was), the others after
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
        if (free_b < 2 * half_bytes + ((size_t)3 << 30)) return;
        while (n_cand >= 2 && free_b + bytes < (size_t)n_cand * half_bytes + ((size_t)3 << 30)) --n_cand;
        if (n_cand < 2) return;
    }
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess) {
        (void)hipGetLastError();
        if (st) (void)hipStreamDestroy(st);
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        return;
    }
    // candidates: halves of level 0 (the first one of the kind the context's allocation mode asks for)
    (void)hipDeviceSynchronize();
    float* cand[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int kind[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int n_got = 0;
    for (int k = 0; k < n_cand; ++k) {
        if (k == 2) {                       // two halves secured: level 0 can live in them whatever happens next -> the original goes
            free_any(c->level[0]);
            c->level[0] = nullptr;
        }
        const int kd = c->env.alloc_malloc ? 0 : (k % 2 == 0 ? 1 : 0);
        float* q = alloc_kind(kd, half_bytes);
        if (!q) break;
        cand[n_got] = q;
        kind[n_got] = kd;
        ++n_got;
    }
    if (n_got < 2) {                        // not even two halves next to the original: the context keeps the range it came with
        for (int k = 0; k < n_got; ++k) free_any(cand[k]);
        (void)hipStreamDestroy(st);
        for (auto& e : ev) (void)hipEventDestroy(e);
        (void)hipGetLastError();
        return;
    }
    if (c->level[0]) {                      // exactly two candidates asked for
        free_any(c->level[0]);
        c->level[0] = nullptr;
    }
    auto pair_rate = [&](float* p0, float* p1) -> float {                    // TB/s of writing both halves at once; 0 on failure
        const size_t n4 = half_bytes / 16;
        float best = 0.0f;
        for (int rep = 0; rep < 3; ++rep) {
            if (rep) (void)hipEventRecord(ev[0], st);
            hipLaunchKernelGGL(stream_write_probe_kernel, dim3(32768), dim3(256), 0, st, reinterpret_cast<float4*>(p0), reinterpret_cast<float4*>(p1), n4, 1.0f);
            if (!rep) continue;                                              // first touch / warm-up, untimed
            float ms = 0.0f;
            if (hipEventRecord(ev[1], st) != hipSuccess || hipEventSynchronize(ev[1]) != hipSuccess ||
                hipEventElapsedTime(&ms, ev[0], ev[1]) != hipSuccess || !(ms > 0.0f)) { (void)hipGetLastError(); return 0.0f; }
            const float r = (float)(2.0 * (double)half_bytes / (ms * 1e-3) / 1e12);
            best = r > best ? r : best;
        }
        return best;
    };
    int bi = 0, bj = n_got > 1 ? 1 : 0;
    float r_best = 0.0f, r_worst = 0.0f;
    if (n_got >= 2) {
        for (int i = 0; i < n_got; ++i)
            for (int j = i + 1; j < n_got; ++j) {
                const float r = pair_rate(cand[i], cand[j]);
                if (r > r_best) { r_best = r; bi = i; bj = j; }
                if (r > 0.0f && (r_worst == 0.0f || r < r_worst)) r_worst = r;
            }
    }
    // No pair of different classes among them (pairs of one class: 5.3-6.5 TB/s, of two: 6.9-7.2; profiles/r05_k1_mode.md section 7 -- it
    // happens where allocations come in runs of one class, e.g. the second and fourth of eight contexts of one process): take further
    // candidates while HOLDING the ones there are, so that the allocator has to move on, and write each together with one range of the
    // best pair; stop at the first pair of different classes.
    int n_extra = 0;
    std::vector<float*> held;
    // (probe_mixed is an MI355X figure: one class of its memory takes ~5.5 TB/s of streaming writes, both at once 7.0.  A part whose best
    // pair stays below 0.7 x that rate (4.7 TB/s) -- MI300X: 5.3 TB/s peak -- cannot reach it with any pair: no further candidates there.)
    const float reachable = 0.7f * 6.75f;      // (of the MI355X figure itself, not of an overridden FVVDP_PLACEMENT_MIXED_TBS)
    if (n_got >= 2 && r_best >= reachable && r_best < c->env.probe_mixed) {
        const int max_extra = c->env.probe_extra < 0 ? 8 : c->env.probe_extra;
        for (int k = 0; k < max_extra; ++k) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); break; }
            if (free_b < half_bytes + ((size_t)3 << 30)) break;
            const int kd = c->env.alloc_malloc ? 0 : ((n_got + k) % 2 == 0 ? 1 : 0);
            float* q = alloc_kind(kd, half_bytes);
            if (!q) break;
            ++n_extra;
            const float r = pair_rate(cand[bi], q);
            if (r > 0.0f && (r_worst == 0.0f || r < r_worst)) r_worst = r;
            if (r > r_best) {
                held.push_back(cand[bj]);
                cand[bj] = q;
                kind[bj] = kd;
                r_best = r;
            } else {
                held.push_back(q);
            }
            if (r_best >= c->env.probe_mixed) break;
        }
    }
    (void)hipStreamSynchronize(st);
    for (auto& e : ev) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(st);
    (void)hipGetLastError();
    for (float* q : held) free_any(q);
    if (n_got >= 2) {
        c->level[0] = cand[bi];
        c->level0_hi = cand[bj];
        c->level0_kind = kind[bi];
        c->level0_kind_hi = kind[bj];
        for (int k = 0; k < n_got; ++k)
            if (k != bi && k != bj) free_any(cand[k]);
    } else {
        // not even two halves: back to one range
        for (int k = 0; k < n_got; ++k) free_any(cand[k]);
        float* q = alloc_kind(c->env.alloc_malloc ? 0 : 1, bytes);
        if (!q) q = alloc_kind(0, bytes);
        c->level[0] = q;
        c->level0_kind = vmm_owns(c, q) ? 1 : 0;
        return;
    }
    c->sel_n = n_got;
    c->sel_kept = bi + 8 * bj;
    c->sel_us[1] = r_best;
    c->sel_us[2] = r_worst;
    c->sel_us[3] = (float)n_extra;
    if (c->env.debug_variant)
        fprintf(stderr, "fvvdp: level 0 in two ranges: %d half-size candidates + %d further ones, pairs written at once at %.2f ... %.2f TB/s -> kept #%d (kind %d) + #%d (kind %d)\n",
                n_got, n_extra, r_worst, r_best, bi, kind[bi], bj, kind[bj]);
    // (Until round 5 the temporal kernel and the pyramid pass were run here once on a synthetic clip "for the record", through the
    // real entry points on the context under construction -- it had to overwrite and hand-restore the CSF tables, the luminance-range
    // state and two switches.  The figure was only ever reported; the pass is gone, and with it the state it touched: ADVICE r5.)
}

extern "C" int fvvdp_ctx_set_view_maps(fvvdp_ctx* c, int band, const float* d_view_x, const float* d_view_y,
                                       const float* d_res_mag, float res_mag_min, float res_mag_max) {
    if (!c) return fail(FVVDP_EINVAL, "null context");
    if (band < 0 || band >= c->n_bands) return fail(FVVDP_EINVAL, "band %d out of range", band);
    if (!d_view_x || !d_view_y || !d_res_mag) {      // clear
        for (int b = 0; b < FVVDP_MAX_BANDS; ++b) c->map_vx[b] = c->map_vy[b] = c->map_rm[b] = nullptr;
        c->maps_set = false;
        c->sub_valid = false;
        return FVVDP_OK;
    }
    c->map_vx[band] = d_view_x;
    c->map_vy[band] = d_view_y;
    c->map_rm[band] = d_res_mag;
    c->map_rm_max[band] = res_mag_max;
    c->map_rm_min[band] = res_mag_min;
    c->maps_set = true;
    for (int b = 0; b < c->n_bands; ++b)
        if (!c->map_vx[b]) c->maps_set = false;      // complete only when every band has its maps
    c->sub_valid = false;
    return FVVDP_OK;
}

extern "C" int fvvdp_pool_jod(const float* d_Q, int n_bands, int n_channels, int n_frames, int q_stride,
                              const fvvdp_pool_params* prm, float* d_jod, void* stream) {
    if (!d_Q || !prm || !d_jod) return fail(FVVDP_EINVAL, "null argument");
    if (n_bands < 1 || n_frames < 1 || q_stride < n_frames) return fail(FVVDP_EINVAL, "bad Q_per_ch shape");
    if (n_channels != 1 && n_channels != 2) return fail(FVVDP_EINVAL, "n_channels must be 1 (image) or 2 (video)");
    {
        int rc = check_pool_params(prm);
        if (rc != FVVDP_OK) return rc;
    }
    PoolArgs a;
    fill_pool_args(a, d_Q, n_bands, n_channels, n_frames, q_stride, prm, d_jod);
    hipLaunchKernelGGL(pool_jod_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

extern "C" int fvvdp_heatmap_reconstruct(fvvdp_ctx* c, int n, const float* const* h_dD, float w_transient, float beta_jod,
                                         float jod_a_abs, float* d_out, void* stream) {
    if (!c || !h_dD || !d_out) return fail(FVVDP_EINVAL, "null argument");
    if (n < 1 || n > c->max_frames) return fail(FVVDP_EINVAL, "n=%d exceeds max_frames=%d", n, c->max_frames);
    for (int b = 0; b < c->n_bands; ++b)
        if (!h_dD[b]) return fail(FVVDP_EINVAL, "missing D map of band %d", b);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int b = 1; b < c->n_bands; ++b)
        if (!c->heat[b]) {
            int rc = dev_alloc(c, &c->heat[b], (size_t)c->max_frames * c->lw[b] * c->lh[b]);
            if (rc != FVVDP_OK) return rc;
        }
    for (int b = c->n_bands - 1; b >= 0; --b) {
        HeatArgs a;
        memset(&a, 0, sizeof(a));
        a.D = h_dD[b];
        a.coarse = (b == c->n_bands - 1) ? nullptr : c->heat[b + 1];   // the base band of a zero image is zero
        a.out = (b == 0) ? d_out : c->heat[b];
        a.w = c->lw[b];
        a.h = c->lh[b];
        a.wc = c->lw[b + 1];
        a.hc = c->lh[b + 1];
        a.tc = c->P / 2;
        a.w_trans = w_transient;
        a.inv_m = (b == 0) ? 1.0f : 0.5f;                              // set_band divides by the band multiplier
        a.beta_jod = beta_jod;
        a.scale = jod_a_abs;
        a.final_level = (b == 0);
        dim3 grid((a.w + 255) / 256, a.h, n), block(256);
        hipLaunchKernelGGL(heat_level_kernel, grid, block, 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

extern "C" int fvvdp_heatmap_colorize(fvvdp_ctx* c, int n, const float* d_dmap, const float* h_knots, const float* h_rgb,
                                      int n_knots, const float* h_lin01, void* d_out_f16, size_t chan_stride, void* stream) {
    if (!c || !d_dmap || !h_knots || !h_rgb || !h_lin01 || !d_out_f16) return fail(FVVDP_EINVAL, "null argument");
    if (n < 1 || n > c->max_frames) return fail(FVVDP_EINVAL, "n=%d exceeds max_frames=%d", n, c->max_frames);
    if (n_knots < 2 || n_knots > 8) return fail(FVVDP_EINVAL, "colour map needs 2..8 knots, got %d", n_knots);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t per_frame = 2 + 2 * COLOUR_BINS;                    // 32-bit words
    unsigned int* range;
    if (!c->colour_ws) {
        int rc = dev_alloc(c, &c->colour_ws, (size_t)c->max_frames * per_frame + COLOUR_BINS);
        if (rc != FVVDP_OK) return rc;
        // the knot table of the tone curve (torch.linspace from the caller) is uploaded once per context
        float* l = reinterpret_cast<float*>(c->colour_ws + (size_t)c->max_frames * per_frame);
        HIP_TRY(hipMemcpyAsync(l, h_lin01, COLOUR_BINS * sizeof(float), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        c->n_sync += 1;
    }
    range = c->colour_ws;                                            // [2][max_frames]: min positive, max
    unsigned int* hist = range + 2 * (size_t)c->max_frames;
    float* curve = reinterpret_cast<float*>(hist + (size_t)c->max_frames * COLOUR_BINS);
    float* lin01 = curve + (size_t)c->max_frames * COLOUR_BINS;
    HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(range), 0x7F800000, (size_t)n, st));           // +inf
    HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(range + c->max_frames), 0, (size_t)n, st));
    HIP_TRY(hipMemsetAsync(hist, 0, (size_t)n * COLOUR_BINS * sizeof(unsigned int), st));
    ColourArgs a;
    memset(&a, 0, sizeof(a));
    a.ctx = level_addr(c, 0, 0);
    a.P = c->P;
    a.HW = (unsigned int)(c->W * c->H);
    a.range = range;
    a.range_stride = c->max_frames;
    a.hist = hist;
    a.curve = curve;
    a.lin01 = lin01;
    a.dmap = d_dmap;
    a.out = reinterpret_cast<__half*>(d_out_f16);
    a.chan_stride = chan_stride;
    a.n_knots = n_knots;
    for (int k = 0; k < 8; ++k) {
        const int kk = k < n_knots ? k : n_knots - 1;
        a.knots[k] = h_knots[kk];
        for (int ch = 0; ch < 3; ++ch) a.rgb[k][ch] = h_rgb[3 * kk + ch];
    }
    a.dr = 0.6f;
    const unsigned int blocks = (a.HW + 256 * 16 - 1) / (256 * 16);  // >= 16 pixels per thread
    dim3 grid(blocks < 1 ? 1 : blocks, n);
    // the two reductions end in atomics on a few addresses per frame: bound the number of blocks that flush
    dim3 grid_red(grid.x > 96 ? 96 : grid.x, n);
    hipLaunchKernelGGL(colour_range_kernel, grid_red, dim3(256), 0, st, a);
    hipLaunchKernelGGL(colour_hist_kernel, grid_red, dim3(256), 0, st, a);
    hipLaunchKernelGGL(colour_curve_kernel, dim3(n), dim3(COLOUR_BINS), 0, st, a);
    hipLaunchKernelGGL(colour_map_kernel, grid, dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

extern "C" int fvvdp_ctx_timing_enable(fvvdp_ctx* c, int on) {
    if (!c) return fail(FVVDP_EINVAL, "null context");
    c->timing = on != 0;
    return FVVDP_OK;
}

extern "C" int fvvdp_ctx_timing_read(fvvdp_ctx* c, float* h_ms, int32_t* h_count, int capacity, int reset) {
    if (!c || !h_ms || !h_count) return fail(FVVDP_EINVAL, "null argument");
    const int nk = c->n_bands + 2;
    if (capacity < nk) return fail(FVVDP_EINVAL, "capacity %d < %d kernels", capacity, nk);
    for (int id = 0; id < nk; ++id) {
        for (auto& pr : c->ev[id]) {
            HIP_TRY(hipEventSynchronize(pr.second));
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
            c->t_ms[id] += ms;
            c->t_cnt[id] += 1;
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        c->ev[id].clear();
        h_ms[id] = c->t_ms[id];
        h_count[id] = c->t_cnt[id];
        if (reset) {
            c->t_ms[id] = 0.0f;
            c->t_cnt[id] = 0;
        }
    }
    return FVVDP_OK;
}


extern "C" int fvvdp_ctx_alloc_info(const fvvdp_ctx* c, int* state, int* chunk_mapped, float* h_us, int capacity, int* n_timed,
                                    int* kept) {
    if (!c || !state || !chunk_mapped || !h_us || !n_timed || !kept || capacity < 1) return fail(FVVDP_EINVAL, "null argument");
    *state = c->sel_phase;
    // one range: its kind (0 hipMalloc, 1 chunk-mapped); two ranges: 100 + 10 * kind of the odd slots + kind of the even slots
    *chunk_mapped = c->level0_hi ? 100 + 10 * c->level0_kind_hi + c->level0_kind : c->level0_kind;
    for (int k = 0; k < capacity; ++k) h_us[k] = (k < 4 && c->sel_n > 0) ? c->sel_us[k] : 0.0f;
    *n_timed = c->sel_n;
    *kept = c->sel_kept;
    return FVVDP_OK;
}

extern "C" int fvvdp_ctx_call_stats(const fvvdp_ctx* c, int64_t* h_counts3) {
    if (!c || !h_counts3) return fail(FVVDP_EINVAL, "null argument");
    h_counts3[0] = c->n_sync;
    h_counts3[1] = c->n_alloc;
    h_counts3[2] = c->n_free;
    return FVVDP_OK;
}

#ifdef BAND2_TIMELINE      // profiling build only (tools/build_variant.sh timeline "-DBAND2_TIMELINE"): not part of the C ABI
extern "C" int fvvdp_debug_timeline(unsigned long long* h_out, size_t n_records) {
    if (n_records > 65536) n_records = 65536;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_band2_timeline), n_records * 4 * sizeof(unsigned long long)));
    return FVVDP_OK;
}
#endif
