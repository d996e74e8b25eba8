// preprocess.hip -- the reference's image transform on the GPU:
//   targetpad_transform(1.25, 224) = TargetPad -> Resize(224, BICUBIC) -> CenterCrop(224) -> ToTensor -> Normalize
//   (/root/reference/src/data_utils.py:49-72, :91-105).  Input: one DECODED image, uint8 RGB, HWC (decoding stays on the host).
//
// torchvision's Resize on a PIL image is PIL's 8-bit resampler (libImaging/Resample.c): per output coordinate a window of
// bicubic (a = -0.5) taps whose support grows with the downscale factor, weights normalised in double and rounded to
// 22-bit fixed point, pixel = clip8((2^21 + sum pixel_k * w_k) >> 22); HORIZONTAL pass first into a uint8 intermediate,
// then the VERTICAL pass.  Bit-exact parity with the reference needs exactly that arithmetic:
//   * the tap tables are computed on the HOST in double, statement for statement as PIL does (no FMA contraction on the
//     x86-64 baseline; a GPU evaluation would contract a*b+c), and copied into the caller's workspace;
//   * the two kernels do the integer multiply-accumulate: HBM-bound byte work, one thread per output pixel (3 channels),
//     only the rows / columns the centre crop keeps; zero padding is a bounds test, not a copy;
//   * ToTensor / Normalize: (u8 / 255.f - mean) / std with IEEE fp32 divisions, as torch does.
#include <math.h>

#include <vector>

#include "common.hpp"

namespace sprc {

constexpr int PRECISION_BITS = 32 - 8 - 2;

struct Taps { std::vector<int32_t> lo, cnt, kk; int ksize = 0; };

static double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// Resample.c: precompute_coeffs + normalize_coeffs_8bpc for output coordinates [first, first + n)
static void make_taps(int in_size, int out_size, int first, int n, Taps& t) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    t.ksize = (int)ceil(support) * 2 + 1;
    t.lo.assign(n, 0); t.cnt.assign(n, 0); t.kk.assign((size_t)n * t.ksize, 0);
    const double ss = 1.0 / filterscale;
    std::vector<double> w(t.ksize);
    for (int i = 0; i < n; ++i) {
        const int xx = first + i;
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            w[x] = bicubic_filter((x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) w[x] /= ww;
            t.kk[(size_t)i * t.ksize + x] = w[x] < 0 ? (int)(-0.5 + w[x] * (1 << PRECISION_BITS)) : (int)(0.5 + w[x] * (1 << PRECISION_BITS));
        }
        t.lo[i] = xmin; t.cnt[i] = xmax;
    }
}

struct Geometry { int hp, vp, pw, ph, rw, rh, left, top; };

// data_utils.py:62-72 (TargetPad); torchvision Resize(int): short side -> dim, long side int(dim * long / short);
// CenterCrop: int(round((size - dim) / 2))  (Python round: half to even)
static Geometry geometry(int w, int h, double ratio, int dim) {
    Geometry g{0, 0, 0, 0, 0, 0, 0, 0};
    const int mx = w > h ? w : h, mn = w > h ? h : w;
    if ((double)mx / mn >= ratio) {
        const double scaled = (double)mx / ratio;
        g.hp = (int)((scaled - w) / 2); if (g.hp < 0) g.hp = 0;
        g.vp = (int)((scaled - h) / 2); if (g.vp < 0) g.vp = 0;
    }
    g.pw = w + 2 * g.hp; g.ph = h + 2 * g.vp;
    if (g.pw <= g.ph) { g.rw = dim; g.rh = (int)((double)dim * g.ph / g.pw); }
    else { g.rw = (int)((double)dim * g.pw / g.ph); g.rh = dim; }
    g.left = (int)nearbyint((g.rw - dim) / 2.0);          // default rounding mode: to nearest, ties to even == Python round
    g.top = (int)nearbyint((g.rh - dim) / 2.0);
    return g;
}

struct PreParams {
    const uint8_t* src; int h, w; int64_t stride;
    int hp, vp;                       // zero padding on each side
    int dim, left, top;               // crop offsets (used by a pass that is the identity)
    const int32_t *h_lo, *h_cnt, *h_kk; int h_ksize;      // horizontal taps for output columns left .. left+dim (null: identity)
    const int32_t *v_lo, *v_cnt, *v_kk; int v_ksize;      // vertical taps for output rows top .. top+dim (null: identity)
    int row0, rows;                   // rows of the padded image the vertical pass reads
    uint8_t* tmp;                     // [rows, dim, 3]
    float* out;                       // [3, dim, dim]
    float mean[3], std[3];
};

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// tmp[y - row0][x][c] = horizontal pass of padded row y at output column left + x
__global__ __launch_bounds__(256) void resample_h_kernel(PreParams p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.rows * p.dim) return;
    const int yr = i / p.dim, x = i - yr * p.dim;
    const int sy = p.row0 + yr - p.vp;                     // source row (outside [0, h): padding)
    int r0 = 0, r1 = 0, r2 = 0;
    if (sy >= 0 && sy < p.h) {
        const uint8_t* row = p.src + (int64_t)sy * p.stride;
        if (p.h_kk == nullptr) {
            const int sx = p.left + x - p.hp;
            if (sx >= 0 && sx < p.w) { r0 = row[sx * 3]; r1 = row[sx * 3 + 1]; r2 = row[sx * 3 + 2]; }
        } else {
            const int lo = p.h_lo[x], n = p.h_cnt[x];
            const int32_t* kk = p.h_kk + (int64_t)x * p.h_ksize;
            int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
            for (int k = 0; k < n; ++k) {
                const int sx = lo + k - p.hp;
                if (sx >= 0 && sx < p.w) {
                    const int wk = kk[k];
                    a0 += row[sx * 3] * wk; a1 += row[sx * 3 + 1] * wk; a2 += row[sx * 3 + 2] * wk;
                }
            }
            r0 = clip8(a0 >> PRECISION_BITS); r1 = clip8(a1 >> PRECISION_BITS); r2 = clip8(a2 >> PRECISION_BITS);
        }
    } else if (p.h_kk != nullptr) {
        // a padded (all-zero) row still goes through the rounding of the pass: clip8((2^21 + 0) >> 22) = 0
        r0 = r1 = r2 = 0;
    }
    uint8_t* o = p.tmp + (int64_t)i * 3;
    o[0] = (uint8_t)r0; o[1] = (uint8_t)r1; o[2] = (uint8_t)r2;
}

// out[c][yy][xx] = ((vertical pass of tmp at output row top + yy) / 255 - mean[c]) / std[c]
__global__ __launch_bounds__(256) void resample_v_kernel(PreParams p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.dim * p.dim) return;
    const int yy = i / p.dim, xx = i - yy * p.dim;
    int r[3];
    if (p.v_kk == nullptr) {
        const uint8_t* t = p.tmp + ((int64_t)(p.top + yy - p.row0) * p.dim + xx) * 3;
        r[0] = t[0]; r[1] = t[1]; r[2] = t[2];
    } else {
        const int lo = p.v_lo[yy] - p.row0, n = p.v_cnt[yy];
        const int32_t* kk = p.v_kk + (int64_t)yy * p.v_ksize;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
        for (int k = 0; k < n; ++k) {
            const uint8_t* t = p.tmp + ((int64_t)(lo + k) * p.dim + xx) * 3;
            const int wk = kk[k];
            a0 += t[0] * wk; a1 += t[1] * wk; a2 += t[2] * wk;
        }
        r[0] = clip8(a0 >> PRECISION_BITS); r[1] = clip8(a1 >> PRECISION_BITS); r[2] = clip8(a2 >> PRECISION_BITS);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        p.out[((int64_t)c * p.dim + yy) * p.dim + xx] = ((float)r[c] / 255.0f - p.mean[c]) / p.std[c];
}

struct Plan { Geometry g; Taps th, tv; bool need_h, need_v; int row0, rows; size_t off_tmp, off_h, off_v, total; };

static void plan(int h, int w, double ratio, int dim, Plan& pl, bool tables) {
    pl.g = geometry(w, h, ratio, dim);
    const Geometry& g = pl.g;
    pl.need_h = g.rw != g.pw;
    pl.need_v = g.rh != g.ph;
    if (pl.need_v) {
        make_taps(g.ph, g.rh, g.top, dim, pl.tv);
        int lo = pl.tv.lo[0], hi = 0;
        for (int i = 0; i < dim; ++i) {
            if (pl.tv.lo[i] < lo) lo = pl.tv.lo[i];
            if (pl.tv.lo[i] + pl.tv.cnt[i] > hi) hi = pl.tv.lo[i] + pl.tv.cnt[i];
        }
        pl.row0 = lo; pl.rows = hi - lo;
    } else {
        pl.row0 = g.top; pl.rows = dim;
    }
    if (pl.need_h && tables) make_taps(g.pw, g.rw, g.left, dim, pl.th);
    else if (pl.need_h) { const double sc = (double)g.pw / g.rw; pl.th.ksize = (int)ceil(2.0 * (sc < 1.0 ? 1.0 : sc)) * 2 + 1; }
    size_t o = 0;
    pl.off_tmp = o; o = align_up(o + (size_t)pl.rows * dim * 3, 256);
    pl.off_h = o; o = align_up(o + (pl.need_h ? (size_t)dim * (2 + pl.th.ksize) * 4 : 0), 256);
    pl.off_v = o; o = align_up(o + (pl.need_v ? (size_t)dim * (2 + pl.tv.ksize) * 4 : 0), 256);
    pl.total = o;
}

}  // namespace sprc

using namespace sprc;

extern "C" size_t sprc_preprocess_workspace_bytes(int32_t src_h, int32_t src_w, float target_ratio, int32_t dim) {
    if (src_h <= 0 || src_w <= 0 || dim <= 0 || target_ratio <= 0.f) return 0;
    Plan pl;
    plan(src_h, src_w, (double)target_ratio, dim, pl, false);
    return pl.total + 256;
}

extern "C" int sprc_preprocess_targetpad(const uint8_t* src, int32_t src_h, int32_t src_w, int64_t src_stride, float target_ratio,
                                         int32_t dim, const float* mean, const float* std_, float* out, void* ws,
                                         size_t ws_bytes, sprc_stream s) {
    SPRC_REQUIRE(src && out && ws && mean && std_, "sprc_preprocess_targetpad: null pointer");
    SPRC_REQUIRE(src_h > 0 && src_w > 0 && dim > 0 && src_stride >= (int64_t)src_w * 3 && target_ratio > 0.f,
                 "sprc_preprocess_targetpad: bad shape %dx%d stride %lld", src_h, src_w, (long long)src_stride);
    SPRC_REQUIRE(((uintptr_t)ws % 256) == 0, "sprc_preprocess_targetpad: workspace must be 256-byte aligned");
    Plan pl;
    plan(src_h, src_w, (double)target_ratio, dim, pl, true);
    if (pl.total > ws_bytes) {
        set_error("sprc_preprocess_targetpad: workspace too small (%zu given, %zu needed)", ws_bytes, pl.total);
        return SPRC_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)s;
    char* base = (char*)ws;
    PreParams p;
    memset(&p, 0, sizeof(p));
    p.src = src; p.h = src_h; p.w = src_w; p.stride = src_stride;
    p.hp = pl.g.hp; p.vp = pl.g.vp; p.dim = dim; p.left = pl.g.left; p.top = pl.g.top;
    p.row0 = pl.row0; p.rows = pl.rows;
    p.tmp = (uint8_t*)(base + pl.off_tmp); p.out = out;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean[c]; p.std[c] = std_[c]; }
    auto upload = [&](const Taps& t, size_t off, const int32_t*& lo, const int32_t*& cnt, const int32_t*& kk, int& ksize) -> bool {
        int32_t* d = (int32_t*)(base + off);
        // pageable host memory: hipMemcpyAsync returns after the data has been staged, so the vectors may go away
        if (hipMemcpyAsync(d, t.lo.data(), (size_t)dim * 4, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        if (hipMemcpyAsync(d + dim, t.cnt.data(), (size_t)dim * 4, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        if (hipMemcpyAsync(d + 2 * dim, t.kk.data(), t.kk.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        lo = d; cnt = d + dim; kk = d + 2 * dim; ksize = t.ksize;
        return true;
    };
    if (pl.need_h && !upload(pl.th, pl.off_h, p.h_lo, p.h_cnt, p.h_kk, p.h_ksize)) { set_error("sprc_preprocess_targetpad: table upload failed"); return SPRC_ELAUNCH; }
    if (pl.need_v && !upload(pl.tv, pl.off_v, p.v_lo, p.v_cnt, p.v_kk, p.v_ksize)) { set_error("sprc_preprocess_targetpad: table upload failed"); return SPRC_ELAUNCH; }
    const double px_in = (double)pl.rows * dim, px_out = (double)dim * dim;
    ProfScope prof(SPRC_K_ROWOPS, st, 0.0, (double)src_h * src_w * 3 + px_in * 6 + px_out * 12);
    hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((pl.rows * dim + 255) / 256)), dim3(256), 0, st, p);
    SPRC_CHECK_LAUNCH("sprc_preprocess_targetpad(horizontal)");
    hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)((dim * dim + 255) / 256)), dim3(256), 0, st, p);
    SPRC_CHECK_LAUNCH("sprc_preprocess_targetpad(vertical)");
    return SPRC_OK;
}
