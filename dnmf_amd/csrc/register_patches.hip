// K8 -- position initialiser: piecewise-rigid registration shifts of a 3-D video against a template, and their application
// to neuron centres.
//
// Reference: Demix/MotionCorrect.py (vendored CaImAn / NoRMCorre; an orphan in the reference tree):
//   tile_and_correct_3d :1518-1608 (the shifts; shifts_opencv=True branch), called per frame with upsample_factor_fft=10
//   (tile_and_correct_wrapper :2004-2060); register_translation_3d :648-797; _upsampled_dft :498-614;
//   sliding_window_3d :1190-1221; MotionCorrect.apply_shifts_points :351-371.  oracle/motion_oracle.py restates them.
//
// What a registration is.  For a box of n0 x n1 x n2 voxels (the whole volume for the rigid shift, then every patch):
// P = DFT(frame box + add) . conj(DFT(template box + add)); the circular cross-correlation CC = IDFT(P) is searched for
// its largest magnitude over a small window of integer shifts (numpy slicing of the reference, emulated index by index),
// then P is transformed again onto a 15^3 grid of fractional shifts around that peak (the "upsampled DFT by matrix
// multiplication") and the largest magnitude there gives the shift to 1 / upsample_factor voxel.
//
// How it is computed here.  Everything is ONE primitive applied along one axis at a time: a matrix-vector DFT of an axis
// of length n onto m arbitrary positions delta_r = pos_r / uf,
//     out[r] = sum_j in[j] exp(sign 2 pi i f_j pos_r / (n uf)),   f_j the signed frequency of index j,
// with the phase kept as an exact integer index into a table of the (n uf)-th roots of unity in LDS (no angle is ever
// rounded).  m = n, pos_r = r, uf = 1 is the forward DFT; m = the window's kept indices is the inverse restricted to the
// window (the other n^3 - m^3 values of CC are never formed); m = 15, uf = 10 is the upsampled DFT.  No FFT library: the
// search windows are tiny, so the work is N (n0 + n1 + n2) complex multiply-adds per box for the forward transforms and
// little after that -- dense matrix work, like the reference's own _upsampled_dft.  fp32 throughout (the reference works
// in complex128): a peak is decided by magnitudes that differ by > 1e-4 relative between neighbouring fractional bins for
// anything but a flat correlation, three orders above the rounding of these sums.
#include <climits>
#include <cstdlib>

#include "common.hpp"

namespace dnmf {

constexpr int MC_MW = 32;        // kept indices of a search window per axis (max_shifts, max_deviation <= 15)
constexpr int MC_TC = 32;        // columns (lines of the transformed axis) per block
constexpr int MC_RPT = 8;        // output rows per thread
constexpr int MC_RTILE = 64;     // output rows per block: 8 thread groups x MC_RPT
constexpr int MC_MAXD = 8192;    // roots of unity in LDS (64 KB): n * upsample_factor of the longest axis

struct McBoxes {          // the boxes one call registers: all of one size, the same set for every frame
    int n[3];             // box size
    int nbox;             // boxes per frame
    const int *start;     // (nbox,3) first voxel of every box (device)
};

struct McAxis {
    // input
    int mode;             // 0: real voxels gathered from the frames (+ add), 1: complex, 2: complex times conj(other)
    const float *frames;  long ldf;  const int *frame_ids;   // mode 0
    int X, Y, Z;  float add;  int first_frame;
    const float2 *in;     // modes 1, 2: (items, outer, n, inner)
    const float2 *other;  // mode 2: (nbox, outer, n, inner), item -> box = item % nbox
    McBoxes bx;
    // the transform
    int outer, n, inner;  // view of one item's data; the middle index is transformed
    int m;                // output positions
    const int *pos;       // (items or 1, pos_ld) numerators of the positions; entries < 0 or >= m: skipped rows
    int pos_item_stride, pos_off;   // pos + item * pos_item_stride + pos_off
    int uf;               // positions are pos / uf
    int sign;             // -1 forward, +1 inverse
    float scale;
    float2 *out;          // (items, outer, m, inner)
};

// One block: one item, MC_TC columns, MC_RTILE output rows.
__global__ __launch_bounds__(256) void mc_axis_kernel(McAxis a) {
    extern __shared__ float2 s_root[];      // exp(2 pi i k / D), k = 0 .. D-1
    const int item = blockIdx.z;
    const int D = a.n * a.uf;
    for (int k = threadIdx.x; k < D; k += 256) {
        float sn, cs;
        sincospif(2.0f * (float)k / (float)D, &sn, &cs);
        s_root[k] = make_float2(cs, sn);
    }
    __syncthreads();
    const int c = threadIdx.x & (MC_TC - 1), g = threadIdx.x >> 5;
    const long ncol = (long)a.outer * a.inner;
    const long col = (long)blockIdx.x * MC_TC + c;
    const bool live = col < ncol;
    const long colc = live ? col : ncol - 1;
    const int o = (int)(colc / a.inner), i = (int)(colc - (long)o * a.inner);
    const int r0 = blockIdx.y * MC_RTILE + g * MC_RPT;
    // phase bookkeeping of this thread's rows: idx = (sign pos f_j) mod D, advanced by `step` per j and by `wrap` once,
    // where the frequency jumps from ceil(n/2) - 1 to ceil(n/2) - n
    const int *pos = a.pos + (long)item * a.pos_item_stride + a.pos_off;
    int step[MC_RPT], wrap[MC_RPT], idx[MC_RPT];
    bool rowok[MC_RPT];
    const int half = (a.n + 1) / 2;
#pragma unroll
    for (int q = 0; q < MC_RPT; ++q) {
        const int r = r0 + q;
        rowok[q] = r < a.m;
        long p = rowok[q] ? (long)pos[r] : 0;
        if (rowok[q] && pos[r] == INT_MIN) rowok[q] = false, p = 0;   // an unused slot of a window
        p *= a.sign;
        long s1 = p % D;
        if (s1 < 0) s1 += D;
        long s2 = (p * (1 - (long)a.n)) % D;
        if (s2 < 0) s2 += D;
        step[q] = (int)s1, wrap[q] = (int)s2, idx[q] = 0;
    }
    float2 acc[MC_RPT];
#pragma unroll
    for (int q = 0; q < MC_RPT; ++q) acc[q] = make_float2(0.0f, 0.0f);
    // where this thread's line starts
    const float *fsrc = nullptr;
    const float2 *csrc = nullptr, *osrc = nullptr;
    long stride = a.inner;
    const int box = item % a.bx.nbox;
    if (a.mode == 0) {
        // (only the z-pass reads voxels: outer = n0 n1, inner = 1, line = the n2 voxels of column (x, y) of the box)
        const int frame = a.first_frame + item / a.bx.nbox;
        const int x = o / a.bx.n[1], y = o - x * a.bx.n[1];
        const int *st = a.bx.start + 3 * box;
        fsrc = a.frames + (long)(a.frame_ids ? a.frame_ids[frame] : frame) * a.ldf +
               ((long)(st[0] + x) * a.Y + (st[1] + y)) * a.Z + st[2];
        stride = 1;
    } else {
        const long base = ((long)o * a.n) * a.inner + i;
        csrc = a.in + (long)item * a.outer * a.n * a.inner + base;
        if (a.mode == 2) osrc = a.other + (long)box * a.outer * a.n * a.inner + base;
    }
    for (int j = 0; j < a.n; ++j) {
        float2 v;
        if (a.mode == 0) {
            v = make_float2(fsrc[(long)j * stride] + a.add, 0.0f);
        } else {
            v = csrc[(long)j * stride];
            if (a.mode == 2) {
                const float2 w = osrc[(long)j * stride];   // v conj(w)
                v = make_float2(fmaf(v.x, w.x, v.y * w.y), fmaf(v.y, w.x, -v.x * w.y));
            }
        }
        const bool jump = j == half;   // f_j - f_{j-1} = 1 - n here, 1 elsewhere
#pragma unroll
        for (int q = 0; q < MC_RPT; ++q) {
            if (j > 0) {
                idx[q] += jump ? wrap[q] : step[q];
                idx[q] -= idx[q] >= D ? D : 0;
            }
            const float2 t = s_root[idx[q]];
            acc[q].x = fmaf(v.x, t.x, fmaf(-v.y, t.y, acc[q].x));
            acc[q].y = fmaf(v.x, t.y, fmaf(v.y, t.x, acc[q].y));
        }
    }
    if (!live) return;
    float2 *dst = a.out + (long)item * a.outer * a.m * a.inner + ((long)o * a.m) * a.inner + i;
#pragma unroll
    for (int q = 0; q < MC_RPT; ++q) {
        const int r = r0 + q;
        if (r < a.m) dst[(long)r * a.inner] = rowok[q] ? make_float2(acc[q].x * a.scale, acc[q].y * a.scale) : make_float2(0.0f, 0.0f);
    }
}

// The same primitive on the matrix pipe (round 3): out = W in with W[r][j] = exp(sign 2 pi i f_j pos_r / (n uf)) generated on
// the fly -- a complex GEMM as four f32 MFMAs (v_mfma_f32_32x32x2_f32: exact f32 products, fmaf accumulation) per step of two
// input rows j:  Re += Wre.vre - Wim.vim,  Im += Wre.vim + Wim.vre.  A wave owns a tile of 32 output rows x 32 columns (two
// 16-register accumulators); per step a lane supplies ONE entry of W (row l & 31, j = j0 + (l >> 5): one table lookup, its
// integer phase advanced by two frequencies) and ONE input value (j = j0 + (l >> 5), column l & 31: one load) -- the table
// lookups the vector kernel made per (row, j, column) are made per (row, j) here, and the multiply-adds leave the vector ALU.
// A block is 2 x 2 waves: 64 rows x 64 columns.  Same arguments, same results up to the order of the sums over j.
typedef float mc_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mc_axis_mfma_kernel(McAxis a) {
    extern __shared__ float2 s_root[];      // exp(2 pi i k / D), k = 0 .. D-1
    const int item = blockIdx.z;
    const int D = a.n * a.uf;
    for (int k = threadIdx.x; k < D; k += 256) {
        float sn, cs;
        sincospif(2.0f * (float)k / (float)D, &sn, &cs);
        s_root[k] = make_float2(cs, sn);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const long ncol = (long)a.outer * a.inner;
    const long col = (long)blockIdx.x * 64 + (wave & 1) * 32 + l31;      // the column this lane loads AND stores
    const bool live = col < ncol;
    const long colc = live ? col : ncol - 1;
    const int o = (int)(colc / a.inner), i = (int)(colc - (long)o * a.inner);
    const int row0 = blockIdx.y * 64 + (wave >> 1) * 32;                  // first output row of the wave's tile
    if (row0 >= a.m) return;                                              // (whole wave; no barrier below)
    // phase bookkeeping of this lane's W row r = row0 + l31: idx = (sign pos f_j) mod D for j = j0 + h
    const int *pos = a.pos + (long)item * a.pos_item_stride + a.pos_off;
    const int half = (a.n + 1) / 2;
    const int r = row0 + l31;
    bool rowok = r < a.m;
    long p = rowok ? (long)pos[r] : 0;
    if (rowok && pos[r] == INT_MIN) rowok = false, p = 0;                 // an unused slot of a window
    p *= a.sign;
    long s1 = p % D;
    if (s1 < 0) s1 += D;
    long s2 = (p * (1 - (long)a.n)) % D;
    if (s2 < 0) s2 += D;
    const int step = (int)s1, wrap = (int)s2;
    int idx = h == 0 ? 0 : (1 == half ? wrap : step);                     // j = h
    idx -= idx >= D ? D : 0;
    // where this lane's input column starts
    const float *fsrc = nullptr;
    const float2 *csrc = nullptr, *osrc = nullptr;
    long stride = a.inner;
    const int box = item % a.bx.nbox;
    if (a.mode == 0) {
        const int frame = a.first_frame + item / a.bx.nbox;
        const int x = o / a.bx.n[1], y = o - x * a.bx.n[1];
        const int *st = a.bx.start + 3 * box;
        fsrc = a.frames + (long)(a.frame_ids ? a.frame_ids[frame] : frame) * a.ldf + ((long)(st[0] + x) * a.Y + (st[1] + y)) * a.Z + st[2];
        stride = 1;
    } else {
        const long base = ((long)o * a.n) * a.inner + i;
        csrc = a.in + (long)item * a.outer * a.n * a.inner + base;
        if (a.mode == 2) osrc = a.other + (long)box * a.outer * a.n * a.inner + base;
    }
    mc_f32x16 are, aim;
#pragma unroll
    for (int q = 0; q < 16; ++q) are[q] = 0.0f, aim[q] = 0.0f;
    for (int j0 = 0; j0 < a.n; j0 += 2) {
        const int j = j0 + h;
        const bool jin = j < a.n;
        const int jc = jin ? j : a.n - 1;
        float2 v;
        if (a.mode == 0) {
            v = make_float2(fsrc[(long)jc * stride] + a.add, 0.0f);
        } else {
            v = csrc[(long)jc * stride];
            if (a.mode == 2) {
                const float2 w = osrc[(long)jc * stride];   // v conj(w)
                v = make_float2(fmaf(v.x, w.x, v.y * w.y), fmaf(v.y, w.x, -v.x * w.y));
            }
        }
        if (!jin) v = make_float2(0.0f, 0.0f);
        float2 t = s_root[idx];
        if (!rowok) t = make_float2(0.0f, 0.0f);
        are = __builtin_amdgcn_mfma_f32_32x32x2f32(t.x, v.x, are, 0, 0, 0);
        are = __builtin_amdgcn_mfma_f32_32x32x2f32(-t.y, v.y, are, 0, 0, 0);
        aim = __builtin_amdgcn_mfma_f32_32x32x2f32(t.x, v.y, aim, 0, 0, 0);
        aim = __builtin_amdgcn_mfma_f32_32x32x2f32(t.y, v.x, aim, 0, 0, 0);
        // two frequencies on: j -> j + 1 -> j + 2 (the step from half - 1 to half is the wrap)
        idx += (j + 1 == half) ? wrap : step;
        idx -= idx >= D ? D : 0;
        idx += (j + 2 == half) ? wrap : step;
        idx -= idx >= D ? D : 0;
    }
    if (!live) return;
    // C/D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float2 *dst = a.out + (long)item * a.outer * a.m * a.inner + ((long)o * a.m) * a.inner + i;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int rr = row0 + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (rr < a.m) dst[(long)rr * a.inner] = make_float2(are[q] * a.scale, aim[q] * a.scale);
    }
}

__global__ void mc_fill_kernel(float *p, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// pos[r] = r for r < n (the forward DFT's positions)
__global__ void mc_iota_kernel(int *pos, int n) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) pos[r] = r;
}

// python's a[lo:hi] on an axis of length n -> [lo', hi')
__device__ __forceinline__ void py_slice(long lo, bool lo_none, long hi, bool hi_none, int n, int &a, int &b) {
    long s = lo_none ? 0 : (lo < 0 ? lo + n : lo), e = hi_none ? n : (hi < 0 ? hi + n : hi);
    s = s < 0 ? 0 : (s > n ? n : s), e = e < 0 ? 0 : (e > n ? n : e);
    a = (int)s, b = (int)e;
}

// The indices of an axis the reference leaves non-zero before its argmax (register_translation_3d :727-747), ascending,
// padded with INT_MIN; win[item][axis][MC_MW].  rigid == nullptr: the max_shifts window (same for every item; launched
// for item 0 only), else the window [ceil(rigid - dev), floor(rigid + dev)) of the item's frame (:1569-1572).
__global__ void mc_window_kernel(int *win, int items, int nbox, const float *rigid, int dev, const int *max_shifts, McBoxes bx) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= items * 3) return;
    const int item = id / 3, d = id - 3 * item;
    const int n = bx.n[d];
    int za[2] = {0, 0}, zb[2] = {0, 0};     // up to two zeroed ranges
    if (rigid) {
        const float r = rigid[(long)(item / nbox) * 3 + d];
        const long lb = (long)ceilf(r - (float)dev), ub = (long)floorf(r + (float)dev);
        if (lb < 0 && ub >= 0) {
            py_slice(ub, false, lb, false, n, za[0], zb[0]);
        } else {
            py_slice(0, true, lb, false, n, za[0], zb[0]);
            py_slice(ub, false, 0, true, n, za[1], zb[1]);
        }
    } else {
        const long m = max_shifts[d];
        py_slice(m, false, -m, false, n, za[0], zb[0]);     // (-0 is 0: nothing is zeroed for max_shifts = 0)
    }
    int *w = win + ((long)item * 3 + d) * MC_MW;
    int cnt = 0;
    for (int i = 0; i < n && cnt < MC_MW; ++i) {
        const bool zeroed = (i >= za[0] && i < zb[0]) || (i >= za[1] && i < zb[1]);
        if (!zeroed) w[cnt++] = i;
    }
    for (; cnt < MC_MW; ++cnt) w[cnt] = INT_MIN;
}

// First largest |cc| of an (m0, m1, m2) complex array in C order, restricted to rows that are in use.  One block per
// item.  STAGE 0: cc over the window `win` -> peak index per axis, and the numerators of the 15 fractional positions
// around it for the upsampled pass (register_translation_3d :757-768: offset = dftshift - shift uf).  STAGE 1: cc over
// those -> the shift (:771-781), written with the signs `sgn` (total_shifts :1596 is (-x, -y, +z); the rigid shift is
// kept as it is).
template <int STAGE>
__global__ __launch_bounds__(256) void mc_peak_kernel(const float2 *cc, int m, const int *win, int win_item_stride, int *peak,
                                                      int *pos_up, int uf, int region, McBoxes bx, float *shifts, float3 sgn) {
    const int item = blockIdx.x;
    const float2 *c = cc + (long)item * m * m * m;
    const int *w = win + (long)item * win_item_stride;
    float best = -1.0f;
    int besti = INT_MAX;
    for (int e = threadIdx.x; e < m * m * m; e += 256) {
        const int i0 = e / (m * m), i1 = (e / m) % m, i2 = e % m;
        if (STAGE == 0 && (w[i0] == INT_MIN || w[MC_MW + i1] == INT_MIN || w[2 * MC_MW + i2] == INT_MIN)) continue;
        if (STAGE == 1 && (i0 >= region || i1 >= region || i2 >= region)) continue;
        const float2 v = c[e];
        const float a = fmaf(v.x, v.x, v.y * v.y);
        if (a > best || (a == best && e < besti)) best = a, besti = e;   // (e ascending per thread: only the first test fires)
    }
    __shared__ float sb[256];
    __shared__ int si[256];
    sb[threadIdx.x] = best, si[threadIdx.x] = besti;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float ob = sb[threadIdx.x + s];
            const int oi = si[threadIdx.x + s];
            if (ob > sb[threadIdx.x] || (ob == sb[threadIdx.x] && oi < si[threadIdx.x])) sb[threadIdx.x] = ob, si[threadIdx.x] = oi;
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    int e = si[0];
    // nothing kept, or every kept value is zero (NaN compares false: also lands here): numpy's argmax of the masked array
    // is then its first zero -- index (0,0,0) for an all-zero window; a NaN correlation has no meaningful peak either
    int i[3] = {0, 0, 0};
    const bool found = e != INT_MAX && sb[0] > 0.0f;
    if (e != INT_MAX) i[0] = e / (m * m), i[1] = (e / m) % m, i[2] = e % m;
    if (STAGE == 0) {
        for (int d = 0; d < 3; ++d) {
            const int n = bx.n[d];
            int ix = found ? w[d * MC_MW + i[d]] : 0;
            const int s = ix > n / 2 ? ix - n : ix;
            peak[item * 3 + d] = s;
            for (int r = 0; r < region; ++r) pos_up[((long)item * 3 + d) * MC_MW + r] = s * uf + (r - region / 2);
        }
    } else {
        const float sg[3] = {sgn.x, sgn.y, sgn.z};
        for (int d = 0; d < 3; ++d) {
            const float s = (float)peak[item * 3 + d] + (float)(i[d] - region / 2) / (float)uf;
            shifts[(long)item * 3 + d] = bx.n[d] == 1 ? 0.0f : sg[d] * s;
        }
    }
}

// MotionCorrect.apply_shifts_points :351-371: one thread per point; nearest patch centre (float64 distances, first
// minimum like cdist(...).argmin(0)), then P_T[k,:,t] = p -+ (shift[t] - shift[0]) with the reference's signs.
__global__ void mc_apply_points_kernel(const float *points, int K, const float *shifts, int T, int NP, const float *centers,
                                       float *out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const double p[3] = {points[3 * k], points[3 * k + 1], points[3 * k + 2]};
    double best = 1e300;
    int bi = 0;
    for (int q = 0; q < NP; ++q) {
        const double dx = centers[3 * q] - p[0], dy = centers[3 * q + 1] - p[1], dz = centers[3 * q + 2] - p[2];
        const double d = sqrt(dx * dx + dy * dy + dz * dz);
        if (d < best) best = d, bi = q;
    }
    const float *s0 = shifts + (long)bi * 3;
    for (int t = 0; t < T; ++t) {
        const float *st = shifts + ((long)t * NP + bi) * 3;
        out[((long)k * 3 + 0) * T + t] = (float)p[0] - st[0] + s0[0];
        out[((long)k * 3 + 1) * T + t] = (float)p[1] - st[1] + s0[1];
        out[((long)k * 3 + 2) * T + t] = (float)p[2] + st[2] - s0[2];
    }
}

static int mc_starts(int size, int overlap, int stride, int *out) {   // sliding_window_3d :1207-1213
    int cnt = 0;
    const int w = overlap + stride;
    for (int s = 0; s < size - w; s += stride) {
        if (out) out[cnt] = s;
        ++cnt;
    }
    if (out) out[cnt] = size - w;
    return cnt + 1;
}

static size_t mc_round(size_t b) { return (b + 255) / 256 * 256; }

// One registration pass over `items` boxes: spectra -> window -> peak -> upsampled -> shifts.
struct McPlan {
    McBoxes bx;
    long boxvox;          // n0 n1 n2
    int items;
};

static bool mc_use_mfma() {
    const char *e = getenv("DNMF_K8_VALU");     // 1: the vector-ALU kernel (the checker of the MFMA one); read at every call
    return !(e && e[0] == '1');
}

static void mc_launch_axis(McAxis a, int items, size_t maxD_bytes, hipStream_t st) {
    const long ncol = (long)a.outer * a.inner;
    if (mc_use_mfma()) {
        dim3 grid((unsigned)((ncol + 63) / 64), (unsigned)((a.m + 63) / 64), (unsigned)items);
        hipLaunchKernelGGL(mc_axis_mfma_kernel, grid, dim3(256), (size_t)a.n * a.uf * sizeof(float2), st, a);
        return;
    }
    dim3 grid((unsigned)((ncol + MC_TC - 1) / MC_TC), (unsigned)((a.m + MC_RTILE - 1) / MC_RTILE), (unsigned)items);
    hipLaunchKernelGGL(mc_axis_kernel, grid, dim3(256), (size_t)a.n * a.uf * sizeof(float2), st, a);
}

// numerators of the positions r + s_d of a frame's shifted grid: pos[(item 3 + d) nmax + r] = r uf + round(s_d uf)
__global__ void mc_shift_pos_kernel(const float *shifts, int uf, McBoxes bx, int nmax, int *pos) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x, item = blockIdx.y;
    if (e >= 3 * nmax) return;
    const int d = e / nmax, r = e - d * nmax;
    const int sn = (int)lrintf(shifts[(long)item * 3 + d] * (float)uf);
    pos[((long)item * 3 + d) * nmax + r] = r < bx.n[d] ? r * uf + sn : INT_MIN;
}

// smallest finite real part of every frame of (nf, P) complex values -> fmin[f] (initialised to +inf by the caller): the
// np.nanmin of border_nan='min' (:1121)
__global__ __launch_bounds__(256) void mc_frame_min_kernel(const float2 *img, long P, float *fmin) {
    const int f = blockIdx.y;
    float m = __builtin_inff();
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < P; g += (long)gridDim.x * 256) m = fminf(m, img[(long)f * P + g].x);
    __shared__ float sm[256];
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] = fminf(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    // (float order = order of the bit patterns for values of one sign; both signs: two atomics on the two orders)
    if (threadIdx.x == 0) {
        const float v = sm[0];
        if (v >= 0.0f)
            atomicMin(reinterpret_cast<int *>(fmin + f), __float_as_int(v));
        else
            atomicMax(reinterpret_cast<unsigned *>(fmin + f), __float_as_uint(v));
    }
}

// apply_shifts_dft :1098-1145 and tile_and_correct_3d :1573: real part of the shifted frames minus add_to_movie, and the
// border where the shift brought in voxels from the other side -- with the reference's pairing: the border of axis 0
// follows the shift of axis 1 and the other way round (:1083 swaps the first two shifts for the frequencies, :1104-1113 use the
// swapped pair on the axes in order).  border: 0 leave, 1 NaN (border_nan=True), 2 the frame's smallest value ('min'),
// 3 the nearest row / column / slice inside ('copy': the reference copies axis by axis, which composes to clamping each
// index).  Written to `corrected` and / or added into the per-voxel sums and counts of the finite values (the nanmean of
// tile_and_correct_wrapper :2057).  The factor exp(i diffphase) of :1097 is left out: diffphase is the argument of the
// correlation's peak value, zero up to rounding for real images (1e-8 here), and it multiplies a real image.
__global__ __launch_bounds__(256) void mc_shifted_frames_kernel(const float2 *img, int nf, int X, int Y, int Z, const float *shifts,
                                                                float add, int border, const float *fmin, float *corrected, long ldc,
                                                                float *tsum, int *tcount) {
    const long P = (long)X * Y * Z;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P) return;
    const int z = (int)(g % Z), y = (int)((g / Z) % Y), x = (int)(g / ((long)Y * Z));
    float acc = 0.0f;
    int cnt = 0;
    for (int f = 0; f < nf; ++f) {
        float v = img[(long)f * P + g].x - add;
        if (border) {
            const float s0 = shifts[3 * f], s1 = shifts[3 * f + 1], s2 = shifts[3 * f + 2];
            const int max_h = (int)ceilf(fmaxf(0.0f, s1)), min_h = (int)floorf(fminf(0.0f, s1));
            const int max_w = (int)ceilf(fmaxf(0.0f, s0)), min_w = (int)floorf(fminf(0.0f, s0));
            const int max_d = (int)ceilf(fmaxf(0.0f, s2)), min_d = (int)floorf(fminf(0.0f, s2));
            // python: a[:max_h] and, for min_h < 0, a[min_h:] (slices clip to the axis)
            const bool out = x < max_h || (min_h < 0 && x >= max(X + min_h, 0)) || y < max_w || (min_w < 0 && y >= max(Y + min_w, 0)) ||
                             z < max_d || (min_d < 0 && z >= max(Z + min_d, 0));
            if (out && border == 1) v = __builtin_nanf("");
            if (out && border == 2) v = fmin[f] - add;
            if (out && border == 3) {
                const int xs = min(max(x, max_h), X + min_h - 1), ys = min(max(y, max_w), Y + min_w - 1),
                          zs = min(max(z, max_d), Z + min_d - 1);
                v = img[(long)f * P + ((long)min(max(xs, 0), X - 1) * Y + min(max(ys, 0), Y - 1)) * Z + min(max(zs, 0), Z - 1)].x - add;
            }
        }
        if (corrected) corrected[(long)f * ldc + g] = v;
        if (v == v) acc += v, ++cnt;
    }
    if (tsum) tsum[g] += acc, tcount[g] += cnt;
}

// The transforms of one call: buffers carved from its workspace, and the passes built from mc_axis_kernel.
struct McRun {
    hipStream_t st;
    int X, Y, Z, uf, region, max_dev;
    float add;
    float2 *bufA, *bufB, *inv1, *inv2, *cc;
    int *win, *pos_up, *peak, *iota, *mshift;

    // forward DFT of `nitems` boxes of frames [first, ...) (or of the template when ld == 0): voxels -> spectrum in A
    void forward(const McBoxes &bx, const float *src, long ld, const int *ids, int first, int nitems, float2 *A, float2 *Bf) const {
        McAxis a{};
        a.bx = bx, a.X = X, a.Y = Y, a.Z = Z, a.add = add, a.first_frame = first;
        a.uf = 1, a.sign = -1, a.scale = 1.0f, a.pos = iota, a.pos_item_stride = 0, a.pos_off = 0;
        // z: voxels -> A
        a.mode = 0, a.frames = src, a.ldf = ld, a.frame_ids = ids;
        a.outer = bx.n[0] * bx.n[1], a.n = bx.n[2], a.inner = 1, a.m = bx.n[2], a.out = A;
        mc_launch_axis(a, nitems, 0, st);
        // y: A -> B
        a.mode = 1, a.in = A, a.outer = bx.n[0], a.n = bx.n[1], a.inner = bx.n[2], a.m = bx.n[1], a.out = Bf;
        mc_launch_axis(a, nitems, 0, st);
        // x: B -> A
        a.in = Bf, a.outer = 1, a.n = bx.n[0], a.inner = bx.n[1] * bx.n[2], a.m = bx.n[0], a.out = A;
        mc_launch_axis(a, nitems, 0, st);
    }
    // P = spec conj(tspec) onto mm positions per axis (numerators in posbuf (items or 1, 3, MC_MW), over ufac) -> cc (items, mm^3)
    void inverse(const McBoxes &bx, const float2 *spec, const float2 *tspec, const int *posbuf, int pstride, int mm, int ufac,
                 int nitems) const {
        McAxis a{};
        a.bx = bx, a.uf = ufac, a.sign = 1, a.pos = posbuf, a.pos_item_stride = pstride, a.m = mm;
        // x (with the product): (1, n0, n1 n2) -> (1, mm, n1 n2)
        a.mode = 2, a.in = spec, a.other = tspec, a.outer = 1, a.n = bx.n[0], a.inner = bx.n[1] * bx.n[2], a.pos_off = 0;
        a.scale = 1.0f / (float)bx.n[0], a.out = inv1;
        mc_launch_axis(a, nitems, 0, st);
        // y: (mm, n1, n2) -> (mm, mm, n2)
        a.mode = 1, a.in = inv1, a.outer = mm, a.n = bx.n[1], a.inner = bx.n[2], a.pos_off = MC_MW;
        a.scale = 1.0f / (float)bx.n[1], a.out = inv2;
        mc_launch_axis(a, nitems, 0, st);
        // z: (mm mm, n2, 1) -> (mm mm, mm, 1)
        a.in = inv2, a.outer = mm * mm, a.n = bx.n[2], a.inner = 1, a.pos_off = 2 * MC_MW;
        a.scale = 1.0f / (float)bx.n[2], a.out = cc;
        mc_launch_axis(a, nitems, 0, st);
    }
    // one registration pass over `nitems` boxes: window -> peak -> upsampled -> shifts
    void registration(const McBoxes &bx, const float2 *spec, const float2 *tspec, const float *rigid, int first, int nitems,
                      float *shifts_out, float3 sgn) const {
        const int witems = rigid ? nitems : 1;
        hipLaunchKernelGGL(mc_window_kernel, dim3((unsigned)((witems * 3 + 63) / 64)), dim3(64), 0, st, win, witems, bx.nbox,
                           rigid ? rigid + (long)first * 3 : nullptr, max_dev, mshift, bx);
        inverse(bx, spec, tspec, win, rigid ? 3 * MC_MW : 0, MC_MW, 1, nitems);
        hipLaunchKernelGGL((mc_peak_kernel<0>), dim3((unsigned)nitems), dim3(256), 0, st, cc, MC_MW, win, rigid ? 3 * MC_MW : 0, peak,
                           pos_up, uf, region, bx, (float *)nullptr, sgn);
        inverse(bx, spec, tspec, pos_up, 3 * MC_MW, region, uf, nitems);
        hipLaunchKernelGGL((mc_peak_kernel<1>), dim3((unsigned)nitems), dim3(256), 0, st, cc, region, win, 0, peak, pos_up, uf, region,
                           bx, shifts_out, sgn);
    }
};

}  // namespace dnmf

extern "C" {

int dnmf_register_patches_grid(int X, int Y, int Z, const int *strides, const int *overlaps, int *dims, int *starts) {
    using namespace dnmf;
    if (!strides || !overlaps || X <= 0 || Y <= 0 || Z <= 0) return 0;
    const int S[3] = {X, Y, Z};
    int cnt[3];
    for (int d = 0; d < 3; ++d) {
        if (strides[d] <= 0 || overlaps[d] < 0 || strides[d] + overlaps[d] > S[d]) return 0;
        cnt[d] = mc_starts(S[d], overlaps[d], strides[d], nullptr);
    }
    if (dims) dims[0] = cnt[0], dims[1] = cnt[1], dims[2] = cnt[2];
    if (starts) {   // (NP,3) in the reference's iteration order: x outermost
        int sx[4096], sy[4096], sz[4096];
        if (cnt[0] > 4096 || cnt[1] > 4096 || cnt[2] > 4096) return 0;
        mc_starts(X, overlaps[0], strides[0], sx), mc_starts(Y, overlaps[1], strides[1], sy), mc_starts(Z, overlaps[2], strides[2], sz);
        int q = 0;
        for (int i = 0; i < cnt[0]; ++i)
            for (int j = 0; j < cnt[1]; ++j)
                for (int k = 0; k < cnt[2]; ++k, ++q) starts[3 * q] = sx[i], starts[3 * q + 1] = sy[j], starts[3 * q + 2] = sz[k];
    }
    return cnt[0] * cnt[1] * cnt[2];
}

// frames per chunk so that one spectrum buffer stays below 256 MB
static int mc_chunk(long vox_per_frame, int B, int NP) {
    long c = (256L << 20) / (8 * vox_per_frame);
    if (c > 65535 / NP) c = 65535 / NP;     // (frame, patch) items ride on gridDim.z
    if (c < 1) c = 1;
    if (c > B) c = B;
    return (int)c;
}

size_t dnmf_register_patches_workspace(int X, int Y, int Z, const int *strides, const int *overlaps, int B) {
    using namespace dnmf;
    int dims[3];
    const int NP = dnmf_register_patches_grid(X, Y, Z, strides, overlaps, dims, nullptr);
    if (NP <= 0 || B <= 0) return 0;
    const long P = (long)X * Y * Z;
    const long w[3] = {strides[0] + overlaps[0], strides[1] + overlaps[1], strides[2] + overlaps[2]};
    const long pvox = (long)NP * w[0] * w[1] * w[2];
    const long per_frame = P > pvox ? P : pvox;
    const int Bc = mc_chunk(per_frame, B, NP);
    const long items = (long)Bc * NP;
    const long biggest_n12 = (long)Y * Z > w[1] * w[2] ? (long)Y * Z : w[1] * w[2];
    size_t b = 0;
    b += 2 * mc_round((size_t)Bc * per_frame * sizeof(float2));                     // spectra, ping-pong
    b += mc_round((size_t)P * sizeof(float2)) + mc_round((size_t)pvox * sizeof(float2)); // template spectra
    b += 2 * mc_round((size_t)items * MC_MW * biggest_n12 * sizeof(float2));         // partial inverses (two stages)
    b += mc_round((size_t)items * MC_MW * MC_MW * MC_MW * sizeof(float2));           // cc on the window / the upsampled grid
    b += 2 * mc_round((size_t)items * 3 * MC_MW * sizeof(int));                      // window indices, upsampled positions
    b += mc_round((size_t)items * 3 * sizeof(int)) + mc_round((size_t)16384 * sizeof(int)) + mc_round((size_t)(NP + 1) * 3 * sizeof(int)) +
         mc_round(64);
    return b;
}

int dnmf_register_patches(const float *frames, long ldf, const int *frame_ids, int B, const float *tmpl, int X, int Y, int Z,
                          const int *strides, const int *overlaps, const int *max_shifts, int max_deviation_rigid,
                          int upsample_factor, float add_to_movie, float *rigid_shifts, float *patch_shifts, void *workspace,
                          size_t workspace_bytes, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(frames && tmpl && strides && overlaps && max_shifts && rigid_shifts && patch_shifts && workspace, DNMF_E_NULL,
                 "dnmf_register_patches: NULL argument");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && B > 0 && ldf >= (long)X * Y * Z, DNMF_E_SHAPE,
                 "dnmf_register_patches: X=%d Y=%d Z=%d B=%d ldf=%ld", X, Y, Z, B, ldf);
    int dims[3];
    const int NP = dnmf_register_patches_grid(X, Y, Z, strides, overlaps, dims, nullptr);
    DNMF_REQUIRE(NP > 0 && NP <= 65535, DNMF_E_SHAPE, "dnmf_register_patches: strides + overlaps must fit the volume (%d patches)", NP);
    DNMF_REQUIRE(upsample_factor >= 1 && (int)((upsample_factor * 3 + 1) / 2) <= MC_MW, DNMF_E_UNSUPPORTED,
                 "dnmf_register_patches: upsample_factor %d (ceil(1.5 factor) <= %d)", upsample_factor, MC_MW);
    const int S[3] = {X, Y, Z};
    const int w[3] = {strides[0] + overlaps[0], strides[1] + overlaps[1], strides[2] + overlaps[2]};
    for (int d = 0; d < 3; ++d) {
        DNMF_REQUIRE(S[d] * upsample_factor <= MC_MAXD, DNMF_E_UNSUPPORTED,
                     "dnmf_register_patches: axis %d of %d voxels x upsample factor %d > %d", d, S[d], upsample_factor, MC_MAXD);
        DNMF_REQUIRE(max_shifts[d] >= 0 && (S[d] <= MC_MW || 2 * max_shifts[d] <= MC_MW) && max_deviation_rigid >= 1 &&
                         (w[d] <= MC_MW || 2 * max_deviation_rigid + 1 <= MC_MW),
                     DNMF_E_UNSUPPORTED, "dnmf_register_patches: search window of axis %d wider than %d shifts", d, MC_MW);
    }
    DNMF_REQUIRE(workspace_bytes >= dnmf_register_patches_workspace(X, Y, Z, strides, overlaps, B), DNMF_E_WORKSPACE,
                 "dnmf_register_patches: workspace %zu < %zu bytes", workspace_bytes,
                 dnmf_register_patches_workspace(X, Y, Z, strides, overlaps, B));
    McRun r{};
    r.st = (hipStream_t)stream, r.X = X, r.Y = Y, r.Z = Z, r.add = add_to_movie, r.uf = upsample_factor;
    r.region = (upsample_factor * 3 + 1) / 2, r.max_dev = max_deviation_rigid;   // ceil(1.5 uf)
    const long P = (long)X * Y * Z;
    const long pvox = (long)NP * w[0] * w[1] * w[2];
    const long per_frame = P > pvox ? P : pvox;
    const int Bc = mc_chunk(per_frame, B, NP);
    const long items_max = (long)Bc * NP;
    const long biggest_n12 = (long)Y * Z > (long)w[1] * w[2] ? (long)Y * Z : (long)w[1] * w[2];
    // carve the workspace
    char *at = static_cast<char *>(workspace);
    auto take = [&](size_t bytes) { char *p = at; at += mc_round(bytes); return p; };
    r.bufA = reinterpret_cast<float2 *>(take((size_t)Bc * per_frame * sizeof(float2)));
    r.bufB = reinterpret_cast<float2 *>(take((size_t)Bc * per_frame * sizeof(float2)));
    float2 *tF_full = reinterpret_cast<float2 *>(take((size_t)P * sizeof(float2)));
    float2 *tF_patch = reinterpret_cast<float2 *>(take((size_t)pvox * sizeof(float2)));
    r.inv1 = reinterpret_cast<float2 *>(take((size_t)items_max * MC_MW * biggest_n12 * sizeof(float2)));
    r.inv2 = reinterpret_cast<float2 *>(take((size_t)items_max * MC_MW * biggest_n12 * sizeof(float2)));
    r.cc = reinterpret_cast<float2 *>(take((size_t)items_max * MC_MW * MC_MW * MC_MW * sizeof(float2)));
    r.win = reinterpret_cast<int *>(take((size_t)items_max * 3 * MC_MW * sizeof(int)));
    r.pos_up = reinterpret_cast<int *>(take((size_t)items_max * 3 * MC_MW * sizeof(int)));
    r.peak = reinterpret_cast<int *>(take((size_t)items_max * 3 * sizeof(int)));
    r.iota = reinterpret_cast<int *>(take((size_t)16384 * sizeof(int)));
    int *starts = reinterpret_cast<int *>(take((size_t)(NP + 1) * 3 * sizeof(int)));
    r.mshift = reinterpret_cast<int *>(take(64));

    // patch starts (the whole-volume box is entry NP), max_shifts, the forward DFT's positions: small host -> device copies
    {
        int *h = static_cast<int *>(malloc((size_t)(NP + 1) * 3 * sizeof(int)));
        DNMF_REQUIRE(h, DNMF_E_WORKSPACE, "dnmf_register_patches: out of host memory");
        dnmf_register_patches_grid(X, Y, Z, strides, overlaps, dims, h);
        h[3 * NP] = h[3 * NP + 1] = h[3 * NP + 2] = 0;
        hipError_t e = hipMemcpyAsync(starts, h, (size_t)(NP + 1) * 3 * sizeof(int), hipMemcpyHostToDevice, r.st);
        if (e == hipSuccess) e = hipMemcpyAsync(r.mshift, max_shifts, 3 * sizeof(int), hipMemcpyHostToDevice, r.st);
        if (e == hipSuccess) e = hipStreamSynchronize(r.st);   // h is freed below; the call is not on a hot path
        free(h);
        DNMF_REQUIRE(e == hipSuccess, (int)e, "dnmf_register_patches: copy of the patch grid: %s", hipGetErrorString(e));
    }
    int nmax = X > Y ? X : Y;
    nmax = nmax > Z ? nmax : Z;
    hipLaunchKernelGGL(mc_iota_kernel, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, r.st, r.iota, nmax);

    McBoxes full;
    full.n[0] = X, full.n[1] = Y, full.n[2] = Z, full.nbox = 1, full.start = starts + 3 * NP;
    McBoxes pat;
    pat.n[0] = w[0], pat.n[1] = w[1], pat.n[2] = w[2], pat.nbox = NP, pat.start = starts;

    // template spectra (once)
    r.forward(full, tmpl, 0, nullptr, 0, 1, tF_full, r.bufB);
    r.forward(pat, tmpl, 0, nullptr, 0, NP, tF_patch, r.bufB);
    for (int f0 = 0; f0 < B; f0 += Bc) {
        const int nf = B - f0 < Bc ? B - f0 : Bc;
        // rigid shift of every frame of the chunk (:1559-1560), kept with its own sign
        r.forward(full, frames, ldf, frame_ids, f0, nf, r.bufA, r.bufB);
        r.registration(full, r.bufA, tF_full, nullptr, f0, nf, rigid_shifts + (long)f0 * 3, make_float3(1.0f, 1.0f, 1.0f));
        // one shift per patch inside the window around the frame's rigid shift (:1562-1583), signs of :1596
        r.forward(pat, frames, ldf, frame_ids, f0, nf * NP, r.bufA, r.bufB);
        r.registration(pat, r.bufA, tF_patch, rigid_shifts, f0, nf * NP, patch_shifts + (long)f0 * NP * 3,
                       make_float3(-1.0f, -1.0f, 1.0f));
    }
    return check_launch("dnmf_register_patches");
}

// frames per chunk of the rigid correction: two complex volumes per frame below 256 MB each
static int mc_rigid_chunk(long P, int B) {
    long c = (256L << 20) / (8 * P);
    if (c > 65535) c = 65535;
    if (c < 1) c = 1;
    if (c > B) c = B;
    return (int)c;
}

size_t dnmf_rigid_correct_workspace(int X, int Y, int Z, int B) {
    using namespace dnmf;
    if (X <= 0 || Y <= 0 || Z <= 0 || B <= 0) return 0;
    const long P = (long)X * Y * Z;
    const int Bc = mc_rigid_chunk(P, B);
    int nmax = X > Y ? X : Y;
    nmax = nmax > Z ? nmax : Z;
    size_t b = 0;
    b += 2 * mc_round((size_t)Bc * P * sizeof(float2)) + mc_round((size_t)P * sizeof(float2));
    b += 2 * mc_round((size_t)Bc * MC_MW * Y * Z * sizeof(float2));
    b += mc_round((size_t)Bc * MC_MW * MC_MW * MC_MW * sizeof(float2));
    b += 2 * mc_round((size_t)Bc * 3 * MC_MW * sizeof(int)) + mc_round((size_t)Bc * 3 * sizeof(int));
    b += mc_round((size_t)16384 * sizeof(int)) + mc_round(64) + mc_round(64) + mc_round((size_t)Bc * 3 * nmax * sizeof(int));
    b += mc_round((size_t)Bc * sizeof(float));
    return b;
}

int dnmf_rigid_correct(const float *frames, long ldf, const int *frame_ids, int B, const float *tmpl, int X, int Y, int Z,
                       const int *max_shifts, int upsample_factor, float add_to_movie, int border_nan, float *rigid_shifts,
                       float *corrected, long ldc, float *tsum, int *tcount, void *workspace, size_t workspace_bytes,
                       dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(frames && tmpl && max_shifts && rigid_shifts && workspace, DNMF_E_NULL, "dnmf_rigid_correct: NULL argument");
    DNMF_REQUIRE((tsum == nullptr) == (tcount == nullptr), DNMF_E_NULL, "dnmf_rigid_correct: tsum and tcount go together");
    const long P = (long)X * Y * Z;
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && B > 0 && ldf >= P && (!corrected || ldc >= P), DNMF_E_SHAPE,
                 "dnmf_rigid_correct: X=%d Y=%d Z=%d B=%d ldf=%ld ldc=%ld", X, Y, Z, B, ldf, ldc);
    DNMF_REQUIRE(upsample_factor >= 1 && (int)((upsample_factor * 3 + 1) / 2) <= MC_MW, DNMF_E_UNSUPPORTED,
                 "dnmf_rigid_correct: upsample_factor %d (ceil(1.5 factor) <= %d)", upsample_factor, MC_MW);
    DNMF_REQUIRE(border_nan >= 0 && border_nan <= 3, DNMF_E_UNSUPPORTED, "dnmf_rigid_correct: border_nan %d (0 .. 3)", border_nan);
    const int S[3] = {X, Y, Z};
    for (int d = 0; d < 3; ++d) {
        DNMF_REQUIRE(S[d] * upsample_factor <= MC_MAXD, DNMF_E_UNSUPPORTED,
                     "dnmf_rigid_correct: axis %d of %d voxels x upsample factor %d > %d", d, S[d], upsample_factor, MC_MAXD);
        DNMF_REQUIRE(max_shifts[d] >= 0 && (S[d] <= MC_MW || 2 * max_shifts[d] <= MC_MW), DNMF_E_UNSUPPORTED,
                     "dnmf_rigid_correct: search window of axis %d wider than %d shifts", d, MC_MW);
    }
    DNMF_REQUIRE(workspace_bytes >= dnmf_rigid_correct_workspace(X, Y, Z, B), DNMF_E_WORKSPACE,
                 "dnmf_rigid_correct: workspace %zu < %zu bytes", workspace_bytes, dnmf_rigid_correct_workspace(X, Y, Z, B));
    McRun r{};
    r.st = (hipStream_t)stream, r.X = X, r.Y = Y, r.Z = Z, r.add = add_to_movie, r.uf = upsample_factor;
    r.region = (upsample_factor * 3 + 1) / 2, r.max_dev = 0;
    const int Bc = mc_rigid_chunk(P, B);
    int nmax = X > Y ? X : Y;
    nmax = nmax > Z ? nmax : Z;
    char *at = static_cast<char *>(workspace);
    auto take = [&](size_t bytes) { char *p = at; at += mc_round(bytes); return p; };
    r.bufA = reinterpret_cast<float2 *>(take((size_t)Bc * P * sizeof(float2)));
    r.bufB = reinterpret_cast<float2 *>(take((size_t)Bc * P * sizeof(float2)));
    float2 *tF_full = reinterpret_cast<float2 *>(take((size_t)P * sizeof(float2)));
    r.inv1 = reinterpret_cast<float2 *>(take((size_t)Bc * MC_MW * Y * Z * sizeof(float2)));
    r.inv2 = reinterpret_cast<float2 *>(take((size_t)Bc * MC_MW * Y * Z * sizeof(float2)));
    r.cc = reinterpret_cast<float2 *>(take((size_t)Bc * MC_MW * MC_MW * MC_MW * sizeof(float2)));
    r.win = reinterpret_cast<int *>(take((size_t)Bc * 3 * MC_MW * sizeof(int)));
    r.pos_up = reinterpret_cast<int *>(take((size_t)Bc * 3 * MC_MW * sizeof(int)));
    r.peak = reinterpret_cast<int *>(take((size_t)Bc * 3 * sizeof(int)));
    r.iota = reinterpret_cast<int *>(take((size_t)16384 * sizeof(int)));
    int *starts = reinterpret_cast<int *>(take(64));
    r.mshift = reinterpret_cast<int *>(take(64));
    int *spos = reinterpret_cast<int *>(take((size_t)Bc * 3 * nmax * sizeof(int)));
    float *fmin = reinterpret_cast<float *>(take((size_t)Bc * sizeof(float)));
    {
        hipError_t e = hipMemsetAsync(starts, 0, 3 * sizeof(int), r.st);
        if (e == hipSuccess) e = hipMemcpyAsync(r.mshift, max_shifts, 3 * sizeof(int), hipMemcpyHostToDevice, r.st);
        DNMF_REQUIRE(e == hipSuccess, (int)e, "dnmf_rigid_correct: copy of max_shifts: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(mc_iota_kernel, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, r.st, r.iota, nmax);
    McBoxes full;
    full.n[0] = X, full.n[1] = Y, full.n[2] = Z, full.nbox = 1, full.start = starts;
    r.forward(full, tmpl, 0, nullptr, 0, 1, tF_full, r.bufB);
    for (int f0 = 0; f0 < B; f0 += Bc) {
        const int nf = B - f0 < Bc ? B - f0 : Bc;
        float *sh = rigid_shifts + (long)f0 * 3;
        r.forward(full, frames, ldf, frame_ids, f0, nf, r.bufA, r.bufB);
        r.registration(full, r.bufA, tF_full, nullptr, f0, nf, sh, make_float3(1.0f, 1.0f, 1.0f));
        if (!corrected && !tsum) continue;
        // apply_shifts_dft :1083-1097: the spectrum times exp(+2 pi i f s / n) per axis, inverse transform = the inverse
        // DFT evaluated at the positions r + s_d, numerators (r uf + s_d uf) over uf (s_d is a multiple of 1 / uf)
        hipLaunchKernelGGL(mc_shift_pos_kernel, dim3((unsigned)((3 * nmax + 255) / 256), (unsigned)nf), dim3(256), 0, r.st, sh, r.uf,
                           full, nmax, spos);
        McAxis a{};
        a.bx = full, a.uf = r.uf, a.sign = 1, a.pos = spos, a.pos_item_stride = 3 * nmax, a.mode = 1;
        a.in = r.bufA, a.outer = 1, a.n = X, a.inner = Y * Z, a.m = X, a.pos_off = 0, a.scale = 1.0f / (float)X, a.out = r.bufB;
        mc_launch_axis(a, nf, 0, r.st);
        a.in = r.bufB, a.outer = X, a.n = Y, a.inner = Z, a.m = Y, a.pos_off = nmax, a.scale = 1.0f / (float)Y, a.out = r.bufA;
        mc_launch_axis(a, nf, 0, r.st);
        a.in = r.bufA, a.outer = X * Y, a.n = Z, a.inner = 1, a.m = Z, a.pos_off = 2 * nmax, a.scale = 1.0f / (float)Z, a.out = r.bufB;
        mc_launch_axis(a, nf, 0, r.st);
        if (border_nan == 2) {
            hipLaunchKernelGGL(mc_fill_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, r.st, fmin, nf, __builtin_inff());
            const unsigned nb = (unsigned)(P / 4096 > 256 ? 256 : (P + 4095) / 4096);
            hipLaunchKernelGGL(mc_frame_min_kernel, dim3(nb, (unsigned)nf), dim3(256), 0, r.st, r.bufB, P, fmin);
        }
        hipLaunchKernelGGL(mc_shifted_frames_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, r.st, r.bufB, nf, X, Y, Z, sh,
                           add_to_movie, border_nan, fmin, corrected ? corrected + (long)f0 * ldc : nullptr, ldc, tsum, tcount);
    }
    return check_launch("dnmf_rigid_correct");
}

int dnmf_apply_shifts_points(const float *points, int K, const float *patch_shifts, int T, int NP, const float *centers, float *out,
                             dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(points && patch_shifts && centers && out, DNMF_E_NULL, "dnmf_apply_shifts_points: NULL argument");
    DNMF_REQUIRE(K > 0 && T > 0 && NP > 0, DNMF_E_SHAPE, "dnmf_apply_shifts_points: K=%d T=%d NP=%d", K, T, NP);
    hipLaunchKernelGGL(mc_apply_points_kernel, dim3((unsigned)((K + 63) / 64)), dim3(64), 0, (hipStream_t)stream, points, K,
                       patch_shifts, T, NP, centers, out);
    return check_launch("dnmf_apply_shifts_points");
}

}  // extern "C"
