// K7 -- registered video Y_i: nearest-neighbour inverse warp of each frame.
//
// Reference: ExponentialFP.spatial_pushforward + image_iwarp, Demix/dNMF.py:81-83, 89-91, 95-103: the warped
// position of every voxel is un-normalised with sz[d] (not sz[d]-1), scipy's NearestNDInterpolator (cKDTree)
// is built on those positions, and every lattice point takes the frame value of its nearest warped voxel.
// Distances are evaluated in float64 on the float32 positions like the reference; ties (exactly equal distances) go
// to the lowest voxel index, cKDTree's choice there is unspecified.
//
// Search.  The warped position s(v) is a quadratic polynomial of the voxel v, so s(a) - s(b) = J((a+b)/2) (a - b)
// EXACTLY, J the Jacobian; with m a lower bound of the smallest singular value of J over the volume (one value per
// frame, iwarp_stretch_kernel), |s(a) - s(b)| >= m |a - b| for a, b inside the volume.  For a lattice point g the kernel
// takes a few fixed-point steps v <- v + (g - s(v)) (S-1)/S towards the pre-image of g, clamps v into the volume and
// measures what is left, rho = |s(v) - g|.  The voxels of the lattice cell around v are candidates; d0 = the smallest
// of their distances to g.  The nearest voxel v* has |s(v*) - g| <= d0, hence m |v* - v| <= |s(v*) - s(v)| <= d0 + rho:
// it lies in the box of radius r = (d0 + rho)/m around the CONTINUOUS point v -- for a near-identity warp r < 1 and the
// box is the cell just searched (4 candidates in 2-D, 8 in 3-D; the first version of this search centred the window on
// a voxel, radius 2 d0/m: 25 candidates).  Only when the box sticks out of the cell is it searched as a whole.  Lattice
// points whose box would exceed IW_RMAX -- or frames with m <= 0: a warp that folds -- are marked for the exhaustive
// kernel, the O(P) per point search of round 1.  Both kernels evaluate s(voxel) with the same fp32 sequence and
// compare the same float64 distances, lowest voxel index first among equals, so the result is the exhaustive search's
// bit for bit (the rounding of s in fp32 is covered by a slack in the radius).
#include "common.hpp"

namespace dnmf {

constexpr int IW_TILE = 1024;
constexpr int IW_RMAX = 64;   // largest window radius searched in place (129 x 129 (x Z) candidates: lattice points near the
                              // border whose pre-image lies ~25 voxels outside the volume; beyond that the exhaustive kernel)

// warped position of voxel (x,y,z) scaled by sz (not sz-1), fp32 like the reference (flow_ is a float32 tensor there)
template <bool HASZ, int FAST = -1>
__device__ __forceinline__ void iwarp_position_t(const float *bt, const Volume &vol, int x, int y, int z, float &sx,
                                                 float &sy, float &sz) {
    const float xf = (float)x, yf = (float)y, zf = HASZ ? (float)z : 0.0f;
    const float nx = grid_n<HASZ, FAST>(bt, vol, 0, xf, yf, zf);
    const float ny = grid_n<HASZ, FAST>(bt, vol, 1, xf, yf, zf);
    sx = __fmul_rn(__fmul_rn(__fadd_rn(nx, 1.0f), 0.5f), (float)vol.X);
    sy = __fmul_rn(__fmul_rn(__fadd_rn(ny, 1.0f), 0.5f), (float)vol.Y);
    sz = HASZ ? __fmul_rn(__fmul_rn(__fadd_rn(grid_n<HASZ, FAST>(bt, vol, 2, xf, yf, zf), 1.0f), 0.5f), (float)vol.Z) : 0.0f;
}
__device__ __forceinline__ void iwarp_position(const float *bt, const Volume &vol, int x, int y, int z, float &sx, float &sy,
                                               float &sz) {
    if (vol.Z > 1)
        iwarp_position_t<true>(bt, vol, x, y, z, sx, sy, sz);
    else
        iwarp_position_t<false>(bt, vol, x, y, z, sx, sy, sz);
}

// Lower bound of |q(a) - q(b)| / |a - b| over the volume for the quadratic map q = basis . beta (the un-scaled warp;
// s = q S/(S-1) stretches every axis by a factor >= 1, so the bound holds for s as well): smallest singular value of
// the Jacobian at the centre minus the largest change of the Jacobian over the volume (Frobenius norm).
__device__ double iwarp_min_stretch(const float *b, const Volume &vol) {
    const bool hz = vol.Z > 1;
    const int nd = hz ? 3 : 2;
    const double cx = 0.5 * (vol.X - 1), cy = 0.5 * (vol.Y - 1), cz = hz ? 0.5 * (vol.Z - 1) : 0.0;
    double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double hx = 0, hy = 0, hzz = 0;   // squared Frobenius norms of dJ/dx, dJ/dy, dJ/dz
    for (int d = 0; d < nd; ++d) {
        const double b1 = b[3 + d], b2 = b[6 + d], b3 = b[9 + d], b4 = b[12 + d], b5 = b[15 + d], b6 = b[18 + d],
                     b7 = b[21 + d], b8 = b[24 + d], b9 = b[27 + d];
        J[d][0] = b1 + 2 * b4 * cx + b7 * cy + b8 * cz;
        J[d][1] = b2 + 2 * b5 * cy + b7 * cx + b9 * cz;
        J[d][2] = b3 + 2 * b6 * cz + b8 * cx + b9 * cy;
        if (hz) {
            hx += 4 * b4 * b4 + b7 * b7 + b8 * b8, hy += b7 * b7 + 4 * b5 * b5 + b9 * b9, hzz += b8 * b8 + b9 * b9 + 4 * b6 * b6;
        } else {
            hx += 4 * b4 * b4 + b7 * b7, hy += b7 * b7 + 4 * b5 * b5;
        }
    }
    // two lower bounds of the smallest singular value at the centre: 1 - |J - I|_F (Weyl; tight for the near-identity
    // warps of a fit) and |det| over the product of the other singular values (any warp)
    double off = 0, fro = 0;
    for (int d = 0; d < nd; ++d)
        for (int e = 0; e < nd; ++e) {
            const double v = J[d][e] - (d == e ? 1.0 : 0.0);
            off += v * v, fro += J[d][e] * J[d][e];
        }
    double smin = 1.0 - sqrt(off);
    if (hz) {
        const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                           J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
        smin = fmax(smin, fabs(det) / (0.5 * fro));   // sigma_3 = |det| / (sigma_1 sigma_2), sigma_1 sigma_2 <= |J|_F^2 / 2
    } else {
        const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        const double disc = fmax(fro * fro - 4.0 * det * det, 0.0);
        smin = fmax(smin, sqrt(fmax(0.5 * (fro - sqrt(disc)), 0.0)) * (1.0 - 1e-9));   // exact for a 2 x 2 matrix
    }
    const double drift = cx * sqrt(hx) + cy * sqrt(hy) + cz * sqrt(hzz);
    const double m = 0.98 * (smin - drift);
    return m == m ? m : 0.0;   // NaN coefficients: no bound
}

// 1/m of every frame of the call, rounded up; +inf where there is no usable bound (every point of such a frame is
// then marked for the exhaustive search)
__global__ void iwarp_stretch_kernel(const float *__restrict__ beta, int T, const int *__restrict__ times, int B, Volume vol,
                                     float *__restrict__ inv_stretch, unsigned *__restrict__ marked) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    marked[b] = 0;
    if (b == 0) marked[B] = 0;   // marked[B]: frames of this call with a marked point
    float bt[30];
    load_beta(beta, T, times[b], bt);
    const double m = iwarp_min_stretch(bt, vol);
    inv_stretch[b] = m > 1e-3 ? (float)(1.0 / m) * 1.000001f : __builtin_inff();
}

struct IwarpScale {
    float k[3], hk[3];   // (S-1)/S and S/(2 (S-1)) per axis (1 and 1/2 along a pinned z)
};

// One thread per lattice point: pre-image estimate, its cell, the box if it is larger; todo[g] = 1 where the box would
// be too large.
template <bool HASZ>
__global__ __launch_bounds__(256) void image_iwarp_window_kernel(const float *__restrict__ frames, long ldf,
                                                                 const int *__restrict__ frame_ids, Volume vol,
                                                                 const float *__restrict__ beta, int T,
                                                                 const int *__restrict__ times,
                                                                 const float *__restrict__ inv_stretch, IwarpScale sc,
                                                                 float *__restrict__ out, long ldo,
                                                                 unsigned char *__restrict__ todo,
                                                                 unsigned *__restrict__ marked) {
    const int b = blockIdx.y;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= vol.P) return;
    unsigned char *flag = todo + (long)b * vol.P + g;
    // marked[b] counts the marked points of frame b (one atomic per wave that marks any): the exhaustive kernel leaves
    // a frame without marks after one scalar load
    auto mark = [&]() {
        *flag = 1;
        const unsigned long long mm = __ballot(1);
        if ((int)(threadIdx.x & 63) == __builtin_ctzll(mm) && atomicAdd(&marked[b], (unsigned)__builtin_popcountll(mm)) == 0)
            atomicAdd(&marked[gridDim.y], 1u);   // (gridDim.y = the frames of the call)
    };
    const float inv_m = inv_stretch[b];
    if (!(inv_m < 1e3f)) {
        mark();
        return;
    }
    const float *y = frames + (long)(frame_ids ? frame_ids[b] : b) * ldf;
    float bt[30], b2[30];
    load_beta(beta, T, times[b], bt);
    double_beta(bt, b2);
    int gx, gy, gz;
    voxel_xyz(g, vol, gx, gy, gz);
    constexpr int ND = HASZ ? 3 : 2;
    // s(v) for a continuous v: q S/(S-1) (dNMF.py:81-83); k = (S-1)/S.  (Any rounding here only moves the estimate v:
    // what is left of g - s(v) is measured below and enters the radius.)
    const float k[3] = {sc.k[0], sc.k[1], sc.k[2]}, hk[3] = {sc.hk[0], sc.hk[1], sc.hk[2]};
    const float gf[3] = {(float)gx, (float)gy, (float)gz};
    float v[3] = {gf[0] * k[0], gf[1] * k[1], HASZ ? gf[2] * k[2] : 0.0f};
    float res[3] = {0.0f, 0.0f, 0.0f};   // g - s(v)
    auto residual = [&]() {
        const Monomials<HASZ> mo = monomials<HASZ>(v[0], v[1], v[2]);
#pragma unroll
        for (int d = 0; d < ND; ++d) res[d] = gf[d] - poly_a<HASZ>(b2, d, mo) * hk[d];
    };
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        residual();
#pragma unroll
        for (int d = 0; d < ND; ++d) v[d] += res[d] * k[d];
    }
    // into the volume (the stretch bound holds between points of the volume); a NaN becomes 0 and ends in a NaN radius
    const int S[3] = {vol.X, vol.Y, vol.Z};
#pragma unroll
    for (int d = 0; d < ND; ++d) v[d] = fminf(fmaxf(v[d], 0.0f), (float)(S[d] - 1));
    residual();
    const float rho = sqrtf(res[0] * res[0] + res[1] * res[1] + res[2] * res[2]);
    // the cell of v
    int c0[3], c1[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        c0[d] = d < ND ? min((int)v[d], max(S[d] - 2, 0)) : 0;
        c1[d] = d < ND ? min(c0[d] + 1, S[d] - 1) : 0;
    }
    double best = 1e300;
    long arg = 0;
    auto candidate = [&](int x, int yy, int z) {
        float sx, sy, sz;
        iwarp_position_t<HASZ>(bt, vol, x, yy, z, sx, sy, sz);
        const double dx = (double)sx - gx, dy = (double)sy - gy, dz = (double)sz - gz;
        const double d = dx * dx + dy * dy + dz * dz;
        const long idx = ((long)x * vol.Y + yy) * vol.Z + z;
        if (d < best || (d == best && idx < arg)) best = d, arg = idx;   // ties: the lowest voxel index, as in the full search
    };
    // The cell's corners are ranked by their fp32 squared distances first (round 3).  The float64 comparison of the reference
    // (on the same fp32 positions) can only order two candidates differently from the fp32 ranking when their fp32 distances
    // are within a few roundings of each other (relative 2e-6 here; the fp32 evaluation's error is below 4e-7): only then
    // are the float64 distances formed -- e.g. at the identity of a Z > 1 volume, where every point has two equidistant
    // candidates (:83 scales by sz, not sz - 1).  (An axis of one voxel names its only candidate twice: harmless.)
    constexpr int NC = HASZ ? 8 : 4;
    float d1 = __builtin_inff(), d2 = __builtin_inff();   // smallest and second smallest fp32 squared distance
    int i1 = 0;
#pragma unroll
    for (int i = 0; i < (HASZ ? 0 : NC); ++i) {
        float sx, sy, sz;
        iwarp_position_t<HASZ>(bt, vol, (i & (HASZ ? 4 : 2)) ? c1[0] : c0[0], (i & (HASZ ? 2 : 1)) ? c1[1] : c0[1],
                               (HASZ && (i & 1)) ? c1[2] : c0[2], sx, sy, sz);
        const float dx = sx - gf[0], dy = sy - gf[1], dz = sz - gf[2];
        const float d = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
        const bool lt = d < d1;
        d2 = lt ? d1 : fminf(d2, d);
        i1 = lt ? i : i1;
        d1 = lt ? d : d1;
    }
    if (!HASZ && d2 > d1 * 1.000002f + 1e-30f) {   // a clear winner (Z > 1: fits start at the identity, where all points tie)
        const int x = (i1 & (HASZ ? 4 : 2)) ? c1[0] : c0[0], yy = (i1 & (HASZ ? 2 : 1)) ? c1[1] : c0[1], z = (HASZ && (i1 & 1)) ? c1[2] : c0[2];
        arg = ((long)x * vol.Y + yy) * vol.Z + z;
        best = (double)d1;
    } else {                              // near tie (or NaN): decide in float64, lowest voxel index among equals
#pragma unroll
        for (int i = 0; i < NC; ++i)
            candidate((i & (HASZ ? 4 : 2)) ? c1[0] : c0[0], (i & (HASZ ? 2 : 1)) ? c1[1] : c0[1], (HASZ && (i & 1)) ? c1[2] : c0[2]);
    }
    // fp32 rounding of the positions (lattice and continuous evaluation): a few units in the last place at magnitudes
    // up to the volume size; the radius itself is evaluated in fp32, rounded generously upwards
    const float eps = 1e-4f + 1e-5f * (float)max(vol.X, max(vol.Y, vol.Z));
    const float r = ((sqrtf((float)best) + rho) * 1.00001f + 3.0f * eps) * inv_m + 1e-3f;
    if (!(r < (float)IW_RMAX)) {   // also NaN
        mark();
        return;
    }
    int lo[3], hi[3];
    bool inside = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        lo[d] = d < ND ? max((int)ceilf(v[d] - r), 0) : 0;
        hi[d] = d < ND ? min((int)floorf(v[d] + r), S[d] - 1) : 0;
        inside = inside && lo[d] >= c0[d] && hi[d] <= c1[d];
    }
    if (!inside) {
        best = 1e300, arg = 0;   // the whole box in float64 (r >= 1 here: it contains the cell)
        for (int x = lo[0]; x <= hi[0]; ++x)
            for (int yy = lo[1]; yy <= hi[1]; ++yy)
                for (int z = lo[2]; z <= hi[2]; ++z) candidate(x, yy, z);
    }
    out[(long)b * ldo + g] = y[arg];
    *flag = 0;
}

// Z == 1, round 3: the window search with a thread that walks IW_ROWS consecutive rows x at a fixed y.  Same arithmetic and
// the same certificate as image_iwarp_window_kernel<false>, organised so that the per-thread costs are shared: the thirty
// coefficients are loaded and doubled once per IW_ROWS lattice points; the pre-image of the next row starts from the last
// one moved by one row (one fixed-point step instead of two); and the cell of the next row usually sits on top of the last
// one, so two of its four corner positions are already known.  Points that need more than their cell (or a frame without
// a usable stretch bound) are handled exactly as there: the box in float64, or a mark for the exhaustive kernel.
#ifndef DNMF_IW_ROWS
#define DNMF_IW_ROWS 8
#endif
constexpr int IW_ROWS = DNMF_IW_ROWS;

template <int FAST>
__global__ __launch_bounds__(256) void image_iwarp_rows_kernel(const float *__restrict__ frames, long ldf,
                                                               const int *__restrict__ frame_ids, Volume vol,
                                                               const float *__restrict__ beta, int T, const int *__restrict__ times,
                                                               const float *__restrict__ inv_stretch, IwarpScale sc,
                                                               float *__restrict__ out, long ldo, unsigned char *__restrict__ todo,
                                                               unsigned *__restrict__ marked, int nyb) {
    const int b = blockIdx.y;
    const int yb = blockIdx.x % nyb, xb = blockIdx.x / nyb;
    const int gy = yb * 256 + threadIdx.x;
    if (gy >= vol.Y) return;
    const int x_first = xb * IW_ROWS, x_end = min(x_first + IW_ROWS, vol.X);
    const float inv_m = inv_stretch[b];
    auto mark = [&](unsigned char *flag) {
        *flag = 1;
        const unsigned long long mm = __ballot(1);
        if ((int)(threadIdx.x & 63) == __builtin_ctzll(mm) && atomicAdd(&marked[b], (unsigned)__builtin_popcountll(mm)) == 0)
            atomicAdd(&marked[gridDim.y], 1u);   // (gridDim.y = the frames of the call)
    };
    if (!(inv_m < 1e3f)) {
        for (int gx = x_first; gx < x_end; ++gx) mark(todo + (long)b * vol.P + (long)gx * vol.Y + gy);
        return;
    }
    const float *y = frames + (long)(frame_ids ? frame_ids[b] : b) * ldf;
    float bt[30], b2[30];
    load_beta(beta, T, times[b], bt);
    double_beta(bt, b2);
    const float k[2] = {sc.k[0], sc.k[1]}, hk[2] = {sc.hk[0], sc.hk[1]};
    const float eps = 1e-4f + 1e-5f * (float)max(vol.X, vol.Y);
    const float gyf = (float)gy;
    float v[2] = {(float)x_first * k[0], gyf * k[1]};
    // corner positions kept from the last row: those of voxel row `kept_x` at columns kept_y, kept_y + 1
    int kept_x = -1000, kept_y = -1000;
    float kept[2][2] = {{0.f, 0.f}, {0.f, 0.f}};   // [column][sx, sy]
    for (int gx = x_first; gx < x_end; ++gx) {
        const long g = (long)gx * vol.Y + gy;
        unsigned char *flag = todo + (long)b * vol.P + g;
        const float gf[2] = {(float)gx, gyf};
        float res[2];
        auto residual = [&]() {
            const Monomials<false> mo = monomials<false>(v[0], v[1], 0.0f);
#pragma unroll
            for (int d = 0; d < 2; ++d) res[d] = gf[d] - poly_a<false>(b2, d, mo) * hk[d];
        };
        // fixed-point steps towards the pre-image: two from scratch, one from the last row's pre-image moved by a row
        const int steps = gx == x_first ? 2 : 1;
        for (int it = 0; it < steps; ++it) {
            residual();
            v[0] += res[0] * k[0], v[1] += res[1] * k[1];
        }
        float vc[2] = {fminf(fmaxf(v[0], 0.0f), (float)(vol.X - 1)), fminf(fmaxf(v[1], 0.0f), (float)(vol.Y - 1))};
        {
            const float keep0 = v[0], keep1 = v[1];
            v[0] = vc[0], v[1] = vc[1];
            residual();
            v[0] = keep0 + k[0], v[1] = keep1;    // the next row starts here
        }
        const float rho = sqrtf(res[0] * res[0] + res[1] * res[1]);
        const int c0x = min((int)vc[0], max(vol.X - 2, 0)), c1x = min(c0x + 1, vol.X - 1);
        const int c0y = min((int)vc[1], max(vol.Y - 2, 0)), c1y = min(c0y + 1, vol.Y - 1);
        // the four corners: row c0x from the last row's upper corners when they are the same voxels
        float px[2][2][2];   // [row][column][sx, sy]
        const bool reuse = kept_x == c0x && kept_y == c0y;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float sz;
            if (reuse)
                px[0][j][0] = kept[j][0], px[0][j][1] = kept[j][1];
            else
                iwarp_position_t<false, FAST>(bt, vol, c0x, j ? c1y : c0y, 0, px[0][j][0], px[0][j][1], sz);
            iwarp_position_t<false, FAST>(bt, vol, c1x, j ? c1y : c0y, 0, px[1][j][0], px[1][j][1], sz);
            kept[j][0] = px[1][j][0], kept[j][1] = px[1][j][1];
        }
        kept_x = c1x, kept_y = c0y;
        float d1 = __builtin_inff(), d2 = __builtin_inff();
        int i1 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dx = px[i >> 1][i & 1][0] - gf[0], dy = px[i >> 1][i & 1][1] - gf[1];
            const float d = fmaf(dx, dx, dy * dy);
            const bool lt = d < d1;
            d2 = lt ? d1 : fminf(d2, d);
            i1 = lt ? i : i1;
            d1 = lt ? d : d1;
        }
        double best = (double)d1;
        long arg = (long)((i1 >> 1) ? c1x : c0x) * vol.Y + ((i1 & 1) ? c1y : c0y);
        auto candidate = [&](int x, int yy) {
            float sx, sy, sz;
            iwarp_position_t<false, FAST>(bt, vol, x, yy, 0, sx, sy, sz);
            const double dx = (double)sx - gx, dy = (double)sy - gy;
            const double d = dx * dx + dy * dy;
            const long idx = (long)x * vol.Y + yy;
            if (d < best || (d == best && idx < arg)) best = d, arg = idx;
        };
        if (!(d2 > d1 * 1.000002f + 1e-30f)) {   // near tie (or NaN): the four in float64, lowest voxel index among equals
            best = 1e300, arg = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) candidate((i >> 1) ? c1x : c0x, (i & 1) ? c1y : c0y);
        }
        const float r = ((sqrtf((float)best) + rho) * 1.00001f + 3.0f * eps) * inv_m + 1e-3f;
        if (!(r < (float)IW_RMAX)) {   // also NaN
            mark(flag);
            continue;
        }
        const int lox = max((int)ceilf(vc[0] - r), 0), hix = min((int)floorf(vc[0] + r), vol.X - 1);
        const int loy = max((int)ceilf(vc[1] - r), 0), hiy = min((int)floorf(vc[1] + r), vol.Y - 1);
        if (!(lox >= c0x && hix <= c1x && loy >= c0y && hiy <= c1y)) {
            best = 1e300, arg = 0;   // the whole box in float64 (r >= 1 here: it contains the cell)
            for (int x = lox; x <= hix; ++x)
                for (int yy = loy; yy <= hiy; ++yy) candidate(x, yy);
        }
        out[(long)b * ldo + g] = y[arg];
        *flag = 0;
    }
}

// The exhaustive search (P candidates per lattice point, tiles of warped positions in LDS) for the marked points.
__global__ __launch_bounds__(256) void image_iwarp_full_kernel(const float *__restrict__ frames, long ldf,
                                                               const int *__restrict__ frame_ids, Volume vol,
                                                               const float *__restrict__ beta, int T,
                                                               const int *__restrict__ times, float *__restrict__ out,
                                                               long ldo, const unsigned char *__restrict__ todo,
                                                               const unsigned *__restrict__ marked, int B) {
    __shared__ float sx[IW_TILE], sy[IW_TILE], sz[IW_TILE];
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;  // lattice point served by this thread
    int gx = 0, gy = 0, gz = 0;
    if (g < vol.P) voxel_xyz(g, vol, gx, gy, gz);
    // the frames b = blockIdx.y, + gridDim.y, ...: a frame without marked points costs one scalar load (a block per frame
    // and tile that only looked at the counter was 1.1 ms of launches per 4000 frames at 512x512)
    if (marked && marked[B] == 0) return;         // nothing marked in the whole call (the usual case): one scalar load
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        if (marked && marked[b] == 0) continue;   // block-uniform
        const bool mine = g < vol.P && todo[(long)b * vol.P + g];
        if (!__syncthreads_or(mine)) continue;    // nothing marked in this block
        const float *y = frames + (long)(frame_ids ? frame_ids[b] : b) * ldf;
        float bt[30];
        load_beta(beta, T, times[b], bt);
        double best = 1e300;
        long arg = 0;
        for (long v0 = 0; v0 < vol.P; v0 += IW_TILE) {
            __syncthreads();
            for (int i = threadIdx.x; i < IW_TILE; i += blockDim.x) {
                const long v = v0 + i;
                if (v < vol.P) {
                    int x, yy, z;
                    voxel_xyz(v, vol, x, yy, z);
                    iwarp_position(bt, vol, x, yy, z, sx[i], sy[i], sz[i]);
                }
            }
            __syncthreads();
            if (mine) {
                const int n = (int)((vol.P - v0) < IW_TILE ? (vol.P - v0) : IW_TILE);
                for (int i = 0; i < n; ++i) {
                    const double dx = (double)sx[i] - gx, dy = (double)sy[i] - gy, dz = (double)sz[i] - gz;
                    const double d = dx * dx + dy * dy + dz * dz;
                    if (d < best) best = d, arg = v0 + i;
                }
            }
        }
        if (mine) out[(long)b * ldo + g] = y[arg];
        __syncthreads();   // the tiles in LDS are rewritten for the next frame
    }
}

__global__ void count_flags_kernel(const unsigned char *__restrict__ todo, long n, unsigned long long *__restrict__ count) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long m = __ballot(i < n && todo[i]);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__builtin_popcountll(m));
}

}  // namespace dnmf

extern "C" {

__global__ void sum_marked_kernel(const unsigned *__restrict__ marked, int B, unsigned long long *__restrict__ count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && marked[b]) atomicAdd(count, (unsigned long long)marked[b]);
}

// one flag byte per lattice point and frame, then one float and one counter per frame
static size_t iwarp_flag_bytes(int X, int Y, int Z, int B) { return ((size_t)X * Y * Z * B + 7) / 8 * 8; }

size_t dnmf_image_iwarp_workspace(int X, int Y, int Z, int B) {
    if (X <= 0 || Y <= 0 || Z <= 0 || B <= 0) return 0;
    return iwarp_flag_bytes(X, Y, Z, B) + (sizeof(float) + sizeof(unsigned)) * (size_t)B + sizeof(unsigned);
}

int dnmf_image_iwarp(const float *frames, long ldf, const int *frame_ids, int X, int Y, int Z, const float *beta, int T,
                     const int *times, int B, float *out, long ldo, void *workspace, size_t workspace_bytes, int exhaustive,
                     unsigned long long *fallback_count, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(frames && beta && times && out && workspace, DNMF_E_NULL, "dnmf_image_iwarp: NULL buffer");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && T > 0 && B > 0 && B <= 65535, DNMF_E_SHAPE,
                 "dnmf_image_iwarp: X=%d Y=%d Z=%d T=%d B=%d", X, Y, Z, T, B);
    const Volume vol = make_volume(X, Y, Z);
    DNMF_REQUIRE(ldf >= vol.P && ldo >= vol.P, DNMF_E_SHAPE, "dnmf_image_iwarp: ldf=%ld ldo=%ld < P=%ld", ldf, ldo, vol.P);
    DNMF_REQUIRE(workspace_bytes >= dnmf_image_iwarp_workspace(X, Y, Z, B), DNMF_E_WORKSPACE,
                 "dnmf_image_iwarp: workspace %zu < %zu bytes", workspace_bytes, dnmf_image_iwarp_workspace(X, Y, Z, B));
    hipStream_t st = (hipStream_t)stream;
    unsigned char *todo = static_cast<unsigned char *>(workspace);
    const dim3 grid((unsigned)((vol.P + 255) / 256), (unsigned)B);
    float *stretch = reinterpret_cast<float *>(todo + iwarp_flag_bytes(X, Y, Z, B));
    unsigned *marked = reinterpret_cast<unsigned *>(stretch + B);
    if (exhaustive) {
        marked = nullptr;
        hipError_t e = hipMemsetAsync(todo, 1, (size_t)vol.P * B, st);
        DNMF_REQUIRE(e == hipSuccess, (int)e, "dnmf_image_iwarp: hipMemsetAsync: %s", hipGetErrorString(e));
    } else {
        hipLaunchKernelGGL(iwarp_stretch_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, beta, T, times, B, vol, stretch,
                           marked);
        IwarpScale sc;
        const int S[3] = {X, Y, Z};
        for (int d = 0; d < 3; ++d) {
            const bool pinned = d == 2 && Z == 1;
            sc.k[d] = pinned ? 1.0f : (float)(S[d] - 1) / (float)S[d];
            sc.hk[d] = pinned ? 0.5f : 0.5f * (float)S[d] / (float)(S[d] - 1);   // an axis of one voxel: inf, and every point of
        }                                                                       // the volume goes to the exhaustive search
        if (Z > 1)
            hipLaunchKernelGGL(image_iwarp_window_kernel<true>, grid, dim3(256), 0, st, frames, ldf, frame_ids, vol, beta, T,
                               times, stretch, sc, out, ldo, todo, marked);
        else {
            const int nyb = (Y + 255) / 256;
            const dim3 rgrid((unsigned)(nyb * ((X + IW_ROWS - 1) / IW_ROWS)), (unsigned)B);
            if (vol.fastdiv)
                hipLaunchKernelGGL(image_iwarp_rows_kernel<1>, rgrid, dim3(256), 0, st, frames, ldf, frame_ids, vol, beta, T, times,
                                   stretch, sc, out, ldo, todo, marked, nyb);
            else
                hipLaunchKernelGGL(image_iwarp_rows_kernel<0>, rgrid, dim3(256), 0, st, frames, ldf, frame_ids, vol, beta, T, times,
                                   stretch, sc, out, ldo, todo, marked, nyb);
        }
    }
    if (fallback_count && marked) {
        hipLaunchKernelGGL(sum_marked_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, marked, B, fallback_count);
    } else if (fallback_count) {
        const long n = vol.P * B;
        hipLaunchKernelGGL(count_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, todo, n, fallback_count);
    }
    // every frame by its own blocks when all points are searched exhaustively, else 64 frames' worth of blocks that walk
    // over the frames and skip those without marks
    const dim3 full_grid(grid.x, exhaustive ? (unsigned)B : (unsigned)(B < 64 ? B : 64));
    hipLaunchKernelGGL(image_iwarp_full_kernel, full_grid, dim3(256), 0, st, frames, ldf, frame_ids, vol, beta, T, times, out,
                       ldo, todo, marked, B);
    return check_launch("dnmf_image_iwarp");
}

}  // extern "C"
