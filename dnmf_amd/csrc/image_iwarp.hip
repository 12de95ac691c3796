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
// Round 3 sharpened the radius twice (both in image_iwarp_rows_kernel, which serves every volume with X, Y > 1): per-axis
// radii from the linear part of s for Z > 1 (iwarp_min_stretch), and a bound of |s(v*) - s(v)| by the angle between that
// vector and g - s(v), which settles the lattice points whose pre-image left the volume without a box.
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
// `an` (Z > 1 only; round 3) receives the constants of the ANISOTROPIC radii, all of the linear map M = D J(centre), D =
// diag(S/(S-1)): s(v*) - s(v) = M (v* - v) + E (v* - v) with |E|_F <= max(D) drift, hence |M (v* - v)| <= R (1 +
// max(D) drift / m) =: R' for R = d0 + rho.  With n the unit normal of the image of the (x, y) plane, mu = |n . M_z|,
// t = M_z - (n . M_z) n and m2 = the smallest singular value of [M_x M_y]:
//   |v*_z - v_z| <= R' / mu                                (the component of M (v* - v) along n is (n . M_z)(v*_z - v_z))
//   |(v* - v)_xy| <= (sqrt(R'^2 - (mu delta)^2) + |t| R' / mu) / m2      for delta <= |v*_z - v_z|
// and a lattice point is at least delta = |v_z - round(v_z)| away from v in z.  At the reference's depth (Z = 2, D_z = 2)
// the odd slice lies one voxel from both slice images (d0 = 1, the isotropic radius 1.02: a box of 18 voxels for every
// point), but mu delta = 1 as well: the (x, y) radius is ~0.2 and the box is the cell again.
// an = {mu (rounded down), |t| (up), 1 / m2 (up), 1 + max(D) drift / m (up)}; {0, 0, inf, inf} when there is none.
// an[4..14] (any Z): max(D) drift (up), |M|_F (up), M row by row (z row and column zero for Z == 1) -- see the rows kernel.
constexpr int IW_NK = 16;   // floats per frame: 1/m, then an[0..14]
__device__ double iwarp_min_stretch(const float *b, const Volume &vol, float *an = nullptr) {
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
    if (an) {
        an[0] = 0.0f, an[1] = 0.0f, an[2] = __builtin_inff(), an[3] = __builtin_inff();
        const double D[3] = {vol.X / (vol.X - 1.0), vol.Y / (vol.Y - 1.0), hz ? vol.Z / (vol.Z - 1.0) : 1.0};
        const double dmax = fmax(D[0], fmax(D[1], D[2]));
        double M[3][3], mf = 0;
        for (int d = 0; d < 3; ++d)
            for (int e = 0; e < 3; ++e) {
                M[d][e] = (d < nd && e < nd) ? D[d] * J[d][e] : 0.0;
                mf += M[d][e] * M[d][e];
                an[6 + 3 * d + e] = (float)M[d][e];
            }
        an[4] = (float)(dmax * drift) * 1.000001f + 1e-30f;
        an[5] = (float)sqrt(mf) * 1.000001f;
        if (!(an[4] < 1e6f && an[5] < 1e6f)) an[4] = an[5] = __builtin_inff();   // (NaN too)
        if (hz && m > 1e-3) {
            double n[3] = {M[1][0] * M[2][1] - M[2][0] * M[1][1], M[2][0] * M[0][1] - M[0][0] * M[2][1],
                           M[0][0] * M[1][1] - M[1][0] * M[0][1]};   // M_x x M_y
            const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            const double ga = M[0][0] * M[0][0] + M[1][0] * M[1][0] + M[2][0] * M[2][0],
                         gc = M[0][1] * M[0][1] + M[1][1] * M[1][1] + M[2][1] * M[2][1],
                         gb = M[0][0] * M[0][1] + M[1][0] * M[1][1] + M[2][0] * M[2][1];
            const double lmin = 0.5 * (ga + gc - sqrt((ga - gc) * (ga - gc) + 4.0 * gb * gb));
            if (nn > 1e-6 && lmin > 1e-6) {
                for (int d = 0; d < 3; ++d) n[d] /= nn;
                const double nz = n[0] * M[0][2] + n[1] * M[1][2] + n[2] * M[2][2];
                double t2 = 0;
                for (int d = 0; d < 3; ++d) {
                    const double t = M[d][2] - nz * n[d];
                    t2 += t * t;
                }
                const double vals[4] = {fabs(nz) * (1.0 - 1e-6), sqrt(t2) * (1.0 + 1e-6) + 1e-12, (1.0 + 1e-6) / sqrt(lmin),
                                        (1.0 + dmax * drift / m) * (1.0 + 1e-6)};
                bool ok = vals[0] > 1e-3;
                for (int i = 0; i < 4; ++i) ok = ok && vals[i] == vals[i] && vals[i] < 1e6;
                if (ok)
                    for (int i = 0; i < 4; ++i) an[i] = (float)vals[i] * (i == 0 ? 0.999999f : 1.000001f);
            }
        }
    }
    return m == m ? m : 0.0;   // NaN coefficients: no bound
}

// 1/m of every frame of the call, rounded up; +inf where there is no usable bound (every point of such a frame is
// then marked for the exhaustive search); then the constants of the sharper radii (IW_NK floats per frame in all)
__global__ void iwarp_stretch_kernel(const float *__restrict__ beta, int T, const int *__restrict__ times, int B, Volume vol,
                                     float *__restrict__ inv_stretch, unsigned *__restrict__ marked) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    marked[b] = 0;
    if (b == 0) marked[B] = 0;   // marked[B]: frames of this call with a marked point
    float bt[30];
    load_beta(beta, T, times[b], bt);
    float an[IW_NK - 1];
    const double m = iwarp_min_stretch(bt, vol, an);
    inv_stretch[IW_NK * b] = m > 1e-3 ? (float)(1.0 / m) * 1.000001f : __builtin_inff();
#pragma unroll
    for (int i = 0; i < IW_NK - 1; ++i) inv_stretch[IW_NK * b + 1 + i] = an[i];
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
    const float inv_m = inv_stretch[IW_NK * b];
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

// Round 3: the window search with a thread that walks IW_ROWS consecutive rows x at a fixed (y, z).  Same arithmetic and
// the same certificate as image_iwarp_window_kernel, organised so that the per-thread costs are shared: the thirty
// coefficients are loaded and doubled once per IW_ROWS lattice points; the pre-image of the next row starts from the last
// one moved by one row (one fixed-point step instead of two); and the cell of the next row usually sits on top of the last
// one, so half of its corner positions (two of four, four of eight) are already known.  The corners are ranked by their
// fp32 squared distances first: the float64 comparison of the reference (on the same fp32 positions) can only order two
// candidates differently from the fp32 ranking when their fp32 distances are within a few roundings of each other
// (relative 2e-6 here; the fp32 evaluation's error is below 4e-7), so float64 distances are formed only for the
// candidates within that margin of the smallest -- two of eight at the identity of a Z == 2 volume, where every lattice
// point of the odd slice is equidistant from both slices (:83 scales by sz, not sz - 1); lowest voxel index among equals.
// Points that need more than their cell (or a frame without a usable stretch bound) are handled exactly as in the window
// kernel: the box in float64, or a mark for the exhaustive kernel.
// An UPPER bound of sqrt(x) for the radii (x >= 0 or NaN): the hardware's square root (one instruction, 1 ulp) rounded up
// generously, instead of the correctly rounded sequence of ~15 instructions sqrtf compiles to; the absolute term covers
// arguments below the normal range, which the instruction may flush to zero.
__device__ __forceinline__ float iw_sqrt_up(float x) { return fmaf(__builtin_amdgcn_sqrtf(x), 1.000001f, 1e-18f); }

#ifndef DNMF_IW_ROWS
#define DNMF_IW_ROWS 16   // (8: 4.33 ms per 4000 frames of 512x512, 16: 4.22, 32: 4.12 -- and fewer blocks for small volumes)
#endif
constexpr int IW_ROWS = DNMF_IW_ROWS;
#ifndef DNMF_IW_LAG_Z
#define DNMF_IW_LAG_Z 1   // Z > 1 too: one evaluation of the continuous map per row (see the row loop)
#endif
#ifndef DNMF_IW_SHARE
#define DNMF_IW_SHARE 8
#endif
constexpr int IW_SHARE = DNMF_IW_SHARE;   // lanes of a wave whose boxes the wave searches together

template <bool HASZ, int FAST>
__global__ __launch_bounds__(256) void image_iwarp_rows_kernel(const float *__restrict__ frames, long ldf,
                                                               const int *__restrict__ frame_ids, Volume vol,
                                                               const float *__restrict__ beta, int T, const int *__restrict__ times,
                                                               const float *__restrict__ inv_stretch, IwarpScale sc,
                                                               float *__restrict__ out, long ldo, unsigned char *__restrict__ todo,
                                                               unsigned *__restrict__ marked, int nyb) {
    constexpr int ND = HASZ ? 3 : 2;
    constexpr int NC = HASZ ? 8 : 4;     // cell corners, index = 4 (x corner) + 2 (y corner) + (z corner) for Z > 1, 2 (x) + (y) else
    constexpr int NF = NC / 2;           // corners of one x-face
    const int b = blockIdx.y;
    const int yb = blockIdx.x % nyb, xb = blockIdx.x / nyb;
    const int YZ = vol.Y * vol.Z;
    const int u_raw = yb * 256 + threadIdx.x;      // position in the (y, z) plane
    const bool valid = u_raw < YZ;                 // (lanes past the plane compute the last position and store nothing: the
    const int u = valid ? u_raw : YZ - 1;          // wave stays whole for the shared box search below)
    const int gy = HASZ ? div_small(u, vol.Z, vol.rcp_z) : u, gz = HASZ ? u - gy * vol.Z : 0;
    const int x_first = xb * IW_ROWS, x_end = min(x_first + IW_ROWS, vol.X);
    const float inv_m = inv_stretch[IW_NK * b];
    const float *an = inv_stretch + IW_NK * b + 1;
    const float an_mu = HASZ ? an[0] : 0.0f, an_imu = HASZ ? 1.0f / an[0] : 0.0f, an_t = HASZ ? an[1] : 0.0f,
                an_im2 = HASZ ? an[2] : 0.0f, an_df = HASZ ? an[3] : 0.0f;

    auto mark = [&](unsigned char *flag) {
        *flag = 1;
        const unsigned long long mm = __ballot(1);
        if ((int)(threadIdx.x & 63) == __builtin_ctzll(mm) && atomicAdd(&marked[b], (unsigned)__builtin_popcountll(mm)) == 0)
            atomicAdd(&marked[gridDim.y], 1u);   // (gridDim.y = the frames of the call)
    };
    if (!(inv_m < 1e3f)) {   // (block-uniform)
        if (valid)
            for (int gx = x_first; gx < x_end; ++gx) mark(todo + (long)b * vol.P + (unsigned)(gx * YZ + u));
        return;
    }
    // (voxel indices are 32-bit here: the host sends volumes of 2^29 voxels or more to the one-point-per-thread kernel)
    const float *y = frames + (long)(frame_ids ? frame_ids[b] : b) * ldf;
    float *out_b = out + (long)b * ldo;
    unsigned char *todo_b = todo + (long)b * vol.P;
    float bt[30], b2[30];
    load_beta(beta, T, times[b], bt);
    double_beta(bt, b2);
    const float k[3] = {sc.k[0], sc.k[1], sc.k[2]}, hk[3] = {sc.hk[0], sc.hk[1], sc.hk[2]};
    const int S[3] = {vol.X, vol.Y, vol.Z};
    const int lane = (int)(threadIdx.x & 63);
    const float eps = 1e-4f + 1e-5f * (float)max(vol.X, max(vol.Y, vol.Z));
    const float tq = (((1.0f - 1e-3f) / inv_m - 3.0f * eps) / 1.00001f) * 0.99999f;   // (d0 + rho < tq: radius < 1)
    float v[3] = {(float)x_first * k[0], (float)gy * k[1], HASZ ? (float)gz * k[2] : 0.0f};
    // corner positions kept from the last row: the upper x-face of its cell (voxel row kept_x, columns kept_y / + 1, slices
    // kept_z / + 1)
    int kept_x = -1000, kept_y = -1000, kept_z = -1000;
    float kept[NF][3];
    for (int gx = x_first; gx < x_end; ++gx) {
        const unsigned g = (unsigned)(gx * YZ + u);
        unsigned char *flag = todo_b + g;
        const float gf[3] = {(float)gx, (float)gy, (float)gz};
        float res[3] = {0.0f, 0.0f, 0.0f};
        auto residual = [&]() {
            const Monomials<HASZ> mo = monomials<HASZ>(v[0], v[1], v[2]);
#pragma unroll
            for (int d = 0; d < ND; ++d) res[d] = gf[d] - poly_a<HASZ>(b2, d, mo) * hk[d];
        };
        // The estimate of the pre-image: fixed-point steps v <- v + (g - s(v)) k, two from g k for the first row of the walk.
        // The two-evaluation walk (round 3's first form): every other row takes one step from the last row's estimate moved by
        // a row, then measures what is left at the clamped point.  The walk used now: every other row starts from the last row's
        // clamped point vc moved by ITS residual and by one row -- one evaluation of s per row: the residual measured here both certifies
        // this row (rho) and corrects the next one; the one-row lag leaves |J - I| of a voxel in rho (0.004 for the warps of a
        // fit), which the radius absorbs: 5.56 -> 5.37 ms per 4000 frames.  (For Z > 1 the same change first took the kernel
        // from 162 to 179 registers, three waves per SIMD to two; compiled without the SLP vectoriser -- build.py -- the kernel
        // needs 120 and the walk is the same at every depth: 5.45 -> 5.23 ms per 1000 frames of 512x512x2.  DNMF_IW_LAG_Z=0
        // keeps the two-evaluation walk for Z > 1.)
        float vc[3] = {0.0f, 0.0f, 0.0f};
        constexpr bool TWO_EVALS = HASZ && !DNMF_IW_LAG_Z;
        if (TWO_EVALS || gx == x_first) {
            const int steps = gx == x_first ? 2 : 1;
            for (int it = 0; it < steps; ++it) {
                residual();
#pragma unroll
                for (int d = 0; d < ND; ++d) v[d] += res[d] * k[d];
            }
        }
        {
            float keep[3] = {v[0], v[1], v[2]};
#pragma unroll
            for (int d = 0; d < ND; ++d) vc[d] = __builtin_amdgcn_fmed3f(v[d], 0.0f, (float)(S[d] - 1)), v[d] = vc[d];   // (one instruction)
            residual();   // at the clamped point (the stretch bound holds between points of the volume)
            if (TWO_EVALS) {
                v[0] = keep[0] + k[0], v[1] = keep[1], v[2] = keep[2];    // the next row starts here
            } else {
#pragma unroll
                for (int d = 0; d < ND; ++d) v[d] = vc[d] + res[d] * k[d];
                v[0] += k[0];
            }
        }
        int c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
#pragma unroll
        for (int d = 0; d < ND; ++d) c0[d] = min((int)vc[d], max(S[d] - 2, 0)), c1[d] = min(c0[d] + 1, S[d] - 1);
        // the corners: the lower x-face from the last row's upper one when they are the same voxels
        float pos[NC][3];
        const bool reuse = kept_x == c0[0] && kept_y == c0[1] && kept_z == c0[2];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int yy = (j & (HASZ ? 2 : 1)) ? c1[1] : c0[1], zz = (HASZ && (j & 1)) ? c1[2] : c0[2];
            if (reuse) {
#pragma unroll
                for (int d = 0; d < 3; ++d) pos[j][d] = kept[j][d];
            } else {
                iwarp_position_t<HASZ, FAST>(bt, vol, c0[0], yy, zz, pos[j][0], pos[j][1], pos[j][2]);
            }
            iwarp_position_t<HASZ, FAST>(bt, vol, c1[0], yy, zz, pos[NF + j][0], pos[NF + j][1], pos[NF + j][2]);
#pragma unroll
            for (int d = 0; d < 3; ++d) kept[j][d] = pos[NF + j][d];
        }
        kept_x = c1[0], kept_y = c0[1], kept_z = c0[2];
        auto corner_index = [&](int i) {
            const int x = (i >= NF) ? c1[0] : c0[0], yy = (i & (HASZ ? 2 : 1)) ? c1[1] : c0[1], zz = (HASZ && (i & 1)) ? c1[2] : c0[2];
            return (x * vol.Y + yy) * vol.Z + zz;
        };
        float dist[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const float dx = pos[i][0] - gf[0], dy = pos[i][1] - gf[1], dz = HASZ ? pos[i][2] - gf[2] : 0.0f;
            dist[i] = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
        }
        // the smallest fp32 distance and its corner, without branches (selects); then: is any other corner within the margin?
        float d1 = dist[0];
        int wi = 0;
#pragma unroll
        for (int i = 1; i < NC; ++i) {
            const bool lt = dist[i] < d1;
            d1 = lt ? dist[i] : d1, wi = lt ? i : wi;
        }
        const float lim = d1 * 1.000002f + 1e-30f;
        int nclose = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i) nclose += dist[i] <= lim ? 1 : 0;
        double best = 1e300;
        int arg = 0;
        if (nclose == 1) {   // the usual case: the fp32 ranking decides
            const int wx = wi >= NF ? c1[0] : c0[0], wy = (wi & (HASZ ? 2 : 1)) ? c1[1] : c0[1], wz = (HASZ && (wi & 1)) ? c1[2] : c0[2];
            arg = (wx * vol.Y + wy) * vol.Z + wz;
        } else {   // float64 among the candidates within the margin; lowest voxel index among equals.  (NaN distances:
                   // nclose == 0, best stays huge and the radius test below marks the point)
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                if (!(dist[i] <= lim)) continue;
                const double dx = (double)pos[i][0] - gx, dy = (double)pos[i][1] - gy, dz = HASZ ? (double)pos[i][2] - gz : 0.0;
                const double d = dx * dx + dy * dy + dz * dz;
                const int idx = corner_index(i);
                if (d < best || (d == best && idx < arg)) best = d, arg = idx;
            }
        }
        const float bestf = nclose == 1 ? d1 : (float)best;   // (the float64 value itself is formed where a box needs it)
        // R bounds |a|, a = s(v*) - s(v) for the nearest voxel v*: by the triangle inequality d0 + rho; and (round 3) by the
        // angle between a and c = g - s(v):  |a - c| <= d0  gives  |a|^2 <= d0^2 - rho^2 + 2 a.c,  and with a = (M + E)(v* - v),
        // |E|_F <= drift_s, |v* - v| <= |a| / m:  a.c <= |a| sigma,  sigma = (|w+| + drift_s rho) / m,  where w = M^T c and
        // w+ keeps of w what can have the sign of v* - v: all of a free axis, nothing of the outward half of an axis on which
        // v sits on a face of the volume.  Hence |a| <= sigma + sqrt(sigma^2 + d0^2 - rho^2).  For a pre-image inside the
        // volume this is d0 instead of d0 + rho (rho ~ 0); for a lattice point whose pre-image LEFT the volume (c points
        // outwards, rho = how far, d0 barely larger: the nearest voxel sits next to v on the face) it is ~1/2 voxel where the
        // triangle inequality gives 2 rho -- boxes of (4 rho + 1)^2 voxels along every edge the warp moves inwards.
        int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        bool inside = true, far = false;
        // Z == 1: the usual case settled without a square root -- a radius below 1 is the cell whatever the point's place
        // in it, and  r < 1  <=  d0 + rho < tq  <=  best < (tq - (|c_x| + |c_y|))^2  (tq: per frame, set before the walk)
        bool quick = false;
        if (!HASZ) {
            const float t = tq - (fabsf(res[0]) + fabsf(res[1]));
            quick = t > 0.0f && bestf < t * t * 0.999999f;   // (NaN: false)
        }
        if (!quick) {
        if (nclose == 1) best = (double)d1;
        const float rho = iw_sqrt_up(res[0] * res[0] + res[1] * res[1] + res[2] * res[2]);   // (upper bounds both)
        const float d0f = iw_sqrt_up(bestf);
        auto window = [&](float R) {   // the radii for |a| <= R, the voxels within them, whether those are the cell's
            const float r = R * inv_m + 1e-3f;
            far = !(r < (float)IW_RMAX);   // also NaN: for the exhaustive kernel
            float rad[3] = {r, r, r};
            if (HASZ) {   // the anisotropic radii (iwarp_min_stretch); never larger than the isotropic one
                const float Rp = R * an_df, md = an_mu * fabsf(vc[2] - rintf(vc[2]));
                const float inplane = iw_sqrt_up(fmaxf((Rp - md) * (Rp + md), 0.0f)) * 1.00001f;
                const float rz = Rp * an_imu * 1.00001f;
                const float rxy = (inplane + an_t * rz) * an_im2 * 1.00001f + 1e-3f;
                rad[0] = rad[1] = fminf(r, rxy);   // (NaN or inf constants: fminf keeps r)
                rad[2] = fminf(r, rz + 1e-3f);
            }
            inside = true;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                lo[d] = max((int)ceilf(vc[d] - rad[d]), 0), hi[d] = min((int)floorf(vc[d] + rad[d]), S[d] - 1);
                inside = inside && lo[d] >= c0[d] && hi[d] <= c1[d];
            }
        };
        const float R1 = (d0f + rho) * 1.00001f + 3.0f * eps;
        window(R1);
        if (far || !inside) {   // (few points: the sharper bound costs nothing where the first one already says "the cell")
            float wp = 0.0f;
#pragma unroll
            for (int e = 0; e < ND; ++e) {
                float w = 0.0f;
#pragma unroll
                for (int d = 0; d < ND; ++d) w = fmaf(an[6 + 3 * d + e], res[d], w);
                const bool lower = !(vc[e] > 0.0f), upper = !(vc[e] < (float)(S[e] - 1));
                const float wv = lower ? (upper ? 0.0f : fmaxf(w, 0.0f)) : (upper ? fmaxf(-w, 0.0f) : fabsf(w));
                wp = fmaf(wv, wv, wp);
            }
            const float drift_s = an[4], m_fro = an[5];
            const float sigma = ((iw_sqrt_up(wp) + 2.0f * m_fro * eps) * 1.0001f + drift_s * (rho + eps)) * inv_m * 1.00001f;
            const float D0 = d0f * 1.00001f + 2.0f * eps, rlo = fmaxf(rho * 0.99999f - eps, 0.0f);   // (0.99999: also below the true rho)
            const float R2 = (sigma + iw_sqrt_up(fmaf(sigma, sigma, fmaxf((D0 - rlo) * (D0 + rlo), 0.0f)))) * 1.00001f + eps;
            if (R2 < R1) window(R2);   // (NaN: the first window stays)
        }
        }
        if (far && valid) mark(flag);
#ifdef DNMF_IW_NOBOX   // timing study only (wrong results): what the kernel costs without the box
        inside = true;
#endif
        const bool box = valid && !far && !inside;
#ifdef DNMF_IW_COUNTBOX   // study only: the count of the call = points that searched a box
        if (box) atomicAdd(&marked[b], 1u);
#endif
        // The box (a radius >= 1: it contains the cell).  Boxes are rare -- lattice points whose pre-image left the volume,
        // a strip of them along an edge under a shift -- and large there (rho of a few voxels: 100-250 candidates), and one
        // lane walking such a box alone holds its wave for ten times the wave's own work: at 512x512x2 under shifts of one
        // voxel 0.3 % of the points doubled the kernel's time.  So the WAVE searches the box of each of its lanes in turn,
        // candidate c of the box by lane c mod 64, in float64 on the fp32 positions (the reference's comparison), the
        // result a lexicographic minimum of (distance, voxel index) over the lanes.  When more than IW_SHARE lanes of a wave
        // need a box (strong warps everywhere) each lane walks its own, ranked in fp32 first.
        unsigned long long need = __ballot(box);
        if (__builtin_popcountll(need) <= IW_SHARE) {
            while (need) {
                const int src = __builtin_ctzll(need);
                need &= need - 1;
                const int l0 = __builtin_amdgcn_readlane(lo[0], src), l1 = __builtin_amdgcn_readlane(lo[1], src),
                          l2 = __builtin_amdgcn_readlane(lo[2], src);
                const int n1 = __builtin_amdgcn_readlane(hi[1], src) - l1 + 1, n2 = __builtin_amdgcn_readlane(hi[2], src) - l2 + 1;
                const int n = (__builtin_amdgcn_readlane(hi[0], src) - l0 + 1) * n1 * n2;
                const int sy = __builtin_amdgcn_readlane(gy, src), sz = __builtin_amdgcn_readlane(gz, src);
                double cb = 1e300;
                int ca = 0;
                for (int c = lane; c < n; c += 64) {
                    const int t = HASZ ? c / n2 : c, zz = HASZ ? l2 + (c - t * n2) : 0;
                    const int xo = t / n1, x = l0 + xo, yy = l1 + (t - xo * n1);
                    float px, py, pz;
                    iwarp_position_t<HASZ, FAST>(bt, vol, x, yy, zz, px, py, pz);
                    const double ex = (double)px - gx, ey = (double)py - sy, ez = HASZ ? (double)pz - sz : 0.0;
                    const double dd = ex * ex + ey * ey + ez * ez;
                    const int idx = (x * vol.Y + yy) * vol.Z + zz;
                    if (dd < cb || (dd == cb && idx < ca)) cb = dd, ca = idx;
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const double od = __shfl_xor(cb, off);
                    const int oa = __shfl_xor(ca, off);
                    if (od < cb || (od == cb && oa < ca)) cb = od, ca = oa;
                }
                if (lane == src) best = cb, arg = ca;
            }
        } else if (box) {
            // voxels beyond the cell are ranked in fp32 against the cell's result; a voxel takes part in the float64
            // comparison only when its fp32 distance is within the margin of the best so far
            float bf = (float)best;
            for (int x = lo[0]; x <= hi[0]; ++x)
                for (int yy = lo[1]; yy <= hi[1]; ++yy)
                    for (int zz = lo[2]; zz <= hi[2]; ++zz) {
                        if (x >= c0[0] && x <= c1[0] && yy >= c0[1] && yy <= c1[1] && zz >= c0[2] && zz <= c1[2]) continue;   // a corner
                        float sx, sy, sz;
                        iwarp_position_t<HASZ, FAST>(bt, vol, x, yy, zz, sx, sy, sz);
                        const float dx = sx - gf[0], dy = sy - gf[1], dz = HASZ ? sz - gf[2] : 0.0f;
                        const float d = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
                        if (!(d <= bf * 1.000002f + 1e-30f)) continue;
                        const double ex = (double)sx - gx, ey = (double)sy - gy, ez = HASZ ? (double)sz - gz : 0.0;
                        const double dd = ex * ex + ey * ey + ez * ez;
                        const int idx = (x * vol.Y + yy) * vol.Z + zz;
                        if (dd < best || (dd == best && idx < arg)) best = dd, arg = idx, bf = fminf(bf, d);
                    }
        }
        if (valid && !far) {
            out_b[g] = y[(unsigned)arg];
            *flag = 0;
        }
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

// one flag byte per lattice point and frame, then IW_NK floats and one counter per frame
static size_t iwarp_flag_bytes(int X, int Y, int Z, int B) { return ((size_t)X * Y * Z * B + 7) / 8 * 8; }

size_t dnmf_image_iwarp_workspace(int X, int Y, int Z, int B) {
    if (X <= 0 || Y <= 0 || Z <= 0 || B <= 0) return 0;
    return iwarp_flag_bytes(X, Y, Z, B) + (dnmf::IW_NK * sizeof(float) + sizeof(unsigned)) * (size_t)B + sizeof(unsigned);
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
    unsigned *marked = reinterpret_cast<unsigned *>(stretch + IW_NK * (size_t)B);   // IW_NK constants per frame
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
        const int nyb = (Y * Z + 255) / 256;
        const dim3 rgrid((unsigned)(nyb * ((X + IW_ROWS - 1) / IW_ROWS)), (unsigned)B);
#define DNMF_IW_ROWS_LAUNCH(HZ, FD)                                                                                              \
    hipLaunchKernelGGL((image_iwarp_rows_kernel<HZ, FD>), rgrid, dim3(256), 0, st, frames, ldf, frame_ids, vol, beta, T, times,  \
                       stretch, sc, out, ldo, todo, marked, nyb)
        if (X == 1 || Y == 1 || vol.P >= (1L << 29)) {   // an axis of one voxel (its hk is infinite: every point is marked), or voxel
                                                          // indices beyond the rows kernel's 32 bits: one point per thread
            if (Z > 1)
                hipLaunchKernelGGL(image_iwarp_window_kernel<true>, grid, dim3(256), 0, st, frames, ldf, frame_ids, vol, beta, T,
                                   times, stretch, sc, out, ldo, todo, marked);
            else
                hipLaunchKernelGGL(image_iwarp_window_kernel<false>, grid, dim3(256), 0, st, frames, ldf, frame_ids, vol, beta, T,
                                   times, stretch, sc, out, ldo, todo, marked);
        } else if (Z > 1) {
            if (vol.fastdiv) DNMF_IW_ROWS_LAUNCH(true, 1); else DNMF_IW_ROWS_LAUNCH(true, 0);
        } else {
            if (vol.fastdiv) DNMF_IW_ROWS_LAUNCH(false, 1); else DNMF_IW_ROWS_LAUNCH(false, 0);
        }
#undef DNMF_IW_ROWS_LAUNCH
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
