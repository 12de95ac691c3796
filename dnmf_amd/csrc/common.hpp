// Shared host/device helpers of libdnmf_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/dnmf_hip.h"

namespace dnmf {

// ---- error plumbing (thread-local text behind dnmf_last_error) ---------------------------------
char *last_error_buffer();
int fail(int code, const char *fmt, ...);
int check_launch(const char *what);

#define DNMF_REQUIRE(cond, code, ...)              \
    do {                                           \
        if (!(cond)) return ::dnmf::fail((code), __VA_ARGS__); \
    } while (0)

// ---- volume geometry -----------------------------------------------------------------------------
struct Volume {
    int X, Y, Z;
    long P;
    float sx1, sy1, sz1;  // (S-1) as fp32, the divisor of Demix/dNMF.py:55 (sz is int64 there)
    float rcp_yz, rcp_z;  // 1/(Y*Z), 1/Z for the voxel-index split
    float rx1, ry1, rz1;  // RN(1/(S-1)) for the division shortcut of normalise_fast
    float hx1, hy1, hz1;  // (S-1)/2 (exact) for unnormalise
    int fastdiv;          // every divided axis has 1 <= S-1 <= 65535: the shortcut is exact (tools/check_fastdiv.c)
};

inline Volume make_volume(int X, int Y, int Z) {
    Volume v;
    v.X = X, v.Y = Y, v.Z = Z;
    v.P = (long)X * Y * Z;
    v.sx1 = (float)(X - 1), v.sy1 = (float)(Y - 1), v.sz1 = (float)(Z - 1);
    v.rcp_yz = 1.0f / ((float)Y * (float)Z), v.rcp_z = 1.0f / (float)Z;
    v.rx1 = X > 1 ? 1.0f / v.sx1 : 0.0f, v.ry1 = Y > 1 ? 1.0f / v.sy1 : 0.0f, v.rz1 = Z > 1 ? 1.0f / v.sz1 : 0.0f;
    v.hx1 = 0.5f * v.sx1, v.hy1 = 0.5f * v.sy1, v.hz1 = 0.5f * v.sz1;
    v.fastdiv = X > 1 && X <= 65536 && Y > 1 && Y <= 65536 && Z <= 65536;  // Z == 1: z is never divided
    return v;
}

// One warped sample position: integer base corner, the two weights per axis and per-corner validity.
struct Sample {
    float ux, uy, uz;     // un-normalised source coordinates (voxel units)
    int x0, y0, z0;       // floor
    float wx0, wx1, wy0, wy1, wz0, wz1;  // weight of corner 0 / corner 1 along each axis
};

// The warped coordinate q_d = sum_a basis_a(x,y,z) * beta[a][d], basis order [1,x,y,z,x^2,y^2,z^2,xy,xz,yz]
// (Demix/dNMF.py:46-51,54).  The reference forms it with an fp32 einsum, i.e. a BLAS product over the ten basis
// terms; torch's CPU path accumulates them in index order with fused multiply-adds starting from zero, and the chain
// below is that sequence, bit for bit (checked on 10^5 voxels under random and under Adam-stepped coefficients: 100 %
// equal; a regrouping by powers of x, two FMAs per coordinate cheaper, agrees on 55 %).  That matters where a
// coordinate lands on a lattice point -- at the identity, and again on whole curves of voxels after the first Adam
// step, which moves every coefficient by exactly the learning rate: there floor() turns on the last bit.
//
// The monomials are fp32 products like the reference's basis tensor (dNMF.py:48-50); a thread that walks along x with
// (y,z) fixed keeps y^2, z^2, yz.  poly_a returns a_d = 2 q_d from DOUBLED coefficients (double_beta): scaling the
// coefficients by a power of two scales every partial sum exactly, and 2 q is what the normalisation needs
// (dNMF.py:55).  `b2` points at 30 floats laid out [a*3 + d].  HASZ = false is Z == 1 (z = 0: the four terms with z
// add an exact zero each for finite coefficients and are left out).
template <bool HASZ>
struct Monomials {
    float x, y, z, xx, yy, zz, xy, xz, yz;
};

template <bool HASZ>
__device__ __forceinline__ Monomials<HASZ> monomials(float x, float y, float z) {
    Monomials<HASZ> m;
    m.x = x, m.y = y, m.z = HASZ ? z : 0.0f;
    m.xx = __fmul_rn(x, x), m.yy = __fmul_rn(y, y), m.xy = __fmul_rn(x, y);
    m.zz = HASZ ? __fmul_rn(z, z) : 0.0f, m.xz = HASZ ? __fmul_rn(x, z) : 0.0f, m.yz = HASZ ? __fmul_rn(y, z) : 0.0f;
    return m;
}

__device__ __forceinline__ void double_beta(const float *b, float *b2) {
#pragma unroll
    for (int i = 0; i < 30; ++i) b2[i] = __fmul_rn(2.0f, b[i]);
}

template <bool HASZ>
__device__ __forceinline__ float poly_a(const float *b2, int d, const Monomials<HASZ> &m) {
    float q = b2[0 + d];                 // fma(1, b0, 0)
    q = fmaf(m.x, b2[3 + d], q);
    q = fmaf(m.y, b2[6 + d], q);
    if (HASZ) q = fmaf(m.z, b2[9 + d], q);
    q = fmaf(m.xx, b2[12 + d], q);
    q = fmaf(m.yy, b2[15 + d], q);
    if (HASZ) q = fmaf(m.zz, b2[18 + d], q);
    q = fmaf(m.xy, b2[21 + d], q);
    if (HASZ) {
        q = fmaf(m.xz, b2[24 + d], q);
        q = fmaf(m.yz, b2[27 + d], q);
    }
    return q;
}

// clamp into [0, n-1] in one instruction (v_med3_i32)
__device__ __forceinline__ int clamp_index(int i, int n) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(i), "v"(n - 1));
    return r;
}

// n = 2q/(S-1) - 1 exactly as the reference evaluates it in fp32 (Demix/dNMF.py:55): IEEE divide, IEEE subtract,
// no contraction.  `a` = 2q (poly_a).
__device__ __forceinline__ float normalise(float a, float sm1) { return __fsub_rn(__fdiv_rn(a, sm1), 1.0f); }

// The same value with the division by the constant S-1 done as a multiplication by r = RN(1/(S-1)) and one
// correction step: q0 = RN(a r), e = RN(a - q0 (S-1)) (exact), RN(q0 + e r).  For the integer divisors 1..65535 this
// is the correctly rounded quotient for every normal fp32 a -- checked exhaustively over all 2^23 significands per
// divisor by tools/check_fastdiv.c; both sequences scale with powers of two, and a quotient too small to be normal
// rounds to n = -1 either way.  Non-finite a gives a non-finite result in both (inf vs NaN), which every caller
// turns into zero weights.  Three instructions instead of the ~12 of the IEEE division sequence.
__device__ __forceinline__ float normalise_fast(float a, float sm1, float rcp) {
    const float q0 = __fmul_rn(a, rcp);
    const float e = __fmaf_rn(-q0, sm1, a);
    return __fsub_rn(__fmaf_rn(e, rcp, q0), 1.0f);
}

// n along axis d of the volume (0, 1, 2) from a = 2q, by the shortcut when the volume allows it.  FAST = 1 / 0: the
// caller has already branched on vol.fastdiv (kernel variants); -1: wave-uniform choice here.
template <int FAST = -1>
__device__ __forceinline__ float normalise_axis(float a, const Volume &vol, int d) {
    const float sm1 = d == 0 ? vol.sx1 : (d == 1 ? vol.sy1 : vol.sz1);
    const float rcp = d == 0 ? vol.rx1 : (d == 1 ? vol.ry1 : vol.rz1);
    if (FAST == 1) return normalise_fast(a, sm1, rcp);
    if (FAST == 0) return normalise(a, sm1);
    return vol.fastdiv ? normalise_fast(a, sm1, rcp) : normalise(a, sm1);
}

// n of voxel (x,y,z) along axis d, everything from scratch (kernels that do not walk along x); `b` = beta as stored
template <bool HASZ, int FAST = -1>
__device__ __forceinline__ float grid_n(const float *b, const Volume &vol, int d, float x, float y, float z) {
    const Monomials<HASZ> m = monomials<HASZ>(x, y, z);
    const float b2[10] = {__fmul_rn(2.0f, b[d]),      __fmul_rn(2.0f, b[3 + d]),  __fmul_rn(2.0f, b[6 + d]),  __fmul_rn(2.0f, b[9 + d]),
                          __fmul_rn(2.0f, b[12 + d]), __fmul_rn(2.0f, b[15 + d]), __fmul_rn(2.0f, b[18 + d]), __fmul_rn(2.0f, b[21 + d]),
                          __fmul_rn(2.0f, b[24 + d]), __fmul_rn(2.0f, b[27 + d])};
    float q = b2[0];
    q = fmaf(m.x, b2[1], q);
    q = fmaf(m.y, b2[2], q);
    if (HASZ) q = fmaf(m.z, b2[3], q);
    q = fmaf(m.xx, b2[4], q);
    q = fmaf(m.yy, b2[5], q);
    if (HASZ) q = fmaf(m.zz, b2[6], q);
    q = fmaf(m.xy, b2[7], q);
    if (HASZ) {
        q = fmaf(m.xz, b2[8], q);
        q = fmaf(m.yz, b2[9], q);
    }
    return normalise_axis<FAST>(q, vol, d);
}

// torch grid_sampler_unnormalize, align_corners=True: ((n + 1) / 2) * (S - 1) in fp32 without contraction, as
// RN(RN(n + 1) * h) with h = (S-1)/2: halving RN(n + 1) is exact (it is 0 or at least 2^-24 in magnitude) and so is
// halving the integer S-1, so the single rounded product equals the reference's two.  The round trip
// normalise -> unnormalise decides floor() for on-lattice points (hard part 1).
__device__ __forceinline__ float unnormalise(float n, float hsm1) { return __fmul_rn(__fadd_rn(n, 1.0f), hsm1); }

__device__ __forceinline__ void axis_weights(float u, int &i0, float &w0, float &w1) {
    const float f = floorf(u);
    // floorf of a huge / non-finite coordinate: clamp so the int conversion is defined; such samples
    // are out of bounds on both corners anyway
    const float fc = fminf(fmaxf(f, -2.0f), 1.0e9f);
    i0 = (int)fc;
    w1 = __fsub_rn(u, f);                    // ix - ix_tnw
    w0 = __fsub_rn(__fadd_rn(f, 1.0f), u);   // ix_bse - ix
}

// Full sample for voxel (x,y,z) under beta `b`.  Z == 1: z pinned to 0 (weight 1 on slice 0).
// HASZ = false is the Z == 1 specialisation (z == 0 folds the z terms of the polynomial away).
template <bool HASZ, int FAST = -1>
__device__ __forceinline__ Sample make_sample_t(const float *b, const Volume &vol, int xi, int yi, int zi) {
    const float x = (float)xi, y = (float)yi, z = HASZ ? (float)zi : 0.0f;
    Sample s;
    s.ux = unnormalise(grid_n<HASZ, FAST>(b, vol, 0, x, y, z), vol.hx1);
    s.uy = unnormalise(grid_n<HASZ, FAST>(b, vol, 1, x, y, z), vol.hy1);
    axis_weights(s.ux, s.x0, s.wx0, s.wx1);
    axis_weights(s.uy, s.y0, s.wy0, s.wy1);
    if (HASZ) {
        s.uz = unnormalise(grid_n<HASZ, FAST>(b, vol, 2, x, y, z), vol.hz1);
        axis_weights(s.uz, s.z0, s.wz0, s.wz1);
    } else {
        s.uz = 0.0f, s.z0 = 0, s.wz0 = 1.0f, s.wz1 = 0.0f;
    }
    return s;
}

__device__ __forceinline__ Sample make_sample(const float *b, const Volume &vol, int xi, int yi, int zi) {
    return vol.Z > 1 ? make_sample_t<true>(b, vol, xi, yi, zi) : make_sample_t<false>(b, vol, xi, yi, zi);
}

// ---- halo layout of the gathered single-channel images ------------------------------------------------------
// The reconstruction images S (K2's source) and the neuron-major footprints At (K3n's source) carry a zero halo of
// DNMF_HALO voxels around x and y: voxel (x,y,z) lives at (x + HALO) row + (y + HALO) Z + z, where a row holds
// (Y + 2 HALO) Z floats rounded up to a multiple of 32 -- rows start on 128-byte lines, so the 256-byte runs a wave
// stores cover whole lines (with rows of 516 floats every run straddled three lines and the stores ran at a third
// of the rate) -- and there are Xp = X + 2 HALO rows.  A tap cell whose base corner is clamped into [-HALO, S] then never leaves the buffer, and a
// corner outside the volume reads a zero instead of being masked: `w * 0` is the exact zero the per-corner bounds
// test of grid_sample yields (its backward skips such corners, i.e. adds zeros too), so neither weight masks nor
// index clamps nor validity flags are needed along x and y.  z keeps masks and clamps (a halo would triple a
// two-slice volume).
constexpr int HALO = DNMF_HALO;

struct HaloLayout {
    int Xp, Yp;        // rows; voxels of a row along y including the border
    int rowf;          // floats of a row: Yp * Z rounded up to a multiple of 32 (the excess is zero like the border)
    long Pp;           // Xp * rowf
    int row4, col4;    // byte strides of one step in x and in y: rowf*4, Z*4
    unsigned origin4;  // byte offset of voxel (0,0,0)
    float xhi, yhi;    // upper clamp of a source coordinate: S + 0.5 (base corner <= S)
    float row4f, col4f, origin4f;  // the strides and the origin as floats
    int f32off;        // an image has fewer than 2^24 bytes: tap offsets are exact in fp32 arithmetic
};

inline HaloLayout make_halo_layout(int X, int Y, int Z) {
    HaloLayout h;
    h.Xp = X + 2 * HALO, h.Yp = Y + 2 * HALO;
    h.rowf = (h.Yp * Z + 31) / 32 * 32;
    h.Pp = (long)h.Xp * h.rowf;
    h.row4 = h.rowf * 4, h.col4 = Z * 4;
    h.origin4 = (unsigned)(((long)HALO * h.rowf + (long)HALO * Z) * 4);
    h.xhi = (float)X + 0.5f, h.yhi = (float)Y + 0.5f;
    h.row4f = (float)h.row4, h.col4f = (float)h.col4, h.origin4f = (float)h.origin4;
    h.f32off = h.Pp * 4 < (1L << 24);
    return h;
}

// One axis of a gather from a halo layout: source coordinate u -> base corner f (an integer in [-HALO, S], kept as a
// float) and the weights of corner f / f + 1 exactly as axis_weights forms them; a coordinate beyond the halo (or
// NaN: v_med3_f32 then returns the smallest operand) is pulled onto it, where both corners read zeros, so its
// weights do not matter as long as they are finite.
__device__ __forceinline__ void axis_taps_halo(float u, float hi, float &f, float &w0, float &w1) {
    const float uc = __builtin_amdgcn_fmed3f(u, -(float)HALO, hi);
    f = floorf(uc);
    w1 = __fsub_rn(uc, f);
    w0 = __fsub_rn(__fadd_rn(f, 1.0f), uc);
}

// Byte offset of base corner (fx, fy) from `origin` (the offset of voxel (0,0,z)).  F32OFF: two FMAs and one
// conversion, exact because every intermediate is an integer below 2^24 (HaloLayout::f32off); else integer
// conversions and 24-bit multiplies.  (On gfx950 conversions, 24-bit multiplies, v_med3 and DPP operations issue at
// half the rate of fp32 adds / multiplies / FMAs -- tools/valu_probe.hip.)
template <bool F32OFF>
__device__ __forceinline__ unsigned halo_offset(float fx, float fy, const HaloLayout &h, unsigned origin, float originf) {
    if (F32OFF) return (unsigned)fmaf(fx, h.row4f, fmaf(fy, h.col4f, originf));
    return (unsigned)(__mul24((int)fx, h.row4) + __mul24((int)fy, h.col4)) + origin;
}

// Z == 2: the two z-taps of a corner are the adjacent floats (slice 0, slice 1) of a halo row and are fetched as that
// pair.  With f = floor(u) the reference's weights are u - f for tap f + 1 and (f + 1) - u for tap f (axis_weights), and a
// tap outside [0, Z) is dropped (grid_sample's per-corner bounds test).  Slice 0 is tap f + 1 for u in [-1, 0) (weight
// u + 1) and tap f for u in [0, 1) (weight 1 - u); slice 1 is tap f + 1 for u in [0, 1) (weight u) and tap f for u in
// [1, 2) (weight 2 - u): the same fp32 operations on the same operands, picked by one v_med3 each (the smaller of the two
// expressions while both are positive, 0 once one is not).  NaN and far-away coordinates are pulled to -2 / 3 first,
// where both weights are 0.  Returns the clamped coordinate.
__device__ __forceinline__ float z_pair_weights(float uz, float &w0, float &w1) {
    const float uc = __builtin_amdgcn_fmed3f(uz, -2.0f, 3.0f);
    w0 = __builtin_amdgcn_fmed3f(0.0f, __fadd_rn(uc, 1.0f), __fsub_rn(1.0f, uc));
    w1 = __builtin_amdgcn_fmed3f(0.0f, uc, __fsub_rn(2.0f, uc));
    return uc;
}

__device__ __forceinline__ bool in_range(int i, int n) { return (unsigned)i < (unsigned)n; }

// The NTAP = 4 (Z == 1) or 8 gather taps of a sample, corner order of ATen's grid_sampler (dx fastest, then dy,
// then dz): weight w[c] (0 for a corner outside the volume: per axis the weight of an outside corner is zeroed,
// and a product with a zero factor is the zero the per-corner bounds test yields) and voxel index vox[c] (corner
// clamped into the volume, so it can always be dereferenced).
template <int NTAP>
__device__ __forceinline__ void make_taps(const Sample &sm, const Volume &vol, float *w, unsigned *vox) {
    const float wxm[2] = {in_range(sm.x0, vol.X) ? sm.wx0 : 0.0f, in_range(sm.x0 + 1, vol.X) ? sm.wx1 : 0.0f};
    const float wym[2] = {in_range(sm.y0, vol.Y) ? sm.wy0 : 0.0f, in_range(sm.y0 + 1, vol.Y) ? sm.wy1 : 0.0f};
    const int xc[2] = {clamp_index(sm.x0, vol.X), clamp_index(sm.x0 + 1, vol.X)};
    const int yc[2] = {clamp_index(sm.y0, vol.Y), clamp_index(sm.y0 + 1, vol.Y)};
    float wzm[2] = {1.0f, 0.0f};
    int zc[2] = {0, 0};
    if (NTAP == 8) {
        wzm[0] = in_range(sm.z0, vol.Z) ? sm.wz0 : 0.0f, wzm[1] = in_range(sm.z0 + 1, vol.Z) ? sm.wz1 : 0.0f;
        zc[0] = clamp_index(sm.z0, vol.Z), zc[1] = clamp_index(sm.z0 + 1, vol.Z);
    }
#pragma unroll
    for (int c = 0; c < NTAP; ++c) {
        const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
        float wc = __fmul_rn(wxm[dx], wym[dy]);
        unsigned v = (unsigned)(xc[dx] * vol.Y + yc[dy]);
        if (NTAP == 8) {
            wc = __fmul_rn(wc, wzm[dz]);
            v = v * (unsigned)vol.Z + (unsigned)zc[dz];
        }
        w[c] = wc, vox[c] = v;
    }
}

// ---- where the taps of a box of voxels can fall ------------------------------------------------------------
// Conservative range of a_d = 2 q_d over a box of voxel coordinates lo <= (x,y,z) <= hi, all >= 0: every monomial is
// monotone there, so a term's range follows from the sign of its coefficient (interval arithmetic on the ten terms).
// `mag` receives the sum of the magnitudes of the ten terms at the far corner of the box: the partial sums of either
// evaluation (the interval arithmetic here, the FMA chain of poly_a) stay below it, so each of their ~10 roundings is
// at most mag 2^-24.
__device__ __forceinline__ void poly_range(const float *b, int d, const float (&lo)[3], const float (&hi)[3], bool hasz,
                                           float &amin, float &amax, float &mag) {
    const float mlo[9] = {lo[0], lo[1], lo[2], lo[0] * lo[0], lo[1] * lo[1], lo[2] * lo[2], lo[0] * lo[1], lo[0] * lo[2], lo[1] * lo[2]};
    const float mhi[9] = {hi[0], hi[1], hi[2], hi[0] * hi[0], hi[1] * hi[1], hi[2] * hi[2], hi[0] * hi[1], hi[0] * hi[2], hi[1] * hi[2]};
    amin = amax = b[d];
    mag = fabsf(b[d]);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const bool zterm = i == 2 || i == 5 || i == 7 || i == 8;
        if (zterm && !hasz) continue;
        const float c = b[3 * (i + 1) + d];
        const float p = c * mlo[i], q = c * mhi[i];
        amin += fminf(p, q), amax += fmaxf(p, q);
        mag += fabsf(q);
    }
    amin *= 2.0f, amax *= 2.0f, mag *= 2.0f;
}

// Integer range [a, c] that contains the base corner AND the second corner (base + 1) along axis d of every voxel of the
// box, clamped to [-4, S + 5]: the source coordinate is a non-decreasing function of a_d, so its range follows from
// poly_range; the margin covers the fp32 rounding of either evaluation: a fixed part for the normalise / un-normalise
// round trip (a few roundings at magnitudes up to ~2S: errors of order S * 1e-6 voxels) plus 2^-20 of the summed term
// magnitudes -- with large quadratic coefficients whose terms cancel (|c| S^2 >> S while the voxel stays in the volume)
// the partial sums, and with them the rounding of the ten-term chain, are far larger than the coordinate itself; a fixed
// margin would then drop a neuron from a tile's list without a trace (tests/test_gpu_parity.py:
// test_neuron_list_gram_with_cancelling_quadratic_terms).  False when the coordinates are NaN (such voxels gather zeros
// from the halo).
__device__ __forceinline__ bool tap_range(const float *b, const Volume &vol, int d, const float (&lo)[3], const float (&hi)[3],
                                          bool hasz, int &a, int &c) {
    const int S = d == 0 ? vol.X : (d == 1 ? vol.Y : vol.Z);
    const float h = d == 0 ? vol.hx1 : (d == 1 ? vol.hy1 : vol.hz1);
    float amin, amax, mag;
    poly_range(b, d, lo, hi, hasz, amin, amax, mag);
    const float margin = 0.0625f + 4e-6f * (float)S + 9.54e-7f * mag;
    const float ulo = unnormalise(normalise_axis<-1>(amin, vol, d), h) - margin;
    const float uhi = unnormalise(normalise_axis<-1>(amax, vol, d), h) + margin;
    if (!(ulo <= uhi)) return false;
    a = (int)floorf(fminf(fmaxf(ulo, -4.0f), (float)S + 4.0f));
    c = (int)floorf(fminf(fmaxf(uhi, -4.0f), (float)S + 4.0f)) + 1;
    return true;
}

// sum over the 64 lanes, valid in lane 63 (fixed tree: row prefix sums, then the row totals)
__device__ __forceinline__ float wave_sum_last(float v) {
#define DNMF_STEP(ctrl, rmask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, true));
    DNMF_STEP(0x111, 0xf)  // row_shr:1
    DNMF_STEP(0x112, 0xf)  // row_shr:2
    DNMF_STEP(0x114, 0xf)  // row_shr:4
    DNMF_STEP(0x118, 0xf)  // row_shr:8 -> lane 15 of a row holds the row sum
    DNMF_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    DNMF_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3
#undef DNMF_STEP
    return v;
}

// Load the 30 coefficients of frame t from beta (10,3,T) into b[a*3+d].
__device__ __forceinline__ void load_beta(const float *__restrict__ beta, int T, int t, float *b) {
#pragma unroll
    for (int i = 0; i < 30; ++i) b[i] = beta[(long)i * T + t];
}

// n / d for 0 <= n < 2^24, d > 0, with rcp = 1/d rounded: the float estimate is off by at most one
__device__ __forceinline__ int div_small(int n, int d, float rcp) {
    int q = (int)((float)n * rcp);
    const int r = n - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// voxel index -> (x,y,z), p = (x*Y + y)*Z + z
__device__ __forceinline__ void voxel_xyz(long p, const Volume &vol, int &x, int &y, int &z) {
    const int yz = vol.Y * vol.Z;
    if (vol.P <= (1L << 24)) {  // uniform branch: every index is exact in fp32
        x = div_small((int)p, yz, vol.rcp_yz);
        const int rem = (int)p - x * yz;
        if (vol.Z == 1) {
            y = rem, z = 0;
        } else {
            y = div_small(rem, vol.Z, vol.rcp_z);
            z = rem - y * vol.Z;
        }
    } else {
        x = (int)(p / yz);
        const int rem = (int)(p - (long)x * yz);
        y = rem / vol.Z;
        z = rem - y * vol.Z;
    }
}

}  // namespace dnmf
