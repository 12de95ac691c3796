// K2 -- one mini-batch of the motion step, fused: warp, reconstruction, squared error and the
// analytic gradient with respect to beta.
//
// Reference: DeformableNMF.update_motion, Demix/dNMF.py:186-190 = ExponentialFP.forward
// (dNMF.py:54-58) + F.mse_loss (dNMF.py:188) + autograd backward through grid_sample (w.r.t. the
// grid), the normalisation and the einsum with the quadratic basis.  With s = A.C_t (dnmf_recon_image)
//   A_tC(v)      = sum_corners w_c(v) s(corner_c(v))
//   d A_tC / d q = sum_corners (d w_c / d q) s(corner_c(v))          (in-bounds corners only)
//   d L / d beta[a,d] = 2/(B P) sum_v basis_a(v) (A_tC(v) - y(v)) dA_tC/dq_d(v)
// The factor (S-1)/2 of grid_sample's backward and the 2/(S-1) of the normalisation cancel.
#include <type_traits>

#include "common.hpp"

namespace dnmf {

typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));   // four floats at an 8-byte aligned address
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

constexpr int K2_ROWS = 32;           // voxels per lane: consecutive x (the block reduction is paid once per K2_ROWS voxels)
constexpr int K2_COLS = 256;          // positions of the (y,z) plane per block: 64 lanes x 4 waves
constexpr int K2_NACC = 32;           // 30 gradient sums + squared error + pad
#ifndef DNMF_K2_UNROLL
#define DNMF_K2_UNROLL 2
#endif
#ifndef DNMF_K2_UNROLL_Z
#define DNMF_K2_UNROLL_Z 1
#endif
#ifndef DNMF_K2_REUSE
#define DNMF_K2_REUSE 1   // the lower tap row from the last voxel's upper one when the whole wave steps by one row
#endif
constexpr bool K2_REUSE = DNMF_K2_REUSE != 0;
#ifndef DNMF_K2_WAVES_Z
#define DNMF_K2_WAVES_Z 1
#endif
constexpr int K2_UNROLL = DNMF_K2_UNROLL;      // rows requested together (Z == 1)
constexpr int K2_UNROLL_Z = DNMF_K2_UNROLL_Z;  // the same for Z >= 2
static_assert(K2_ROWS % K2_UNROLL == 0 && K2_ROWS % K2_UNROLL_Z == 0, "the row loop takes K2_UNROLL rows at a time");

// (sum_loss_kernel only; the main kernel reduces with DPP adds, common.hpp: wave_sum_last)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Work layout: a block owns K2_ROWS x-rows by 256 consecutive positions of the (y,z) plane (lane = position, so the frame
// and the reconstruction image are read in 256-byte runs and the taps of neighbouring lanes share cache lines); a
// thread walks down its rows with (y,z) fixed: the monomials without x are per-thread constants, and x, x^2 are the
// same for the whole block: they come from a table (scalar loads).  The polynomial itself is the reference's chain of
// ten (six for Z = 1) fused multiply-adds, in its order (common.hpp: poly_a).
//
// S carries the zero halo of common.hpp around x and y: a tap outside the volume reads a zero instead of being
// masked, for the value and for the gradient alike (grid_sample's backward skips out-of-bounds corners: it adds the
// same zeros).  Per voxel that leaves: two FMAs per coordinate, the fp32 normalise / un-normalise round trip of the
// reference, one clamp + floor + two weights per axis, one base offset, the taps, the blends.
//
// ZM = 1 is the Z == 1 specialisation: two coordinates, four taps, and only the six basis terms without z
// (the other 18 gradient sums are identically zero and are written as such).  PLAIN = the fit step's call (frames
// given, no upstream gradient, A_tC not wanted): the row loop then has no branches.
//
// Z >= 2 (ZM = 2: Z == 2, ZM = 3: Z > 2).  The two z-taps of a corner are adjacent floats of a halo row
// ((y + HALO) Z + z): they are fetched as ONE pair (izc, izc + 1) with izc = the base slice clamped into [0, Z - 2], i.e.
// the pair inside the volume that holds every in-range z-tap of the sample; a tap outside [0, Z) has no partner in the
// pair and its weight is dropped -- what grid_sample's per-corner bounds test does -- by giving each member of the pair
// the weight of the tap it stands for (or 0).  For Z == 2 the pair is always (0, 1) and the pairs of the corners y and
// y + 1 are 16 contiguous bytes: a voxel is TWO sixteen-byte gathers (one per x-corner; round 2 issued eight four-byte
// ones, 8.1 ms per 4000 frames at 512x512x2); for Z > 2 four eight-byte ones.
template <int ZM, int FAST, bool F32OFF, bool PLAIN>
__global__ __launch_bounds__(256, (ZM > 1 ? DNMF_K2_WAVES_Z : 1)) void warp_recon_grad_kernel(const float *__restrict__ S, long lds,
                                                              const int *__restrict__ s_ids,
                                                              const float *__restrict__ frames, long ldf,
                                                              const int *__restrict__ frame_ids,
                                                              const float *__restrict__ gout, Volume vol, HaloLayout hl,
                                                              const float *__restrict__ beta, int T,
                                                              const int *__restrict__ times,
                                                              float *__restrict__ recon, float *__restrict__ partial,
                                                              const float2 *__restrict__ xtab, int nub) {
    constexpr bool HASZ = ZM > 1;
    constexpr bool ZPAIR = ZM == 2;                  // a thread owns BOTH slices of its (x, y) columns
    constexpr int NV = ZPAIR ? 2 : 1;                // voxels per thread and row
    constexpr int ND = HASZ ? 3 : 2;                 // warped coordinates that exist
    constexpr int NA = HASZ ? 10 : 6;                // basis terms that are not identically zero
    constexpr int BASIS_ID[10] = {0, 1, 2, 4, 5, 7, 3, 6, 8, 9};  // z-free terms first
    const int b = blockIdx.y;
    const char *__restrict__ s = reinterpret_cast<const char *>(S + (long)(s_ids ? s_ids[b] : b) * lds);
    const float *__restrict__ y = frames ? frames + (long)(frame_ids ? frame_ids[b] : b) * ldf : nullptr;
    const float *__restrict__ go = !PLAIN && gout ? gout + (long)b * vol.P : nullptr;
    float *__restrict__ rc = !PLAIN && recon ? recon + (long)b * vol.P : nullptr;
    float bt[30];
    load_beta(beta, T, times[b], bt);

    // y and z are fixed along a thread's rows, so inside the loop only the moments of the per-voxel term over x are
    // accumulated: mom[.][m][d] = sum_i (resid g_d)_i x_i^m, m = 0, 1, 2; the ten basis sums follow from them afterwards
    float mom[NV][3][ND];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int d = 0; d < ND; ++d) mom[v][m][d] = 0.0f;
    float sq = 0.0f;
    const int YZ = vol.Y * vol.Z;
    const int plane = ZPAIR ? vol.Y : YZ;              // positions the lanes of a frame's blocks run over
    const int bu = blockIdx.x % nub, bx = blockIdx.x / nub;
    const int u = bu * K2_COLS + threadIdx.x;          // position in the (y,z) plane; Z == 2: y
    const int yy = (HASZ && !ZPAIR) ? div_small(u, vol.Z, vol.rcp_z) : u;
    const int z = (HASZ && !ZPAIR) ? u - yy * vol.Z : 0;
    const float yf = (float)yy, zf = (float)z;

    if (u < plane) {
        float b2[30];
        double_beta(bt, b2);
        const Monomials<true> mono = monomials<true>(0.0f, yf, zf);   // y, z, y^2, z^2, yz: fixed along the rows
        const int x_first = bx * K2_ROWS;
        const int nrow = min(K2_ROWS, vol.X - x_first);

        // A row is handled in two steps so that the loop below can keep the taps of the next rows in flight while
        // it blends the current ones: request() = coordinates, weights, tap offsets, loads; consume() = the blends,
        // the residual and the moment sums.
        struct Vox {
            float sv[HASZ ? 2 : 1][2][2];  // [dz][dy][dx]
            float wx1, wy1, wzm[2], vz[2];
            float other;                  // frame value, or the caller's upstream gradient
        };
        struct Req {
            float2 xv;                    // (x, x*x): block-uniform
            long prow;
            Vox v[NV];
        };
        unsigned u4 = (unsigned)u * (4u * NV);
        // ZC: the voxel's slice when it is known at compile time (Z == 2: 0 or 1), else -1.  z == 0: the four terms with z
        // add an exact zero each (the Z == 1 chain); z == 1: their monomials are 1, 1, x, y.
        unsigned prev_o[NV];
        float prev_v[NV][HASZ ? 4 : 2];
#pragma unroll
        for (int v = 0; v < NV; ++v) prev_o[v] = 0xffffffffu;
        auto taps = [&](const float2 xv, float xy, auto zc, int slot, Vox &q) {
            constexpr int ZC = decltype(zc)::value;
            float a[3] = {0.0f, 0.0f, 0.0f};
            if constexpr (ZC == 0) {
                Monomials<false> m;
                m.x = xv.x, m.y = yf, m.z = 0.0f, m.xx = xv.y, m.yy = mono.yy, m.zz = 0.0f, m.xy = xy, m.xz = 0.0f, m.yz = 0.0f;
#pragma unroll
                for (int d = 0; d < 3; ++d) a[d] = poly_a<false>(b2, d, m);
            } else {
                Monomials<true> m = mono;
                m.x = xv.x, m.xx = xv.y, m.xy = xy;
                if constexpr (ZC == 1) {
                    m.z = 1.0f, m.zz = 1.0f, m.xz = xv.x, m.yz = yf;
                } else if (HASZ) {
                    m.xz = __fmul_rn(xv.x, zf);
                }
#pragma unroll
                for (int d = 0; d < ND; ++d) a[d] = HASZ ? poly_a<true>(b2, d, m) : poly_a<false>(b2, d, Monomials<false>{m.x, m.y, 0.0f, m.xx, m.yy, 0.0f, m.xy, 0.0f, 0.0f});
            }
            float fx, fy, w0;
            axis_taps_halo(unnormalise(normalise_axis<FAST>(a[0], vol, 0), vol.hx1), hl.xhi, fx, w0, q.wx1);
            axis_taps_halo(unnormalise(normalise_axis<FAST>(a[1], vol, 1), vol.hy1), hl.yhi, fy, w0, q.wy1);
            const unsigned o0 = halo_offset<F32OFF>(fx, fy, hl, hl.origin4, hl.origin4f);   // base corner, slice 0
            unsigned zo = 0u;
            if (HASZ) {
                const float uz = unnormalise(normalise_axis<FAST>(a[2], vol, 2), vol.hz1);
                if (ZM == 2) {
                    // the pair is (0, 1): weights of its members by common.hpp: z_pair_weights.  d weight / d u of
                    // member 0 is +1 for f = floor(u) = -1 and -1 for f = 0, of member 1 +1 for f = 0 and -1 for f = 1,
                    // else 0: with c = 2 f + 1 (c - 2) that is -c where |c| = 1.
                    const float uc = z_pair_weights(uz, q.wzm[0], q.wzm[1]);
                    const float c0 = fmaf(2.0f, floorf(uc), 1.0f), c1 = c0 - 2.0f;
                    q.vz[0] = fabsf(c0) == 1.0f ? -c0 : 0.0f;
                    q.vz[1] = fabsf(c1) == 1.0f ? -c1 : 0.0f;
                } else {
                    int iz;
                    float wz[2];
                    axis_weights(uz, iz, wz[0], wz[1]);
                    // member 0 / 1 of the pair (izc, izc + 1) stands for tap iz / iz + 1 when iz == izc, member 0 for tap
                    // iz + 1 when iz == izc - 1 (tap iz = -1 is outside), member 1 for tap iz when iz == izc + 1 (tap
                    // iz + 1 = Z is outside); vz = d weight / d u_z of the tap a member stands for
                    const int izc = clamp_index(iz, vol.Z - 1);
                    const bool same = iz == izc, below = iz + 1 == izc, above = iz == izc + 1;
                    q.wzm[0] = same ? wz[0] : (below ? wz[1] : 0.0f);
                    q.wzm[1] = same ? wz[1] : (above ? wz[0] : 0.0f);
                    q.vz[0] = same ? -1.0f : (below ? 1.0f : 0.0f);
                    q.vz[1] = same ? 1.0f : (above ? -1.0f : 0.0f);
                    zo = (unsigned)izc * 4u;
                }
            }
            // The tap row dx = 0 of this voxel is the tap row dx = 1 of the thread's last voxel whenever the warp moves the
            // base corner by exactly one row between consecutive x (every near-identity warp): when that holds for all lanes of
            // the wave the values are still in registers and only the upper row is gathered -- the kernel is bound by the
            // gather path (TA 97 % busy at Z = 1, 79 % at Z = 2: misaligned 8 / 16-byte gathers, ~16 cycles per
            // wave-instruction), this halves its load for such warps: Z = 1 1.95 -> 1.72 ms per 4000 frames at 512x512 in the
            // bench, 4.6 TB/s -> 5.05 TB/s = what a device copy reaches.  Z >= 2 is bound by instruction issue instead: the
            // test and the moves cost it 8 % (4.66 -> 5.05 ms), so it is Z == 1 only.  Same bytes either way.
            constexpr int NT = HASZ ? 4 : 2;
            const unsigned olo = o0 + (ZM == 3 ? zo : 0u), ohi = olo + (unsigned)hl.row4;
            auto gather = [&](unsigned o, float (&v)[NT]) {
                asm("" : "+v"(o));
                const char *t = s + o;   // (scalar base + 32-bit offset) loads
                if constexpr (ZM == 2) {          // (y, z0), (y, z1), (y + 1, z0), (y + 1, z1): one load, 8-byte aligned
                    const f32x4_a8 r = *reinterpret_cast<const f32x4_a8 *>(t);
                    v[0] = r.x, v[1] = r.y, v[2] = r.z, v[3] = r.w;
                } else if constexpr (ZM == 3) {   // the z-pair of corner y, then of corner y + 1
                    const f32x2_a4 r0 = *reinterpret_cast<const f32x2_a4 *>(t);
                    const f32x2_a4 r1 = *reinterpret_cast<const f32x2_a4 *>(t + hl.col4);
                    v[0] = r0.x, v[1] = r0.y, v[2] = r1.x, v[3] = r1.y;
                } else {
                    v[0] = *reinterpret_cast<const float *>(t);
                    v[1] = *reinterpret_cast<const float *>(t + 4);
                }
            };
            float lo[NT], hi[NT];
            if (K2_REUSE && (ZM == 1 || DNMF_K2_REUSE > 1) && __ballot(olo != prev_o[slot]) == 0) {   // wave-uniform
#pragma unroll
                for (int i = 0; i < NT; ++i) lo[i] = prev_v[slot][i];
            } else {
                gather(olo, lo);
            }
            gather(ohi, hi);
            prev_o[slot] = ohi;
#pragma unroll
            for (int i = 0; i < NT; ++i) prev_v[slot][i] = hi[i];
            if constexpr (HASZ) {
                q.sv[0][0][0] = lo[0], q.sv[1][0][0] = lo[1], q.sv[0][1][0] = lo[2], q.sv[1][1][0] = lo[3];
                q.sv[0][0][1] = hi[0], q.sv[1][0][1] = hi[1], q.sv[0][1][1] = hi[2], q.sv[1][1][1] = hi[3];
            } else {
                q.sv[0][0][0] = lo[0], q.sv[0][1][0] = lo[1], q.sv[0][0][1] = hi[0], q.sv[0][1][1] = hi[1];
            }
        };
        auto request = [&](int x, Req &q) {
            q.xv = xtab[x];                   // a scalar load
            q.prow = (long)x * YZ;            // block-uniform: (scalar base + 32-bit lane offset) accesses
            const float xy = __fmul_rn(q.xv.x, yf);
            asm("" : "+v"(u4));
            const char *op = reinterpret_cast<const char *>((!PLAIN && go ? go : y) + q.prow) + u4;
            if constexpr (ZPAIR) {
                taps(q.xv, xy, std::integral_constant<int, 0>{}, 0, q.v[0]);
                taps(q.xv, xy, std::integral_constant<int, 1>{}, NV - 1, q.v[1]);
                const float2 o2 = *reinterpret_cast<const float2 *>(op);   // (x, y, 0), (x, y, 1): eight aligned bytes
                q.v[0].other = o2.x, q.v[NV - 1].other = o2.y;
            } else {
                taps(q.xv, xy, std::integral_constant<int, -1>{}, 0, q.v[0]);
                q.v[0].other = *reinterpret_cast<const float *>(op);
            }
        };
        auto consume = [&](const Req &rq) {
            float recs[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const Vox &q = rq.v[v];
                float rec = 0.0f, g[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int dz = 0; dz < (HASZ ? 2 : 1); ++dz) {
                    // x- and y-blends as s0 + w1 (s1 - s0): the weights of an axis add up to exactly 1 (u - f and
                    // (f + 1) - u are exact), the differences are the gradient's anyway -- one operation less per blend
                    // than w0 s0 + w1 s1
                    const float d0 = q.sv[dz][0][1] - q.sv[dz][0][0];                          // x-differences of the rows
                    const float d1 = q.sv[dz][1][1] - q.sv[dz][1][0];
                    const float a0 = fmaf(q.wx1, d0, q.sv[dz][0][0]);                          // x-interpolated rows y0, y1
                    const float a1 = fmaf(q.wx1, d1, q.sv[dz][1][0]);
                    const float gy2 = a1 - a0;
                    const float r2 = fmaf(q.wy1, gy2, a0);                                     // value of this z-slice
                    const float gx2 = fmaf(q.wy1, d1 - d0, d0);
                    if (HASZ) {
                        rec = fmaf(q.wzm[dz], r2, rec);
                        g[0] = fmaf(q.wzm[dz], gx2, g[0]);
                        g[1] = fmaf(q.wzm[dz], gy2, g[1]);
                        g[2] = fmaf(q.vz[dz], r2, g[2]);
                    } else {
                        rec = r2, g[0] = gx2, g[1] = gy2;
                    }
                }
                recs[v] = rec;
                // upstream gradient: the mse residual (scaled by 2/(B P) in the finish kernel) or the caller's
                const float resid = !PLAIN && go ? q.other : rec - q.other;
                sq = fmaf(resid, resid, sq);
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const float gd = resid * g[d];
                    mom[v][0][d] += gd;
                    mom[v][1][d] = fmaf(rq.xv.x, gd, mom[v][1][d]);
                    mom[v][2][d] = fmaf(rq.xv.y, gd, mom[v][2][d]);
                }
            }
            if (!PLAIN && rc) {
                char *dst = reinterpret_cast<char *>(rc + rq.prow) + u4;
                if constexpr (ZPAIR)
                    *reinterpret_cast<float2 *>(dst) = make_float2(recs[0], recs[NV - 1]);
                else
                    *reinterpret_cast<float *>(dst) = recs[0];
            }
        };
        if (nrow == K2_ROWS) {
            // K2_UNROLL rows at a time: their taps are requested together, then blended (keeping the next rows' taps in
            // flight behind the blends of the current ones did not pay: 2.07 against 1.99 ms, tools/time_k2.py)
            constexpr int UR = HASZ ? K2_UNROLL_Z : K2_UNROLL;
#pragma unroll 1
            for (int i = 0; i < K2_ROWS; i += UR) {
                Req q[UR];
#pragma unroll
                for (int j = 0; j < UR; ++j) request(x_first + i + j, q[j]);
#pragma unroll
                for (int j = 0; j < UR; ++j) consume(q[j]);
            }
        } else {
            for (int i = 0; i < nrow; ++i) {
                Req q;
                request(x_first + i, q);
                consume(q);
            }
        }
    }
    // basis order [1, x, y, x^2, y^2, xy, z, z^2, xz, yz] (BASIS_ID maps it to the reference's)
    float acc[NA][ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        // Z == 2: the z-free terms sum both slices, the terms with z are slice 1's (z = 1; slice 0 contributes zeros)
        const float m0 = ZPAIR ? mom[0][0][d] + mom[NV - 1][0][d] : mom[0][0][d];
        const float m1 = ZPAIR ? mom[0][1][d] + mom[NV - 1][1][d] : mom[0][1][d];
        const float m2 = ZPAIR ? mom[0][2][d] + mom[NV - 1][2][d] : mom[0][2][d];
        acc[0][d] = m0, acc[1][d] = m1, acc[2][d] = yf * m0, acc[3][d] = m2;
        acc[4][d] = (yf * yf) * m0, acc[5][d] = yf * m1;
        if (HASZ) {
            const float z0 = mom[NV - 1][0][d], z1 = mom[NV - 1][1][d];
            const float zz = ZPAIR ? 1.0f : zf;
            acc[6][d] = zz * z0, acc[7][d] = (zz * zz) * z0, acc[8][d] = zz * z1;
            acc[9][d] = (yf * zz) * z0;
        }
    }

    // block reduction: a DPP tree inside each wave (six vector adds per sum, total in lane 63; the butterflies of
    // __shfl_xor it replaces are LDS permutes with an address computation each -- 31 sums x 6 steps per 16 rows were
    // a fifth of the kernel's instructions), then the four waves' totals through LDS
    __shared__ float red[4][K2_NACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < K2_NACC) red[wave][lane] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const float v = wave_sum_last(acc[a][d]);
            if (lane == 63) red[wave][BASIS_ID[a] * 3 + d] = v;
        }
    {
        const float v = wave_sum_last(sq);
        if (lane == 63) red[wave][30] = v;
    }
    __syncthreads();
    if (threadIdx.x < K2_NACC) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        partial[((long)b * gridDim.x + blockIdx.x) * K2_NACC + threadIdx.x] = v;
    }
}

// (x, x*x) as floats for every x of the volume: the table the main kernel reads with scalar loads
__global__ void k2_xtab_kernel(float2 *__restrict__ xtab, int X) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < X) {
        const float xf = (float)x;
        xtab[x] = make_float2(xf, __fmul_rn(xf, xf));
    }
}

// log|det J| at a point, literally Demix/dNMF.py:107-122 (rows 8/9 used as yz/xz there).
__device__ float log_det_jac_dev(const float *b, float x, float y, float z) {
    float J[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        J[0][c] = b[3 + c] + 2.0f * b[12 + c] * x + b[21 + c] * y + b[27 + c] * z;
        J[1][c] = b[6 + c] + 2.0f * b[15 + c] * y + b[21 + c] * x + b[24 + c] * z;
        J[2][c] = b[9 + c] + 2.0f * b[18 + c] * z + b[24 + c] * y + b[27 + c] * x;
    }
    const float a = J[0][0], bb = J[1][0], c = J[2][0], d = J[0][1], e = J[1][1], f = J[2][1], g = J[0][2],
                h = J[1][2], i = J[2][2];
    const float det = a * (e * i - f * h) - bb * (d * i - f * g) + c * (d * h - e * g);
    return logf(fabsf(det));
}

// One block per frame: ordered sum of the per-block partials, scale, add into grad, per-frame loss, reg.
__global__ __launch_bounds__(64) void warp_recon_grad_finish_kernel(const float *__restrict__ partial, int nblk,
                                                                    Volume vol, const float *__restrict__ beta, int T,
                                                                    const int *__restrict__ times, int norm_frames,
                                                                    float grad_scale, float *__restrict__ grad,
                                                                    float *__restrict__ frame_loss,
                                                                    float *__restrict__ reg) {
    const int b = blockIdx.x;
    const int t = times[b];
    const int j = threadIdx.x;
    const float inv_n = 1.0f / ((float)norm_frames * (float)vol.P);
    if (j < 31) {
        const float *src = partial + (long)b * nblk * K2_NACC + j;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;  // four interleaved chains, fixed order
        int k = 0;
        for (; k + 3 < nblk; k += 4) {
            s0 += src[(long)(k + 0) * K2_NACC];
            s1 += src[(long)(k + 1) * K2_NACC];
            s2 += src[(long)(k + 2) * K2_NACC];
            s3 += src[(long)(k + 3) * K2_NACC];
        }
        for (; k < nblk; ++k) s0 += src[(long)k * K2_NACC];
        const float tot = (s0 + s1) + (s2 + s3);
        if (j < 30) {
            if (grad) grad[(long)j * T + t] += grad_scale * tot;
        } else if (frame_loss) {
            frame_loss[b] = tot * inv_n;
        }
    }
    if (j == 32 && reg) {
        float bt[30];
        load_beta(beta, T, t, bt);
        const float l1 = log_det_jac_dev(bt, vol.sx1, vol.sy1, vol.sz1);
        const float l0 = log_det_jac_dev(bt, 0.0f, 0.0f, 0.0f);
        reg[b] = l1 * l1 + l0 * l0;
    }
}

__global__ __launch_bounds__(64) void sum_loss_kernel(const float *__restrict__ frame_loss, int B,
                                                      float *__restrict__ loss) {
    float v = 0.0f;
    for (int i = threadIdx.x; i < B; i += 64) v += frame_loss[i];
    v = wave_sum(v);
    if (threadIdx.x == 0) loss[0] = v;
}

}  // namespace dnmf

extern "C" {

static long k2_blocks(int X, int Y, int Z, int *nub_out) {
    const long nub = ((long)Y * (Z == 2 ? 1 : Z) + dnmf::K2_COLS - 1) / dnmf::K2_COLS;   // Z == 2: a lane owns both slices
    if (nub_out) *nub_out = (int)nub;
    return nub * ((X + dnmf::K2_ROWS - 1) / dnmf::K2_ROWS);
}

static size_t k2_xtab_bytes(int X) { return ((size_t)X * sizeof(float2) + 255) / 256 * 256; }

size_t dnmf_warp_recon_grad_workspace(int X, int Y, int Z, int B) {
    if (X <= 0 || Y <= 0 || Z <= 0 || B <= 0) return 0;
    return k2_xtab_bytes(X) + (size_t)B * k2_blocks(X, Y, Z, nullptr) * dnmf::K2_NACC * sizeof(float) +
           (size_t)B * sizeof(float);
}

long dnmf_halo_voxels(int X, int Y, int Z) {
    if (X <= 0 || Y <= 0 || Z <= 0) return 0;
    return dnmf::make_halo_layout(X, Y, Z).Pp;
}

int dnmf_halo_row(int Y, int Z) {
    if (Y <= 0 || Z <= 0) return 0;
    return dnmf::make_halo_layout(1, Y, Z).rowf;
}

// the main kernel of K2 for B frames: per-block partial sums into `partial` (B, nblk, K2_NACC)
static void k2_launch_main(const float *S, long lds, const int *s_ids, const float *frames, long ldf, const int *frame_ids,
                           const float *gout, const dnmf::Volume &vol, const dnmf::HaloLayout &hl, const float *beta, int T,
                           const int *times, int B, float *recon, float *partial, const float2 *xtab, int nblk, int nub,
                           hipStream_t st) {
    using namespace dnmf;
    const dim3 grid((unsigned)nblk, (unsigned)B);
#define DNMF_K2_LAUNCH(HZ, FD, FO, PL)                                                                                     \
    hipLaunchKernelGGL((warp_recon_grad_kernel<HZ, FD, FO, PL>), grid, dim3(256), 0, st, S, lds, s_ids, frames, ldf,      \
                       frame_ids, gout, vol, hl, beta, T, times, recon, partial, xtab, nub)
#define DNMF_K2_VARIANTS(HZ, PL)                                    \
    if (vol.fastdiv && hl.f32off) DNMF_K2_LAUNCH(HZ, 1, true, PL);  \
    else if (vol.fastdiv) DNMF_K2_LAUNCH(HZ, 1, false, PL);         \
    else DNMF_K2_LAUNCH(HZ, 0, false, PL)
    const bool plain = frames && !gout && !recon;
    if (vol.Z > 2) {
        if (plain) { DNMF_K2_VARIANTS(3, true); } else { DNMF_K2_VARIANTS(3, false); }
    } else if (vol.Z == 2) {
        if (plain) { DNMF_K2_VARIANTS(2, true); } else { DNMF_K2_VARIANTS(2, false); }
    } else {
        if (plain) { DNMF_K2_VARIANTS(1, true); } else { DNMF_K2_VARIANTS(1, false); }
    }
#undef DNMF_K2_VARIANTS
#undef DNMF_K2_LAUNCH
}

// argument checks shared by the two entry points; fills vol / hl / nblk / nub
static int k2_geometry(const char *who, int X, int Y, int Z, dnmf::Volume &vol, dnmf::HaloLayout &hl, int &nblk, int &nub) {
    using namespace dnmf;
    vol = make_volume(X, Y, Z);
    hl = make_halo_layout(X, Y, Z);
    // 32-bit byte offsets into an image; 24-bit multiplies for the tap offsets
    DNMF_REQUIRE(hl.Pp < (1L << 29) && hl.row4 < (1 << 23) && hl.Xp < (1 << 23), DNMF_E_UNSUPPORTED,
                 "%s: volume %dx%dx%d too large for 32-bit tap offsets", who, X, Y, Z);
    const long nblk_l = k2_blocks(X, Y, Z, &nub);
    DNMF_REQUIRE(nblk_l < (1L << 31), DNMF_E_UNSUPPORTED, "%s: %ld blocks per frame", who, nblk_l);
    nblk = (int)nblk_l;
    return DNMF_OK;
}

int dnmf_warp_recon_grad(const float *S, long lds, const int *s_ids, const float *frames, long ldf,
                         const int *frame_ids, const float *gout, int X, int Y, int Z, const float *beta, int T,
                         const int *times, int B, int norm_frames, float *recon, float *grad, float *loss, float *frame_loss, float *reg,
                         void *workspace, size_t workspace_bytes, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(S && (frames || gout) && beta && times && workspace, DNMF_E_NULL,
                 "dnmf_warp_recon_grad: NULL input");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && T > 0 && B > 0 && B <= 65535, DNMF_E_SHAPE,
                 "dnmf_warp_recon_grad: X=%d Y=%d Z=%d T=%d B=%d", X, Y, Z, T, B);
    Volume vol;
    HaloLayout hl;
    int nblk = 0, nub = 0;
    const int rc = k2_geometry("dnmf_warp_recon_grad", X, Y, Z, vol, hl, nblk, nub);
    if (rc != 0) return rc;
    DNMF_REQUIRE(lds >= hl.Pp && (!frames || ldf >= vol.P), DNMF_E_SHAPE,
                 "dnmf_warp_recon_grad: lds=%ld < %ld (halo layout) or ldf=%ld < P=%ld", lds, hl.Pp, ldf, vol.P);
    DNMF_REQUIRE(workspace_bytes >= dnmf_warp_recon_grad_workspace(X, Y, Z, B), DNMF_E_WORKSPACE,
                 "dnmf_warp_recon_grad: workspace %zu < %zu bytes", workspace_bytes,
                 dnmf_warp_recon_grad_workspace(X, Y, Z, B));
    float2 *xtab = static_cast<float2 *>(workspace);
    float *partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + k2_xtab_bytes(X));
    float *fl = frame_loss ? frame_loss : partial + (size_t)B * nblk * K2_NACC;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k2_xtab_kernel, dim3((unsigned)((X + 255) / 256)), dim3(256), 0, st, xtab, X);
    k2_launch_main(S, lds, s_ids, frames, ldf, frame_ids, gout, vol, hl, beta, T, times, B, recon, partial, xtab, nblk, nub, st);
    if (norm_frames <= 0) norm_frames = B;
    const float grad_scale = gout ? 1.0f : 2.0f / ((float)norm_frames * (float)vol.P);
    hipLaunchKernelGGL(warp_recon_grad_finish_kernel, dim3((unsigned)B), dim3(64), 0, st, partial, nblk, vol, beta, T,
                       times, norm_frames, grad_scale, grad, (loss || frame_loss) ? fl : nullptr, reg);
    if (loss) hipLaunchKernelGGL(sum_loss_kernel, dim3(1), dim3(64), 0, st, fl, B, loss);
    return check_launch("dnmf_warp_recon_grad");
}

// ---- motion gradient with the reconstruction images kept in the last-level cache ---------------------------------
// The fused epoch of update_motion needs, for every frame, S_t = A.C_t (written by dnmf_recon_image_lists) and then
// K2's gather from it.  Done for all T frames at once, S makes a round trip through HBM (4.5 GB written and read back
// at 512x512x4000).  Done `chunk` frames at a time into ONE buffer of `chunk` images, the images are still in the
// 256 MB Infinity Cache when K2 gathers from them, and the next piece overwrites them there before they are ever
// written back.  Only the order of the launches changes: the reconstruction kernel and K2's main kernel alternate, K2's
// finish kernel runs once over all frames at the end; every kernel and every sum is the one of the two separate calls.
static size_t motion_images_bytes(int X, int Y, int Z, int chunk) {
    return ((size_t)dnmf_halo_voxels(X, Y, Z) * chunk * sizeof(float) + 255) / 256 * 256;
}

size_t dnmf_motion_grad_lists_workspace(int X, int Y, int Z, int chunk, int B) {
    if (X <= 0 || Y <= 0 || Z <= 0 || chunk <= 0 || B <= 0) return 0;
    return motion_images_bytes(X, Y, Z, chunk) + k2_xtab_bytes(X) +
           (size_t)B * k2_blocks(X, Y, Z, nullptr) * dnmf::K2_NACC * sizeof(float) + (size_t)B * sizeof(float);
}

int dnmf_motion_grad_lists(const float *At, const int *bbox, int K, const float *C, long ldc, const float *frames, long ldf,
                           const int *frame_ids, int X, int Y, int Z, const float *beta, int T, const int *times, int B,
                           int norm_frames, float *grad, float *frame_loss, float *reg, int chunk, void *workspace,
                           size_t workspace_bytes, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(At && bbox && C && frames && beta && times && grad && workspace, DNMF_E_NULL,
                 "dnmf_motion_grad_lists: NULL buffer");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && K > 0 && T > 0 && B > 0 && norm_frames > 0 && chunk > 0 && chunk <= 65535,
                 DNMF_E_SHAPE, "dnmf_motion_grad_lists: X=%d Y=%d Z=%d K=%d T=%d B=%d norm_frames=%d chunk=%d", X, Y, Z, K, T, B,
                 norm_frames, chunk);
    Volume vol;
    HaloLayout hl;
    int nblk = 0, nub = 0;
    int rc = k2_geometry("dnmf_motion_grad_lists", X, Y, Z, vol, hl, nblk, nub);
    if (rc != 0) return rc;
    DNMF_REQUIRE(ldf >= vol.P, DNMF_E_SHAPE, "dnmf_motion_grad_lists: ldf=%ld < P=%ld", ldf, vol.P);
    DNMF_REQUIRE(workspace_bytes >= dnmf_motion_grad_lists_workspace(X, Y, Z, chunk, B), DNMF_E_WORKSPACE,
                 "dnmf_motion_grad_lists: workspace %zu < %zu bytes", workspace_bytes,
                 dnmf_motion_grad_lists_workspace(X, Y, Z, chunk, B));
    float *S = static_cast<float *>(workspace);
    char *at = static_cast<char *>(workspace) + motion_images_bytes(X, Y, Z, chunk);
    float2 *xtab = reinterpret_cast<float2 *>(at);
    float *partial = reinterpret_cast<float *>(at + k2_xtab_bytes(X));
    float *fl = frame_loss ? frame_loss : partial + (size_t)B * nblk * K2_NACC;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k2_xtab_kernel, dim3((unsigned)((X + 255) / 256)), dim3(256), 0, st, xtab, X);
    for (int c0 = 0; c0 < B; c0 += chunk) {
        const int n = B - c0 < chunk ? B - c0 : chunk;
        rc = dnmf_recon_image_lists(At, bbox, K, X, Y, Z, C, ldc, times + c0, n, S, hl.Pp, stream);
        if (rc != 0) return rc;
        k2_launch_main(S, hl.Pp, nullptr, frame_ids ? frames : frames + (long)c0 * ldf, ldf, frame_ids ? frame_ids + c0 : nullptr,
                       nullptr, vol, hl, beta, T, times + c0, n, nullptr, partial + (size_t)c0 * nblk * K2_NACC, xtab, nblk, nub,
                       st);
    }
    const float grad_scale = 2.0f / ((float)norm_frames * (float)vol.P);
    hipLaunchKernelGGL(warp_recon_grad_finish_kernel, dim3((unsigned)B), dim3(64), 0, st, partial, nblk, vol, beta, T, times,
                       norm_frames, grad_scale, grad, frame_loss ? fl : nullptr, reg);
    return check_launch("dnmf_motion_grad_lists");
}

}  // extern "C"
