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
#include "common.hpp"

namespace dnmf {

constexpr int K2_ROWS = 16;           // voxels per lane: consecutive x
constexpr int K2_COLS = 256;          // positions of the (y,z) plane per block: 64 lanes x 4 waves
constexpr int K2_NACC = 32;           // 30 gradient sums + squared error + pad
#ifndef DNMF_K2_UNROLL
#define DNMF_K2_UNROLL 2
#endif
constexpr int K2_UNROLL = DNMF_K2_UNROLL;  // rows requested together
static_assert(K2_ROWS % K2_UNROLL == 0, "the row loop takes K2_UNROLL rows at a time");

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Work layout: a block owns 16 x-rows by 256 consecutive positions of the (y,z) plane (lane = position, so the frame
// and the reconstruction image are read in 256-byte runs and the taps of neighbouring lanes share cache lines); a
// thread walks down its 16 rows with (y,z) fixed: the monomials without x are per-thread constants, and x, x^2 are the
// same for the whole block: they come from a table (scalar loads).  The polynomial itself is the reference's chain of
// ten (six for Z = 1) fused multiply-adds, in its order (common.hpp: poly_a).
//
// S carries the zero halo of common.hpp around x and y: a tap outside the volume reads a zero instead of being
// masked, for the value and for the gradient alike (grid_sample's backward skips out-of-bounds corners: it adds the
// same zeros).  Per voxel that leaves: two FMAs per coordinate, the fp32 normalise / un-normalise round trip of the
// reference, one clamp + floor + two weights per axis, one base offset, the taps, the blends.
//
// HASZ = false is the Z == 1 specialisation: two coordinates, four taps, and only the six basis terms without z
// (the other 18 gradient sums are identically zero and are written as such).  PLAIN = the fit step's call (frames
// given, no upstream gradient, A_tC not wanted): the row loop then has no branches.
template <bool HASZ, int FAST, bool F32OFF, bool PLAIN>
__global__ __launch_bounds__(256) void warp_recon_grad_kernel(const float *__restrict__ S, long lds,
                                                              const int *__restrict__ s_ids,
                                                              const float *__restrict__ frames, long ldf,
                                                              const int *__restrict__ frame_ids,
                                                              const float *__restrict__ gout, Volume vol, HaloLayout hl,
                                                              const float *__restrict__ beta, int T,
                                                              const int *__restrict__ times,
                                                              float *__restrict__ recon, float *__restrict__ partial,
                                                              const float2 *__restrict__ xtab, int nub) {
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
    // accumulated: mom[m][d] = sum_i (resid g_d)_i x_i^m, m = 0, 1, 2; the ten basis sums follow from them afterwards
    float mom[3][ND];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int d = 0; d < ND; ++d) mom[m][d] = 0.0f;
    float sq = 0.0f;
    const int YZ = vol.Y * vol.Z;
    const int bu = blockIdx.x % nub, bx = blockIdx.x / nub;
    const int u = bu * K2_COLS + threadIdx.x;          // position in the (y,z) plane
    const int yy = HASZ ? div_small(u, vol.Z, vol.rcp_z) : u;
    const int z = HASZ ? u - yy * vol.Z : 0;
    const float yf = (float)yy, zf = (float)z;

    if (u < YZ) {
        float b2[30];
        double_beta(bt, b2);
        Monomials<HASZ> mono = monomials<HASZ>(0.0f, yf, zf);   // y, z, y^2, z^2, yz: fixed along the rows
        const int x_first = bx * K2_ROWS;
        const int nrow = min(K2_ROWS, vol.X - x_first);

        // A row is handled in two steps so that the loop below can keep the taps of the next rows in flight while
        // it blends the current ones: request() = coordinates, weights, tap offsets, loads; consume() = the blends,
        // the residual and the moment sums.
        struct Req {
            float2 xv;                    // (x, x*x): block-uniform
            float sv[HASZ ? 2 : 1][2][2];  // [dz][dy][dx]
            float wx[2], wy[2], wzm[2], vz[2];
            float other;                  // frame value, or the caller's upstream gradient
            long prow;
        };
        unsigned u4 = (unsigned)u * 4u;
        auto request = [&](int x, Req &q) {
            q.xv = xtab[x];                   // a scalar load
            q.prow = (long)x * YZ;            // block-uniform: (scalar base + 32-bit lane offset) accesses
            Monomials<HASZ> m = mono;
            m.x = q.xv.x, m.xx = q.xv.y, m.xy = __fmul_rn(q.xv.x, yf);
            if (HASZ) m.xz = __fmul_rn(q.xv.x, zf);
            float fx, fy;
            axis_taps_halo(unnormalise(normalise_axis<FAST>(poly_a<HASZ>(b2, 0, m), vol, 0), vol.hx1), hl.xhi, fx, q.wx[0],
                           q.wx[1]);
            axis_taps_halo(unnormalise(normalise_axis<FAST>(poly_a<HASZ>(b2, 1, m), vol, 1), vol.hy1), hl.yhi, fy, q.wy[0],
                           q.wy[1]);
            const unsigned o0 = halo_offset<F32OFF>(fx, fy, hl, hl.origin4, hl.origin4f);   // base corner, slice 0
            int iz = 0;
            if (HASZ) {
                float wz[2];
                axis_weights(unnormalise(normalise_axis<FAST>(poly_a<HASZ>(b2, 2, m), vol, 2), vol.hz1), iz, wz[0], wz[1]);
#pragma unroll
                for (int dz = 0; dz < 2; ++dz) {
                    const bool zin = in_range(iz + dz, vol.Z);
                    q.wzm[dz] = zin ? wz[dz] : 0.0f;
                    q.vz[dz] = zin ? (dz ? 1.0f : -1.0f) : 0.0f;
                }
            }
#pragma unroll
            for (int dz = 0; dz < (HASZ ? 2 : 1); ++dz) {
                const unsigned zo = HASZ ? (unsigned)clamp_index(iz + dz, vol.Z) * 4u : 0u;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    unsigned o = o0 + zo + (dx ? (unsigned)hl.row4 : 0u);   // (scalar base + 32-bit offset) loads
                    asm("" : "+v"(o));
                    const char *t = s + o;
                    q.sv[dz][0][dx] = *reinterpret_cast<const float *>(t);
                    q.sv[dz][1][dx] = *reinterpret_cast<const float *>(t + (HASZ ? hl.col4 : 4));
                }
            }
            asm("" : "+v"(u4));
            q.other = *reinterpret_cast<const float *>(reinterpret_cast<const char *>((!PLAIN && go ? go : y) + q.prow) + u4);
        };
        auto consume = [&](const Req &q) {
            float rec = 0.0f, g[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int dz = 0; dz < (HASZ ? 2 : 1); ++dz) {
                const float a0 = fmaf(q.wx[1], q.sv[dz][0][1], q.wx[0] * q.sv[dz][0][0]);  // x-interpolated rows y0, y1
                const float a1 = fmaf(q.wx[1], q.sv[dz][1][1], q.wx[0] * q.sv[dz][1][0]);
                const float d0 = q.sv[dz][0][1] - q.sv[dz][0][0];                          // x-differences of the rows
                const float d1 = q.sv[dz][1][1] - q.sv[dz][1][0];
                const float r2 = fmaf(q.wy[1], a1, q.wy[0] * a0);                          // value of this z-slice
                const float gx2 = fmaf(q.wy[1], d1, q.wy[0] * d0);
                const float gy2 = a1 - a0;
                if (HASZ) {
                    rec = fmaf(q.wzm[dz], r2, rec);
                    g[0] = fmaf(q.wzm[dz], gx2, g[0]);
                    g[1] = fmaf(q.wzm[dz], gy2, g[1]);
                    g[2] = fmaf(q.vz[dz], r2, g[2]);
                } else {
                    rec = r2, g[0] = gx2, g[1] = gy2;
                }
            }
            if (!PLAIN && rc) *reinterpret_cast<float *>(reinterpret_cast<char *>(rc + q.prow) + u4) = rec;
            // upstream gradient: the mse residual (scaled by 2/(B P) in the finish kernel) or the caller's
            const float resid = !PLAIN && go ? q.other : rec - q.other;
            sq = fmaf(resid, resid, sq);
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const float gd = resid * g[d];
                mom[0][d] += gd;
                mom[1][d] = fmaf(q.xv.x, gd, mom[1][d]);
                mom[2][d] = fmaf(q.xv.y, gd, mom[2][d]);
            }
        };
        if (nrow == K2_ROWS) {
            // K2_UNROLL rows at a time: their taps are requested together, then blended (keeping the next rows' taps in
            // flight behind the blends of the current ones did not pay: 2.07 against 1.99 ms, tools/time_k2.py)
#pragma unroll 1
            for (int i = 0; i < K2_ROWS; i += K2_UNROLL) {
                Req q[K2_UNROLL];
#pragma unroll
                for (int j = 0; j < K2_UNROLL; ++j) request(x_first + i + j, q[j]);
#pragma unroll
                for (int j = 0; j < K2_UNROLL; ++j) consume(q[j]);
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
        acc[0][d] = mom[0][d], acc[1][d] = mom[1][d], acc[2][d] = yf * mom[0][d], acc[3][d] = mom[2][d];
        acc[4][d] = (yf * yf) * mom[0][d], acc[5][d] = yf * mom[1][d];
        if (HASZ) {
            acc[6][d] = zf * mom[0][d], acc[7][d] = (zf * zf) * mom[0][d], acc[8][d] = zf * mom[1][d];
            acc[9][d] = (yf * zf) * mom[0][d];
        }
    }

    // block reduction: butterflies inside each wave, then the four wave leaders through LDS
    __shared__ float red[4][K2_NACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < K2_NACC) red[wave][lane] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const float v = wave_sum(acc[a][d]);
            if (lane == 0) red[wave][BASIS_ID[a] * 3 + d] = v;
        }
    {
        const float v = wave_sum(sq);
        if (lane == 0) red[wave][30] = v;
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
    const long nub = ((long)Y * Z + dnmf::K2_COLS - 1) / dnmf::K2_COLS;
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
    if (vol.Z > 1) {
        if (plain) { DNMF_K2_VARIANTS(true, true); } else { DNMF_K2_VARIANTS(true, false); }
    } else {
        if (plain) { DNMF_K2_VARIANTS(false, true); } else { DNMF_K2_VARIANTS(false, false); }
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
