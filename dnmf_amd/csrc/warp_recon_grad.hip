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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Work layout: a block owns 16 x-rows by 256 consecutive positions of the (y,z) plane (lane = position, so the frame
// and the reconstruction image are read in 256-byte runs and the taps of neighbouring lanes share cache lines); a
// thread walks down its 16 rows.  x, y, z follow from the block and lane indices by additions.
//
// HASZ = false is the Z == 1 specialisation: two coordinates, four taps, and only the six basis terms without z
// (the other 18 gradient sums are identically zero and are written as such).
template <bool HASZ, int FAST>
__global__ __launch_bounds__(256) void warp_recon_grad_kernel(const float *__restrict__ S, long lds,
                                                              const int *__restrict__ s_ids,
                                                              const float *__restrict__ frames, long ldf,
                                                              const int *__restrict__ frame_ids,
                                                              const float *__restrict__ gout, Volume vol,
                                                              const float *__restrict__ beta, int T,
                                                              const int *__restrict__ times,
                                                              float *__restrict__ recon, float *__restrict__ partial,
                                                              int nub) {
    constexpr int ND = HASZ ? 3 : 2;                 // warped coordinates that exist
    constexpr int NA = HASZ ? 10 : 6;                // basis terms that are not identically zero
    constexpr int BASIS_ID[10] = {0, 1, 2, 4, 5, 7, 3, 6, 8, 9};  // z-free terms first
    const int b = blockIdx.y;
    const char *__restrict__ s = reinterpret_cast<const char *>(S + (long)(s_ids ? s_ids[b] : b) * lds);
    const float *__restrict__ y = frames ? frames + (long)(frame_ids ? frame_ids[b] : b) * ldf : nullptr;
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
        const int nrow = min(K2_ROWS, vol.X - bx * K2_ROWS);
#pragma unroll 2
        for (int i = 0; i < nrow; ++i) {
            const int x = bx * K2_ROWS + i;
            const long p = (long)x * YZ + u;
            const Sample sm = make_sample_t<HASZ, FAST>(bt, vol, x, yy, z);
            // Branch-free gather: corner indices clamped into the volume, per-axis weights (w) and validity flags (v)
            // zeroed for corners outside it.  A corner's value weight is wx*wy*wz and its x-derivative weight
            // +-vx*wy*wz (ATen's grid_sampler backward skips out-of-bounds corners), so everything factorises per
            // axis.
            const float wx[2] = {in_range(sm.x0, vol.X) ? sm.wx0 : 0.0f, in_range(sm.x0 + 1, vol.X) ? sm.wx1 : 0.0f};
            const float wy[2] = {in_range(sm.y0, vol.Y) ? sm.wy0 : 0.0f, in_range(sm.y0 + 1, vol.Y) ? sm.wy1 : 0.0f};
            const float vx[2] = {in_range(sm.x0, vol.X) ? 1.0f : 0.0f, in_range(sm.x0 + 1, vol.X) ? 1.0f : 0.0f};
            const float vy[2] = {in_range(sm.y0, vol.Y) ? 1.0f : 0.0f, in_range(sm.y0 + 1, vol.Y) ? 1.0f : 0.0f};
            const unsigned xo[2] = {(unsigned)(clamp_index(sm.x0, vol.X) * YZ), (unsigned)(clamp_index(sm.x0 + 1, vol.X) * YZ)};
            const unsigned yo[2] = {(unsigned)(clamp_index(sm.y0, vol.Y) * vol.Z),
                                    (unsigned)(clamp_index(sm.y0 + 1, vol.Y) * vol.Z)};
            float rec = 0.0f, g[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int dz = 0; dz < (HASZ ? 2 : 1); ++dz) {
                const unsigned zo = HASZ ? (unsigned)clamp_index(sm.z0 + dz, vol.Z) : 0u;
                float sv[2][2];  // [dy][dx]
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        unsigned o = (xo[dx] + yo[dy] + zo) * 4u;  // byte offset: (scalar base + 32-bit offset) loads
                        asm("" : "+v"(o));
                        sv[dy][dx] = *reinterpret_cast<const float *>(s + o);
                    }
                const float a0 = fmaf(wx[1], sv[0][1], wx[0] * sv[0][0]);  // x-interpolated rows y0, y1
                const float a1 = fmaf(wx[1], sv[1][1], wx[0] * sv[1][0]);
                const float d0 = fmaf(vx[1], sv[0][1], -(vx[0] * sv[0][0]));  // x-differences of the rows
                const float d1 = fmaf(vx[1], sv[1][1], -(vx[0] * sv[1][0]));
                const float r2 = fmaf(wy[1], a1, wy[0] * a0);       // value of this z-slice
                const float gx2 = fmaf(wy[1], d1, wy[0] * d0);
                const float gy2 = fmaf(vy[1], a1, -(vy[0] * a0));
                if (HASZ) {
                    const bool zin = in_range(sm.z0 + dz, vol.Z);
                    const float wz = zin ? (dz ? sm.wz1 : sm.wz0) : 0.0f;
                    const float vz = zin ? (dz ? 1.0f : -1.0f) : 0.0f;
                    rec = fmaf(wz, r2, rec);
                    g[0] = fmaf(wz, gx2, g[0]);
                    g[1] = fmaf(wz, gy2, g[1]);
                    g[2] = fmaf(vz, r2, g[2]);
                } else {
                    rec = r2, g[0] = gx2, g[1] = gy2;
                }
            }
            if (recon) recon[(long)b * vol.P + p] = rec;
            // upstream gradient: the mse residual (scaled by 2/(B P) in the finish kernel) or the caller's
            const float resid = gout ? gout[(long)b * vol.P + p] : rec - y[p];
            sq = fmaf(resid, resid, sq);
            const float xf = (float)x, xx = xf * xf;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const float gd = resid * g[d];
                mom[0][d] += gd;
                mom[1][d] = fmaf(xf, gd, mom[1][d]);
                mom[2][d] = fmaf(xx, gd, mom[2][d]);
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

size_t dnmf_warp_recon_grad_workspace(int X, int Y, int Z, int B) {
    if (X <= 0 || Y <= 0 || Z <= 0 || B <= 0) return 0;
    return (size_t)B * k2_blocks(X, Y, Z, nullptr) * dnmf::K2_NACC * sizeof(float) + (size_t)B * sizeof(float);
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
    const Volume vol = make_volume(X, Y, Z);
    DNMF_REQUIRE(lds >= vol.P && (!frames || ldf >= vol.P), DNMF_E_SHAPE, "dnmf_warp_recon_grad: lds=%ld ldf=%ld < P=%ld", lds,
                 ldf, vol.P);
    DNMF_REQUIRE(workspace_bytes >= dnmf_warp_recon_grad_workspace(X, Y, Z, B), DNMF_E_WORKSPACE,
                 "dnmf_warp_recon_grad: workspace %zu < %zu bytes", workspace_bytes,
                 dnmf_warp_recon_grad_workspace(X, Y, Z, B));
    DNMF_REQUIRE(vol.P < (1L << 30), DNMF_E_UNSUPPORTED, "dnmf_warp_recon_grad: P=%ld does not fit 32-bit byte offsets",
                 vol.P);
    int nub = 0;
    const long nblk_l = k2_blocks(X, Y, Z, &nub);
    DNMF_REQUIRE(nblk_l < (1L << 31), DNMF_E_UNSUPPORTED, "dnmf_warp_recon_grad: %ld blocks per frame", nblk_l);
    const int nblk = (int)nblk_l;
    float *partial = static_cast<float *>(workspace);
    float *fl = frame_loss ? frame_loss : partial + (size_t)B * nblk * K2_NACC;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)nblk, (unsigned)B);
#define DNMF_K2_LAUNCH(HZ, FD)                                                                                            \
    hipLaunchKernelGGL((warp_recon_grad_kernel<HZ, FD>), grid, dim3(256), 0, st, S, lds, s_ids, frames, ldf, frame_ids, \
                       gout, vol, beta, T, times, recon, partial, nub)
    if (Z > 1) {
        if (vol.fastdiv) DNMF_K2_LAUNCH(true, 1); else DNMF_K2_LAUNCH(true, 0);
    } else {
        if (vol.fastdiv) DNMF_K2_LAUNCH(false, 1); else DNMF_K2_LAUNCH(false, 0);
    }
#undef DNMF_K2_LAUNCH
    if (norm_frames <= 0) norm_frames = B;
    const float grad_scale = gout ? 1.0f : 2.0f / ((float)norm_frames * (float)vol.P);
    hipLaunchKernelGGL(warp_recon_grad_finish_kernel, dim3((unsigned)B), dim3(64), 0, st, partial, nblk, vol, beta, T,
                       times, norm_frames, grad_scale, grad, (loss || frame_loss) ? fl : nullptr, reg);
    if (loss) hipLaunchKernelGGL(sum_loss_kernel, dim3(1), dim3(64), 0, st, fl, B, loss);
    return check_launch("dnmf_warp_recon_grad");
}

}  // extern "C"
