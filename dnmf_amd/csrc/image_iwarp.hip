// K7 -- registered video Y_i: nearest-neighbour inverse warp of each frame.
//
// Reference: ExponentialFP.spatial_pushforward + image_iwarp, Demix/dNMF.py:81-83, 89-91, 95-103: the warped
// position of every voxel is un-normalised with sz[d] (not sz[d]-1), scipy's NearestNDInterpolator (cKDTree)
// is built on those positions, and every lattice point takes the frame value of its nearest warped voxel.
// Distances are evaluated in float64 on the float32 positions like the reference; ties (exactly equal distances) go
// to the lowest voxel index, cKDTree's choice there is unspecified.
//
// Search.  The warped position s(v) is a quadratic polynomial of the voxel v, so s(a) - s(b) = J((a+b)/2) (a - b)
// EXACTLY, J the Jacobian; with m a lower bound of the smallest singular value of J over the volume,
// |s(a) - s(b)| >= m |a - b|.  For a lattice point g and ANY voxel v0 at distance d0 = |s(v0) - g| the nearest voxel
// v* has |s(v*) - g| <= d0, hence |v* - v0| <= 2 d0 / m: it lies in the window of that radius around v0.  The kernel
// gets v0 from a few fixed-point steps of v <- v + (g - s(v)) (S-1)/S, searches that window (49 candidates for a
// near-identity warp of a 2-D volume instead of all P) and marks the lattice points whose window would exceed
// IW_RMAX -- or for which m <= 0: a warp that folds -- for the exhaustive kernel, which is the O(P) per point search
// of round 1 and now runs only for those.  Both kernels evaluate s(v) with the same fp32 sequence and compare the
// same float64 distances, so the result is the exhaustive search's bit for bit (the rounding of s in fp32 is
// covered by a slack in the radius).
#include "common.hpp"

namespace dnmf {

constexpr int IW_TILE = 1024;
constexpr int IW_RMAX = 24;   // largest window radius searched in place (49 x 49 (x Z) candidates)

// warped position of voxel (x,y,z) scaled by sz (not sz-1), fp32 like the reference (flow_ is a float32 tensor there)
__device__ __forceinline__ void iwarp_position(const float *bt, const Volume &vol, int x, int y, int z, float &sx, float &sy,
                                               float &sz) {
    const float xf = (float)x, yf = (float)y, zf = (float)z;
    const bool hz = vol.Z > 1;
    const float nx = hz ? grid_n<true>(bt, vol, 0, xf, yf, zf) : grid_n<false>(bt, vol, 0, xf, yf, 0.0f);
    const float ny = hz ? grid_n<true>(bt, vol, 1, xf, yf, zf) : grid_n<false>(bt, vol, 1, xf, yf, 0.0f);
    sx = __fmul_rn(__fmul_rn(__fadd_rn(nx, 1.0f), 0.5f), (float)vol.X);
    sy = __fmul_rn(__fmul_rn(__fadd_rn(ny, 1.0f), 0.5f), (float)vol.Y);
    sz = hz ? __fmul_rn(__fmul_rn(__fadd_rn(grid_n<true>(bt, vol, 2, xf, yf, zf), 1.0f), 0.5f), (float)vol.Z) : 0.0f;
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

// One thread per lattice point: seed, window search; todo[g] = 1 where the window would be too large.
__global__ __launch_bounds__(256) void image_iwarp_window_kernel(const float *__restrict__ frames, long ldf,
                                                                 const int *__restrict__ frame_ids, Volume vol,
                                                                 const float *__restrict__ beta, int T,
                                                                 const int *__restrict__ times, float *__restrict__ out,
                                                                 long ldo, unsigned char *__restrict__ todo) {
    const int b = blockIdx.y;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= vol.P) return;
    const float *y = frames + (long)(frame_ids ? frame_ids[b] : b) * ldf;
    float bt[30];
    load_beta(beta, T, times[b], bt);
    const double m = iwarp_min_stretch(bt, vol);
    unsigned char *flag = todo + (long)b * vol.P + g;
    if (!(m > 1e-3)) {
        *flag = 1;
        return;
    }
    int gx, gy, gz;
    voxel_xyz(g, vol, gx, gy, gz);
    const bool hz = vol.Z > 1;
    // seed: fixed-point steps on the continuous map, then the nearest voxel inside the volume
    const float kx = vol.sx1 / (float)vol.X, ky = vol.sy1 / (float)vol.Y, kz = hz ? vol.sz1 / (float)vol.Z : 1.0f;
    float vx = gx * kx, vy = gy * ky, vz = hz ? gz * kz : 0.0f;
    for (int it = 0; it < 4; ++it) {
        float s[3] = {0.0f, 0.0f, 0.0f};
        float b2[30];
        double_beta(bt, b2);
        for (int d = 0; d < (hz ? 3 : 2); ++d) {
            const float a = hz ? poly_a<true>(b2, d, monomials<true>(vx, vy, vz)) : poly_a<false>(b2, d, monomials<false>(vx, vy, 0.0f));
            s[d] = 0.5f * a / (d == 0 ? kx : (d == 1 ? ky : kz));   // q S/(S-1)
        }
        vx += (gx - s[0]) * kx, vy += (gy - s[1]) * ky;
        if (hz) vz += (gz - s[2]) * kz;
    }
    const int x0 = min(max((int)rintf(fminf(fmaxf(vx, -1.0f), (float)vol.X)), 0), vol.X - 1);
    const int y0 = min(max((int)rintf(fminf(fmaxf(vy, -1.0f), (float)vol.Y)), 0), vol.Y - 1);
    const int z0 = hz ? min(max((int)rintf(fminf(fmaxf(vz, -1.0f), (float)vol.Z)), 0), vol.Z - 1) : 0;
    float sx, sy, sz;
    iwarp_position(bt, vol, x0, y0, z0, sx, sy, sz);
    const double ex = (double)sx - gx, ey = (double)sy - gy, ez = (double)sz - gz;
    const double d0 = sqrt(ex * ex + ey * ey + ez * ez);
    // fp32 rounding of the positions: a few units in the last place at magnitudes up to the volume size
    const double eps = 1e-4 + 1e-5 * (double)max(vol.X, max(vol.Y, vol.Z));
    const double rr = (2.0 * d0 + 2.0 * eps) / m;
    if (!(rr < (double)IW_RMAX)) {   // also NaN
        *flag = 1;
        return;
    }
    const int R = (int)rr + 1;
    double best = 1e300;
    long arg = 0;
    const int xa = max(x0 - R, 0), xb = min(x0 + R, vol.X - 1);
    const int ya = max(y0 - R, 0), yb = min(y0 + R, vol.Y - 1);
    const int za = max(z0 - R, 0), zb = min(z0 + R, vol.Z - 1);
    for (int x = xa; x <= xb; ++x)           // ascending voxel index: a tie goes to the lowest, as in the full search
        for (int yy = ya; yy <= yb; ++yy)
            for (int z = za; z <= zb; ++z) {
                iwarp_position(bt, vol, x, yy, z, sx, sy, sz);
                const double dx = (double)sx - gx, dy = (double)sy - gy, dz = (double)sz - gz;
                const double d = dx * dx + dy * dy + dz * dz;
                if (d < best) best = d, arg = ((long)x * vol.Y + yy) * vol.Z + z;
            }
    out[(long)b * ldo + g] = y[arg];
    *flag = 0;
}

// The exhaustive search (P candidates per lattice point, tiles of warped positions in LDS) for the marked points.
__global__ __launch_bounds__(256) void image_iwarp_full_kernel(const float *__restrict__ frames, long ldf,
                                                               const int *__restrict__ frame_ids, Volume vol,
                                                               const float *__restrict__ beta, int T,
                                                               const int *__restrict__ times, float *__restrict__ out,
                                                               long ldo, const unsigned char *__restrict__ todo) {
    __shared__ float sx[IW_TILE], sy[IW_TILE], sz[IW_TILE];
    const int b = blockIdx.y;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;  // lattice point served by this thread
    const bool mine = g < vol.P && todo[(long)b * vol.P + g];
    if (!__syncthreads_or(mine)) return;                         // nothing marked in this block
    const float *y = frames + (long)(frame_ids ? frame_ids[b] : b) * ldf;
    float bt[30];
    load_beta(beta, T, times[b], bt);
    int gx = 0, gy = 0, gz = 0;
    if (g < vol.P) voxel_xyz(g, vol, gx, gy, gz);
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
}

__global__ void count_flags_kernel(const unsigned char *__restrict__ todo, long n, unsigned long long *__restrict__ count) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long m = __ballot(i < n && todo[i]);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__builtin_popcountll(m));
}

}  // namespace dnmf

extern "C" {

size_t dnmf_image_iwarp_workspace(int X, int Y, int Z, int B) {
    if (X <= 0 || Y <= 0 || Z <= 0 || B <= 0) return 0;
    return (size_t)X * Y * Z * B;
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
    if (exhaustive) {
        hipError_t e = hipMemsetAsync(todo, 1, (size_t)vol.P * B, st);
        DNMF_REQUIRE(e == hipSuccess, (int)e, "dnmf_image_iwarp: hipMemsetAsync: %s", hipGetErrorString(e));
    } else {
        hipLaunchKernelGGL(image_iwarp_window_kernel, grid, dim3(256), 0, st, frames, ldf, frame_ids, vol, beta, T, times, out,
                           ldo, todo);
    }
    if (fallback_count) {
        const long n = vol.P * B;
        hipLaunchKernelGGL(count_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, todo, n, fallback_count);
    }
    hipLaunchKernelGGL(image_iwarp_full_kernel, grid, dim3(256), 0, st, frames, ldf, frame_ids, vol, beta, T, times, out, ldo,
                       todo);
    return check_launch("dnmf_image_iwarp");
}

}  // extern "C"
