// K7 -- registered video Y_i: nearest-neighbour inverse warp of each frame.
//
// Reference: ExponentialFP.spatial_pushforward + image_iwarp, Demix/dNMF.py:81-83, 89-91, 95-103: the warped
// position of every voxel is un-normalised with sz[d] (not sz[d]-1), scipy's NearestNDInterpolator (cKDTree)
// is built on those positions, and every lattice point takes the frame value of its nearest warped voxel.
// This build searches exhaustively (P candidates per lattice point, tiles of warped positions in LDS):
// exact, O(P^2) per frame, meant for the volumes the dense return value exists for.  Distances are evaluated in
// float64 on the float32 positions like the reference; ties (exactly equal distances) go to the lowest voxel
// index, cKDTree's choice there is unspecified.
#include "common.hpp"

namespace dnmf {

constexpr int IW_TILE = 1024;

__global__ __launch_bounds__(256) void image_iwarp_kernel(const float *__restrict__ frames, long ldf,
                                                          const int *__restrict__ frame_ids, Volume vol,
                                                          const float *__restrict__ beta, int T,
                                                          const int *__restrict__ times, float *__restrict__ out,
                                                          long ldo) {
    __shared__ float sx[IW_TILE], sy[IW_TILE], sz[IW_TILE];
    const int b = blockIdx.y;
    const float *y = frames + (long)(frame_ids ? frame_ids[b] : b) * ldf;
    float bt[30];
    load_beta(beta, T, times[b], bt);
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;  // lattice point served by this thread
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
                const float xf = (float)x, yf = (float)yy, zf = (float)z;
                // ((n + 1) / 2) * sz[d], fp32 like the reference (flow_ is a float32 tensor there)
                sx[i] = __fmul_rn(__fmul_rn(__fadd_rn(normalise_axis(poly_q(bt, 0, xf, yf, zf), vol, 0), 1.0f), 0.5f), (float)vol.X);
                sy[i] = __fmul_rn(__fmul_rn(__fadd_rn(normalise_axis(poly_q(bt, 1, xf, yf, zf), vol, 1), 1.0f), 0.5f), (float)vol.Y);
                sz[i] = vol.Z > 1 ? __fmul_rn(__fmul_rn(__fadd_rn(normalise_axis(poly_q(bt, 2, xf, yf, zf), vol, 2), 1.0f), 0.5f), (float)vol.Z)
                                  : 0.0f;
            }
        }
        __syncthreads();
        const int n = (int)((vol.P - v0) < IW_TILE ? (vol.P - v0) : IW_TILE);
        for (int i = 0; i < n; ++i) {
            const double dx = (double)sx[i] - gx, dy = (double)sy[i] - gy, dz = (double)sz[i] - gz;
            const double d = dx * dx + dy * dy + dz * dz;
            if (d < best) best = d, arg = v0 + i;
        }
    }
    if (g < vol.P) out[(long)b * ldo + g] = y[arg];
}

}  // namespace dnmf

extern "C" int dnmf_image_iwarp(const float *frames, long ldf, const int *frame_ids, int X, int Y, int Z,
                                const float *beta, int T, const int *times, int B, float *out, long ldo,
                                dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(frames && beta && times && out, DNMF_E_NULL, "dnmf_image_iwarp: NULL buffer");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && T > 0 && B > 0 && B <= 65535, DNMF_E_SHAPE,
                 "dnmf_image_iwarp: X=%d Y=%d Z=%d T=%d B=%d", X, Y, Z, T, B);
    const Volume vol = make_volume(X, Y, Z);
    DNMF_REQUIRE(ldf >= vol.P && ldo >= vol.P, DNMF_E_SHAPE, "dnmf_image_iwarp: ldf=%ld ldo=%ld < P=%ld", ldf, ldo, vol.P);
    DNMF_REQUIRE(vol.P <= (1L << 20), DNMF_E_UNSUPPORTED,
                 "dnmf_image_iwarp: P=%ld: the exhaustive search is limited to 2^20 voxels", vol.P);
    hipLaunchKernelGGL(image_iwarp_kernel, dim3((unsigned)((vol.P + 255) / 256), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, frames, ldf, frame_ids, vol, beta, T, times, out, ldo);
    return check_launch("dnmf_image_iwarp");
}
