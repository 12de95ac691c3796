// Reconstruction image S[b,p] = sum_k C[k,t_b] A[p,k] for COMPACT footprints, from the neuron-major copy At and the
// boxes of dnmf_pack_footprints_lists.  At and S are both in the halo layout of common.hpp; the kernel sweeps the
// whole padded image, so the border of S is rewritten (as the zeros the border of At holds) by every call.
//
// Same role as recon_image.hip (the reference's einsum of Demix/dNMF.py:58, taken before the warp because the
// gather is linear).  A tile of 4 rows x 64 positions of the contiguous axis needs only the neurons whose non-zero box
// meets it: a static list.  A lane owns four consecutive positions of one row (16 lanes per row), so the wave keeps the
// footprint values of its voxels as one 16-byte register quad per neuron and sweeps a run of frames: per frame and lane
// 4 x (listed neurons) FMAs and ONE 16-byte store (a wave writes four 256-byte runs with one instruction; four 4-byte
// stores per lane reached 65 % of the HBM roof) -- the kernel is bound by writing S.
//
// Terms with an exact zero factor are the only ones dropped, so S equals the dense product up to the order of the
// fp32 additions (ascending neuron index here).
#include "common.hpp"

namespace dnmf {

typedef float rl_f4 __attribute__((ext_vector_type(4)));
constexpr int RL_NG = 8;  // neurons held in registers at a time

template <int NW>
__global__ __launch_bounds__(256) void recon_lists_kernel(const float *__restrict__ At, const int *__restrict__ bbox,
                                                          int K, Volume vol, HaloLayout hl, const float *__restrict__ C, long ldc,
                                                          const int *__restrict__ times, int B, float *__restrict__ S,
                                                          long lds, int nu, int frames_per_wave, int skip_empty) {
    __shared__ int s_list[4][64 * NW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x;                       // (x group, run of 64 positions of the (y,z) plane)
    const int b0 = (blockIdx.y * 4 + wave) * frames_per_wave;
    if (b0 >= B) return;                               // whole wave leaves; no workgroup barrier below
    const int b1 = min(b0 + frames_per_wave, B);
    const int YZ = hl.rowf;                            // one padded row (a multiple of 32 floats)
    const int qu = tile % nu, qx = tile / nu;
    const int row = lane >> 4;                         // row of the tile this lane works in
    const int u = 64 * qu + 4 * (lane & 15);           // its first position in the padded (y,z) plane (YZ is a multiple of 4)
    const int x0 = 4 * qx;                             // first padded row of the tile
    const bool in = u < YZ && x0 + row < hl.Xp;
    // box of the tile in volume coordinates
    const int ylo = (64 * qu) / vol.Z - HALO, yhi = min(64 * qu + 63, hl.Yp * vol.Z - 1) / vol.Z - HALO;
    const int xlo = x0 - HALO, xhi = min(x0 + 3, hl.Xp - 1) - HALO;

    int *lst = s_list[wave];
    int n = 0;
#pragma unroll
    for (int wd = 0; wd < NW; ++wd) {
        const int k = lane + 64 * wd;
        bool hit = false;
        if (k < K) {
            const int *bb = bbox + k * 6;
            hit = bb[0] <= xhi && bb[1] >= xlo && bb[2] <= yhi && bb[3] >= ylo && bb[4] <= bb[5];
        }
        const unsigned long long m = __ballot(hit);
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (hit) lst[n + before] = k;
        n += __builtin_popcountll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const long voxel = (long)(x0 + row) * YZ + u;
    float *__restrict__ out = S + voxel;
    const rl_f4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (n == 0) {
        // skip_empty: the caller vouches that this tile of every row of S already holds zeros (an earlier call with the same
        // boxes wrote them): about a third of the tiles at 512x512, K = 100 -- of a kernel bound by its stores
        if (skip_empty) return;
        for (int b = b0; b < b1; ++b)
            if (in) *reinterpret_cast<rl_f4 *>(out + (long)b * lds) = zero4;
        return;
    }
    for (int g = 0; g < n; g += RL_NG) {
        // footprint values of this group at the lane's voxels (0 beyond the list / outside the volume)
        const int mine = lst[min(g + (lane & 7), n - 1)];
        int ks[RL_NG];
        rl_f4 a[RL_NG];
#pragma unroll
        for (int i = 0; i < RL_NG; ++i) {
            ks[i] = g + i < n ? __builtin_amdgcn_readlane(mine, i) : -1;
            a[i] = zero4;
            if (ks[i] >= 0 && in) a[i] = *reinterpret_cast<const rl_f4 *>(At + (long)ks[i] * hl.Pp + voxel);
        }
        // frames in runs of 64: lane j fetches the traces of frame bb + j, the run then reads them as scalars
        for (int bb = b0; bb < b1; bb += 64) {
            const int bl = min(bb + lane, b1 - 1);
            const int tcol = times ? times[bl] : bl;
            float cv[RL_NG];
#pragma unroll
            for (int i = 0; i < RL_NG; ++i) cv[i] = ks[i] >= 0 ? C[(long)ks[i] * ldc + tcol] : 0.0f;
            const int nb = min(64, b1 - bb);
            for (int j = 0; j < nb; ++j) {
                rl_f4 s = zero4;
                rl_f4 *__restrict__ dst = reinterpret_cast<rl_f4 *>(out + (long)(bb + j) * lds);
                if (g > 0 && in) s = *dst;
#pragma unroll
                for (int i = 0; i < RL_NG; ++i) {
                    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cv[i]), j));
#pragma unroll
                    for (int v = 0; v < 4; ++v) s[v] = fmaf(a[i][v], c, s[v]);
                }
                if (in) *dst = s;
            }
        }
    }
}

}  // namespace dnmf

extern "C" {

int dnmf_recon_image_lists(const float *At, const int *bbox, int K, int X, int Y, int Z, const float *C, long ldc,
                           const int *times, int B, float *S, long lds, dnmf_stream_t stream) {
    return dnmf_recon_image_lists_ex(At, bbox, K, X, Y, Z, C, ldc, times, B, S, lds, 0, stream);
}

int dnmf_recon_image_lists_ex(const float *At, const int *bbox, int K, int X, int Y, int Z, const float *C, long ldc,
                              const int *times, int B, float *S, long lds, int skip_empty, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(At && bbox && C && S, DNMF_E_NULL, "dnmf_recon_image_lists: NULL buffer");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && K > 0 && B > 0, DNMF_E_SHAPE, "dnmf_recon_image_lists: X=%d Y=%d Z=%d K=%d B=%d",
                 X, Y, Z, K, B);
    DNMF_REQUIRE(K <= 256, DNMF_E_UNSUPPORTED, "dnmf_recon_image_lists: K=%d > 256", K);
    const Volume vol = make_volume(X, Y, Z);
    const HaloLayout hl = make_halo_layout(X, Y, Z);
    DNMF_REQUIRE(lds >= hl.Pp, DNMF_E_SHAPE, "dnmf_recon_image_lists: lds=%ld < %ld floats of a halo-layout image", lds, hl.Pp);
    DNMF_REQUIRE((lds & 3) == 0 && ((size_t)S & 15) == 0 && ((size_t)At & 15) == 0, DNMF_E_SHAPE,
                 "dnmf_recon_image_lists: S, At must be 16-byte aligned and lds (%ld) a multiple of 4 floats", lds);
    const int nu = (hl.rowf + 63) / 64;
    const long ntile = (long)((hl.Xp + 3) / 4) * nu;
    DNMF_REQUIRE(ntile < (1L << 31), DNMF_E_UNSUPPORTED, "dnmf_recon_image_lists: %ld tiles", ntile);
    // frames per wave: enough waves to fill the chip (>= 16k), long enough runs to amortise the footprint loads
    int fpw = (int)(((long)B * ntile + 16383) / 16384);
    fpw = fpw < 16 ? 16 : fpw;
    fpw = fpw > B ? B : fpw;
    const int ny = (B + 4 * fpw - 1) / (4 * fpw);
    DNMF_REQUIRE(ny <= 65535, DNMF_E_UNSUPPORTED, "dnmf_recon_image_lists: B=%d too large", B);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)ntile, (unsigned)ny);
    if (K <= 64)
        hipLaunchKernelGGL(recon_lists_kernel<1>, grid, dim3(256), 0, st, At, bbox, K, vol, hl, C, ldc, times, B, S, lds, nu, fpw, skip_empty);
    else if (K <= 128)
        hipLaunchKernelGGL(recon_lists_kernel<2>, grid, dim3(256), 0, st, At, bbox, K, vol, hl, C, ldc, times, B, S, lds, nu, fpw, skip_empty);
    else
        hipLaunchKernelGGL(recon_lists_kernel<4>, grid, dim3(256), 0, st, At, bbox, K, vol, hl, C, ldc, times, B, S, lds, nu, fpw, skip_empty);
    return check_launch("dnmf_recon_image_lists");
}

}  // extern "C"
