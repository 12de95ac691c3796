// K3 -- fused warp + per-frame Gram matrix + right-hand side on the fp32 matrix cores.
//
// Reference: the two large contractions of DeformableNMF.update_temporal,
//   A_ts = einsum('mnzkt,mnzlt->klt', A_t, A_t)   (Demix/dNMF.py:141)
//   C1   = einsum('mnzkt,mnzt->kt',  A_t, Y)      (Demix/dNMF.py:142)
// on the warped footprints A_t that spatial_pushforward materialises for every frame in float64
// (Demix/dNMF.py:69-87; 839 GB at 512x512x4000, K=100).  Here A_t exists only as MFMA operands.
//
// Shape of the computation.  For one frame, M = [A_t | y] is a (P x Kp) matrix (Kp = 16*NB, column K
// carries the frame, the rest of the pad is zero) and the kernel needs the upper triangle of M^T M:
// NT = NB(NB+1)/2 tiles of 16x16, each accumulated with v_mfma_f32_16x16x4_f32 over the voxels, four
// voxels per instruction.  A wave owns ALL NT accumulator tiles (4*NT registers) for a contiguous chunk
// of voxels of one frame, so it needs no LDS exchange and no barrier with other waves.
//
// Operand layout.  For v_mfma_f32_16x16x4_f32 lane l supplies A[i=l&15][k=l>>4] and B[k=l>>4][j=l&15]:
// with k = voxel and i/j = channel the A and B fragments of a 16-channel block are the SAME register
// (channel l&15 of voxel l>>4), so one k-step (4 voxels) needs NB fragment registers per lane.  The
// channel <-> (block, lane) assignment is free (a Gram matrix is permutation-equivariant); it is chosen
// so that a lane's NB values are contiguous runs of its footprint row (16/8/4-byte loads):
//   blocks 4g..4g+3 -> channels 64g + 4i + (b-4g);  the R = NB%4 remaining blocks -> base + R*i + (b-b0).
// The frame rides in channel Kp-1 (= block NB-1, lane slot 15, always a pad column), so A_t^T y is the last
// row of the same product.  The permutation is undone when the tiles are scattered into G
// (gram_finish_kernel).
//
// Warp.  Coordinates, floors and corner weights are computed once per voxel by one lane (64 voxels per
// pass), parked in wave-private LDS and re-read as broadcasts by the 16 lanes that share the voxel.
// The fp32 sequence for the coordinates is common.hpp's (bit-compatible with the reference's
// normalise / grid_sample un-normalise round trip).
#include "../dnmf_amd/csrc/common.hpp"

namespace dnmf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int K3_SS = 64;  // voxels per coordinate pass (one per lane)

template <int NB>
__host__ __device__ constexpr int chan_of(int b, int i) {
    constexpr int NG4 = NB / 4, R = NB % 4, base = 64 * NG4;
    if (b < 4 * NG4) return 64 * (b / 4) + 4 * i + (b % 4);
    return base + R * i + (b - 4 * NG4);
}

__host__ __device__ constexpr int tile_index(int NB, int bi, int bj) { return bi * NB - bi * (bi - 1) / 2 + (bj - bi); }

// NB fragment values of one footprint row for lane slot i (channel permutation above): NB/4 16-byte loads
// at byte 256 g + 16 i plus one 4*(NB%4)-byte load at byte 256 (NB/4) + 4 (NB%4) i.
template <int NB>
struct RowFrag {
    float v[NB];
};

template <int NB>
__device__ __forceinline__ void load_row(RowFrag<NB> &f, const char *__restrict__ base, unsigned off_a, unsigned off_b) {
    constexpr int NG4 = NB / 4, R = NB % 4;
#pragma unroll
    for (int g = 0; g < NG4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(base + (size_t)(off_a + 256u * g));
        f.v[4 * g + 0] = t[0], f.v[4 * g + 1] = t[1], f.v[4 * g + 2] = t[2], f.v[4 * g + 3] = t[3];
    }
    if constexpr (R == 1) {
        f.v[4 * NG4] = *reinterpret_cast<const float *>(base + (size_t)off_b);
    } else if constexpr (R == 2) {
        const f32x2 t = *reinterpret_cast<const f32x2 *>(base + (size_t)off_b);
        f.v[4 * NG4] = t[0], f.v[4 * NG4 + 1] = t[1];
    } else if constexpr (R == 3) {
        const f32x3 t = *reinterpret_cast<const f32x3 *>(base + (size_t)off_b);
        f.v[4 * NG4] = t[0], f.v[4 * NG4 + 1] = t[1], f.v[4 * NG4 + 2] = t[2];
    }
}

struct GramParams {
    const float *Apk;
    long a_frame_stride;
    int Kp, K;
    Volume vol;
    const float *beta;
    int T;
    const int *times;
    int B;
    const float *frames;
    long ldf;
    const int *frame_ids;
    float *slab;       // (B, nchunks, NT, 64, 4)
    int nchunks;
    long chunk_len;    // passes (patches of 64 voxels) per chunk
    // a pass is a compact patch of 2^lgx x 2^lgy x 2^lgz = 64 voxels (8x8 for Z == 1): its gathers touch ~80
    // distinct footprint rows instead of the ~130 of a 64-voxel run, so more of them hit in L1
    int lgy, lgz, npy, npz;
    long npatch;
};

// Voxel record of the coordinate pass: byte offsets of the NTAP footprint rows, their weights (0 outside the volume,
// and for a voxel of a patch that sticks out of the volume) and the frame value.
template <int NTAP>
__device__ __forceinline__ void voxel_record(const float *bt, const Volume &vol, unsigned row_bytes,
                                             const float *__restrict__ yb, int x, int y, int z, unsigned (&rows)[NTAP],
                                             float (&w)[NTAP], float &yv) {
#pragma unroll
    for (int c = 0; c < NTAP; ++c) rows[c] = 0u, w[c] = 0.0f;
    yv = 0.0f;
    if (x < vol.X && y < vol.Y && z < vol.Z) {
        unsigned voxs[NTAP];
        if (bt) {
            const Sample sm = make_sample_t<(NTAP == 8)>(bt, vol, x, y, z);
            make_taps<NTAP>(sm, vol, w, voxs);
        } else {   // no warp (beta == NULL): the footprint row of the voxel itself, weight 1 -- exact on any lattice
            const unsigned self = (unsigned)((x * vol.Y + y) * vol.Z + z);
#pragma unroll
            for (int c = 0; c < NTAP; ++c) voxs[c] = self, w[c] = c == 0 ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int c = 0; c < NTAP; ++c) rows[c] = voxs[c] * row_bytes;
        yv = yb[((long)x * vol.Y + y) * vol.Z + z];
    }
}

// NTAP = 4 (Z == 1, bilinear) or 8 (trilinear).  Tap c: dx = c&1, dy = (c>>1)&1, dz = c>>2 (ATen's corner order).
//
// Schedule.  A k-step is {request the NTAP rows, blend them into NB fragments, NT MFMAs}.  The three
// stages of consecutive k-steps overlap inside one wave: while the MFMAs of step k are issued the rows of
// step k+2 are in flight and the fragments of step k+1 are blended in the issue slots the matrix pipe
// leaves free (an MFMA holds vector issue for 8 of its 32 cycles).  Two register sets (A/B) alternate, so
// the loop body is a pair of k-steps.  Voxel records are double-buffered in LDS one pass (64 voxels) ahead.
template <int NB, int NTAP>
__global__ __launch_bounds__(256, NTAP == 4 ? 2 : 1) void warp_gram_kernel(GramParams p) {
    constexpr int NT = NB * (NB + 1) / 2;
    constexpr int NQ = NTAP / 4;
    constexpr int NG4 = NB / 4, R = NB % 4;
    __shared__ u32x4 s_row[4][2][NQ][K3_SS];
    __shared__ f32x4 s_w[4][2][NQ][K3_SS];
    __shared__ float s_y[4][2][K3_SS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long item = (long)blockIdx.x * 4 + wave;  // chunk-major: neighbours share footprint rows in L2
    if (item >= (long)p.nchunks * p.B) return;      // whole wave leaves; no workgroup barrier below
    const int chunk = (int)(item / p.B);
    const int b = (int)(item - (long)chunk * p.B);
    const int t = p.times ? p.times[b] : b;
    const char *__restrict__ Ab = reinterpret_cast<const char *>(p.Apk + (long)b * p.a_frame_stride);
    const float *__restrict__ yb = p.frames + (long)(p.frame_ids ? p.frame_ids[b] : b) * p.ldf;
    const Volume vol = p.vol;
    const unsigned row_bytes = (unsigned)p.Kp * 4u;

    float bt[30];
    if (p.beta) load_beta(p.beta, p.T, t, bt);

    const int ci = lane & 15;   // channel slot
    const int vq = lane >> 4;   // voxel slot inside a k-step
    const unsigned lane_a = 16u * ci, lane_b = 256u * NG4 + 4u * R * ci;
    const bool ylane = ci == 15;  // (block NB-1, slot 15) = channel Kp-1 carries the frame

    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long q_begin = (long)chunk * p.chunk_len;
    const long q_end = q_begin + p.chunk_len < p.npatch ? q_begin + p.chunk_len : p.npatch;
    const int nss = (int)(q_end - q_begin);
    const int nk = nss * (K3_SS / 4);
    // patch origin of the next coordinate pass (wave-uniform, advanced with carries); lane -> voxel in the patch
    const int lgy = p.lgy, lgz = p.lgz, lgx = 6 - lgy - lgz;
    int pz = (int)(q_begin % p.npz), py = (int)((q_begin / p.npz) % p.npy), px = (int)(q_begin / ((long)p.npz * p.npy));
    const int lz = lane & ((1 << lgz) - 1), ly = (lane >> lgz) & ((1 << lgy) - 1), lx = lane >> (lgz + lgy);

    // coordinate pass s (called for s = 0, 1, 2, ... in order): lane -> one voxel of patch q_begin + s, record into
    // LDS buffer s & 1
    auto coord_pass = [&](int s) {
        const int x = (px << lgx) + lx, y = (py << lgy) + ly, z = (pz << lgz) + lz;
        if (++pz == p.npz) {
            pz = 0;
            if (++py == p.npy) py = 0, ++px;
        }
        unsigned rows[NTAP];
        float w[NTAP];
        float yv;
        voxel_record<NTAP>(p.beta ? bt : nullptr, vol, row_bytes, yb, x, y, z, rows, w, yv);
        const int buf = s & 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            s_row[wave][buf][q][lane] = u32x4{rows[4 * q], rows[4 * q + 1], rows[4 * q + 2], rows[4 * q + 3]};
            s_w[wave][buf][q][lane] = f32x4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
        }
        s_y[wave][buf][lane] = yv;
    };
    // LDS traffic of one wave is processed in order; only the compiler has to be kept from reordering
    auto wave_fence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    struct Stage {
        RowFrag<NB> raw[NTAP];
        f32x4 w[NQ];
        float y;
    };
    // voxel record of one k-step, fetched from LDS one step before the rows are requested
    struct Rec {
        u32x4 rr[NQ];
        f32x4 w[NQ];
        float y;
    };
    auto fetch_rec = [&](int kk, Rec &rc) {
        const int kc = kk < nk ? kk : nk - 1;  // past the end: re-read the last step (result unused)
        const int buf = (kc >> 4) & 1;
        const int vi = (kc & 15) * 4 + vq;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            rc.rr[q] = s_row[wave][buf][q][vi];
            rc.w[q] = s_w[wave][buf][q][vi];
        }
        rc.y = s_y[wave][buf][vi];
    };
    auto issue = [&](const Rec &rc, Stage &st) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            st.w[q] = rc.w[q];
#pragma unroll
#if ABL == 2
            load_row<NB>(st.raw[4 * q], Ab, rc.rr[q][0] + lane_a, rc.rr[q][0] + lane_b);
            for (int e = 1; e < 4; ++e) st.raw[4 * q + e] = st.raw[4 * q];
#else
            for (int e = 0; e < 4; ++e) load_row<NB>(st.raw[4 * q + e], Ab, rc.rr[q][e] + lane_a, rc.rr[q][e] + lane_b);
#endif
        }
        st.y = rc.y;
    };
    auto blend = [&](const Stage &st, float (&frag)[NB]) {
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) frag[bb] = st.raw[0].v[bb] * st.w[0][0];
#if ABL != 1
#pragma unroll
        for (int c = 1; c < NTAP; ++c) {
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) frag[bb] = fmaf(st.raw[c].v[bb], st.w[c >> 2][c & 3], frag[bb]);
        }
#endif
        frag[NB - 1] = ylane ? st.y : frag[NB - 1];
    };
    auto mfmas = [&](const float (&frag)[NB]) {
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
#pragma unroll
            for (int bj = bi; bj < NB; ++bj) {
                const int idx = tile_index(NB, bi, bj);
                acc[idx] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[bi], frag[bj], acc[idx], 0, 0, 0);
            }
        }
    };

    // one k-step: request the rows of step kk_req (record already in registers) and fetch the record of
    // step kk_req+1, then blend the fragments of the NEXT step (rows requested one step ago) in the shadow of
    // the MFMAs of the CURRENT step, one vector instruction per MFMA gap
    Rec rec;
    auto step = [&](int kk_req, Stage &st_req, const Stage &st_next, float (&f_next)[NB], const float (&f_cur)[NB]) {
        __builtin_amdgcn_sched_barrier(0);
        issue(rec, st_req);
        fetch_rec(kk_req + 1, rec);
        blend(st_next, f_next);
        mfmas(f_cur);
        // One stream: the row requests are spread over the MFMAs instead of being issued as a burst (a vector-memory
        // instruction occupies the wave for ~30 cycles when eight waves of a CU issue theirs at the same moment),
        // address adds and the blend fill the remaining vector slots.
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            if (i == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NQ + 1, 0);  // the next record (LDS reads)
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // address adds, then the blend
            if (i % 3 == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one row request
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    Stage sa, sb;
    float fa[NB], fb[NB];
    coord_pass(0);
    wave_fence();
    fetch_rec(0, rec);
    issue(rec, sa);
    fetch_rec(1, rec);
    issue(rec, sb);
    fetch_rec(2, rec);
    blend(sa, fa);
    for (int s = 0; s < nss; ++s) {
        if (s + 1 < nss) coord_pass(s + 1);
        wave_fence();
#pragma unroll 1
        for (int kk = s * (K3_SS / 4); kk < (s + 1) * (K3_SS / 4); kk += 2) {
            step(kk + 2, sa, sb, fb, fa);
            step(kk + 3, sb, sa, fa, fb);
        }
    }

    f32x4 *out = reinterpret_cast<f32x4 *>(p.slab) + (((long)b * p.nchunks + chunk) * NT) * 64 + lane;
#pragma unroll
    for (int i = 0; i < NT; ++i) out[(long)i * 64] = acc[i];
}

// Ordered sum of the chunk partials of one frame and scatter into G (K,K) / r (K) with the channel
// permutation undone.  D-tile element (row 4*(lane>>4)+reg, col lane&15) of tile (bi,bj).
template <int NB>
__global__ __launch_bounds__(256) void gram_finish_kernel(const float *__restrict__ slab, int nchunks, int K,
                                                          float *__restrict__ G, float *__restrict__ r) {
    constexpr int NT = NB * (NB + 1) / 2;
    const int b = blockIdx.x;
    const int e = threadIdx.x;  // element inside a tile: lane*4 + reg
    const int lane = e >> 2, reg = e & 3;
    const int i = 4 * (lane >> 4) + reg, j = lane & 15;
    float *Gb = G + (long)b * K * K;
    float *rb = r + (long)b * K;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
#pragma unroll
        for (int bj = bi; bj < NB; ++bj) {
            const int idx = tile_index(NB, bi, bj);
            const float *src = slab + (((long)b * nchunks) * NT + idx) * 256 + e;
            float s = 0.0f;
            for (int c = 0; c < nchunks; ++c) s += src[(long)c * NT * 256];
            const int k = chan_of<NB>(bi, i), l = chan_of<NB>(bj, j);
            constexpr int YC = 16 * NB - 1;  // channel that carries the frame
            if (k < K && l < K) {
                Gb[(long)k * K + l] = s;
                Gb[(long)l * K + k] = s;
            } else if (k < K && l == YC) {
                rb[k] = s;
            } else if (k == YC && l < K) {
                rb[l] = s;
            }
        }
    }
}

// K3b -- the same contraction with bf16 operands (v_mfma_f32_16x16x32_bf16, fp32 accumulate): the warped footprints
// are blended in fp32 exactly as in K3, rounded to bf16 (round to nearest even) and multiplied on the bf16 matrix
// pipe; the frame rides along as a bf16 column.  Not a reference precision -- BASELINE config 5 asks for it.
//
// Operand layout: lane l supplies A[i = l&15][k = 8(l>>4) + j], j = 0..7, and B[k][j = l&15] for the same k, so with
// k = voxel a lane blends EIGHT voxels (8(l>>4) .. +7 of a group of 32) of its channel slot per instruction, again
// one register set serving as both operands.  The accumulator tile is laid out like K3's, so the slab and
// gram_finish_kernel are shared.
//
// A wave alternates two phases per 32 voxels: gather + blend of its 8 voxels (rows of voxel j+1 in flight while
// voxel j is blended), then NT MFMAs.  Two waves per SIMD run the phases against each other; the gather phase is
// the longer one (same row traffic as K3 for an eighth of the matrix-pipe time).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const bf16x2 t = __builtin_convertvector(f32x2{lo, hi}, bf16x2);
    return __builtin_bit_cast(unsigned, t);
}

template <int NB, int NTAP>
__global__ __launch_bounds__(256, NTAP == 4 ? 2 : 1) void warp_gram_bf16_kernel(GramParams p) {
    constexpr int NT = NB * (NB + 1) / 2;
    constexpr int NQ = NTAP / 4;
    constexpr int NG4 = NB / 4, R = NB % 4;
    __shared__ u32x4 s_row[4][2][NQ][K3_SS];
    __shared__ f32x4 s_w[4][2][NQ][K3_SS];
    __shared__ float s_y[4][2][K3_SS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)p.nchunks * p.B) return;  // whole wave leaves; no workgroup barrier below
    const int chunk = (int)(item / p.B);
    const int b = (int)(item - (long)chunk * p.B);
    const int t = p.times ? p.times[b] : b;
    const char *__restrict__ Ab = reinterpret_cast<const char *>(p.Apk + (long)b * p.a_frame_stride);
    const float *__restrict__ yb = p.frames + (long)(p.frame_ids ? p.frame_ids[b] : b) * p.ldf;
    const Volume vol = p.vol;
    const unsigned row_bytes = (unsigned)p.Kp * 4u;

    float bt[30];
    if (p.beta) load_beta(p.beta, p.T, t, bt);

    const int ci = lane & 15;  // channel slot
    const int kq = lane >> 4;  // which eight voxels of a group of 32
    const unsigned lane_a = 16u * ci, lane_b = 256u * NG4 + 4u * R * ci;
    const bool ylane = ci == 15;

    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long q_begin = (long)chunk * p.chunk_len;
    const long q_end = q_begin + p.chunk_len < p.npatch ? q_begin + p.chunk_len : p.npatch;
    const int nss = (int)(q_end - q_begin);
    const int lgy = p.lgy, lgz = p.lgz, lgx = 6 - lgy - lgz;
    int pz = (int)(q_begin % p.npz), py = (int)((q_begin / p.npz) % p.npy), px = (int)(q_begin / ((long)p.npz * p.npy));
    const int lz = lane & ((1 << lgz) - 1), ly = (lane >> lgz) & ((1 << lgy) - 1), lx = lane >> (lgz + lgy);

    auto coord_pass = [&](int s) {
        const int x = (px << lgx) + lx, y = (py << lgy) + ly, z = (pz << lgz) + lz;
        if (++pz == p.npz) {
            pz = 0;
            if (++py == p.npy) py = 0, ++px;
        }
        unsigned rows[NTAP];
        float w[NTAP];
        float yv;
        voxel_record<NTAP>(p.beta ? bt : nullptr, vol, row_bytes, yb, x, y, z, rows, w, yv);
        const int buf = s & 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            s_row[wave][buf][q][lane] = u32x4{rows[4 * q], rows[4 * q + 1], rows[4 * q + 2], rows[4 * q + 3]};
            s_w[wave][buf][q][lane] = f32x4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
        }
        s_y[wave][buf][lane] = yv;
    };
    auto wave_fence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    struct Stage {
        RowFrag<NB> raw[NTAP];
        f32x4 w[NQ];
        float y;
    };
    // request the rows of voxel n (counted over the whole chunk, 64 per pass) of this lane's eight-voxel runs
    auto issue = [&](int n, Stage &st) {
        const int nc = n < nss * 16 ? n : nss * 16 - 1;  // past the end: re-read the last voxel (result unused)
        const int buf = (nc >> 4) & 1;
        const int vi = ((nc >> 3) & 1) * 32 + 8 * kq + (nc & 7);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const u32x4 rr = s_row[wave][buf][q][vi];
            st.w[q] = s_w[wave][buf][q][vi];
#pragma unroll
            for (int e = 0; e < 4; ++e) load_row<NB>(st.raw[4 * q + e], Ab, rr[e] + lane_a, rr[e] + lane_b);
        }
        st.y = s_y[wave][buf][vi];
    };
    auto blend = [&](const Stage &st, float (&val)[NB]) {
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) val[bb] = st.raw[0].v[bb] * st.w[0][0];
#pragma unroll
        for (int c = 1; c < NTAP; ++c) {
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) val[bb] = fmaf(st.raw[c].v[bb], st.w[c >> 2][c & 3], val[bb]);
        }
        val[NB - 1] = ylane ? st.y : val[NB - 1];
        // opaque to the vectoriser: otherwise the blends of two voxels are fused into packed-pair arithmetic (they
        // meet in one v_cvt_pk_bf16_f32), which ties the first voxel's blend to the arrival of the second's rows
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) asm("" : "+v"(val[bb]));
    };

    Stage sa, sb;
    unsigned frag[NB][4];
    float va[NB], vb[NB];
    coord_pass(0);
    wave_fence();
    issue(0, sa);
    for (int s = 0; s < nss; ++s) {
        if (s + 1 < nss) coord_pass(s + 1);
        wave_fence();  // pass s+1 is in LDS before the look-ahead of the last voxel of pass s reads it
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const int n0 = s * 16 + h * 8;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                issue(n0 + j + 1, sb);
                __builtin_amdgcn_sched_barrier(0);  // keep the two row sets from being requested at once
                blend(sa, va);
                __builtin_amdgcn_sched_barrier(0);
                issue(n0 + j + 2, sa);
                __builtin_amdgcn_sched_barrier(0);
                blend(sb, vb);
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) frag[bb][j >> 1] = pack_bf16(va[bb], vb[bb]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int bi = 0; bi < NB; ++bi) {
                const bf16x8 fi = __builtin_bit_cast(bf16x8, u32x4{frag[bi][0], frag[bi][1], frag[bi][2], frag[bi][3]});
#pragma unroll
                for (int bj = bi; bj < NB; ++bj) {
                    const bf16x8 fj = __builtin_bit_cast(bf16x8, u32x4{frag[bj][0], frag[bj][1], frag[bj][2], frag[bj][3]});
                    const int idx = tile_index(NB, bi, bj);
                    acc[idx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fi, fj, acc[idx], 0, 0, 0);
                }
            }
        }
    }

    f32x4 *out = reinterpret_cast<f32x4 *>(p.slab) + (((long)b * p.nchunks + chunk) * NT) * 64 + lane;
#pragma unroll
    for (int i = 0; i < NT; ++i) out[(long)i * 64] = acc[i];
}

static void patch_shape(const Volume &vol, int &lgy, int &lgz, int &npy, int &npz, long &npatch) {
    lgz = vol.Z == 1 ? 0 : (vol.Z == 2 ? 1 : 2);
    lgy = vol.Z <= 2 ? 3 : 2;
    const int lgx = 6 - lgy - lgz;
    npz = (vol.Z + (1 << lgz) - 1) >> lgz;
    npy = (vol.Y + (1 << lgy) - 1) >> lgy;
    npatch = (long)((vol.X + (1 << lgx) - 1) >> lgx) * npy * npz;
}

static void choose_chunks(long npatch, int B, int &nchunks, long &chunk_len) {
    const long nss = npatch;  // coordinate passes per frame
    long want = (4096 + B - 1) / B;            // aim for >= 4096 wave-sized work items
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    if (want > nss) want = nss;
    const long ss_per_chunk = (nss + want - 1) / want;
    chunk_len = ss_per_chunk;
    nchunks = (int)((npatch + chunk_len - 1) / chunk_len);
}

template <int NB>
static int launch_gram(GramParams p, float *G, float *r, bool bf16, hipStream_t st) {
    const long nitems = (long)p.nchunks * p.B;
    const unsigned nwg = (unsigned)((nitems + 3) / 4);
    if (bf16 && p.vol.Z > 1)
        hipLaunchKernelGGL((warp_gram_bf16_kernel<NB, 8>), dim3(nwg), dim3(256), 0, st, p);
    else if (bf16)
        hipLaunchKernelGGL((warp_gram_bf16_kernel<NB, 4>), dim3(nwg), dim3(256), 0, st, p);
    else if (p.vol.Z > 1)
        hipLaunchKernelGGL((warp_gram_kernel<NB, 8>), dim3(nwg), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((warp_gram_kernel<NB, 4>), dim3(nwg), dim3(256), 0, st, p);
    hipLaunchKernelGGL((gram_finish_kernel<NB>), dim3((unsigned)p.B), dim3(256), 0, st, p.slab, p.nchunks, p.K, G, r);
    return check_launch("dnmf_warp_gram_rhs");
}

}  // namespace dnmf

extern "C" {

size_t dnmf_warp_gram_rhs_workspace(long P, int K, int B) {
    if (P <= 0 || K <= 0 || B <= 0) return 0;
    const int NB = dnmf_padded_k(K) / 16;
    long want = (4096 + B - 1) / B;  // upper bound of the chunk count chosen at launch (it depends on the shape)
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    return (size_t)B * (size_t)(want + 1) * (NB * (NB + 1) / 2) * 256 * sizeof(float);
}

static int gram_entry(bool bf16, const float *Apk, int Kp, int K, long a_frame_stride, int X, int Y, int Z,
                      const float *beta, int T, const int *times, int B, const float *frames, long ldf,
                      const int *frame_ids, float *G, float *r, void *workspace, size_t workspace_bytes,
                      dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(Apk && frames && G && r && workspace, DNMF_E_NULL, "dnmf_warp_gram_rhs: NULL buffer");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && K > 0 && T > 0 && B > 0 && Kp == dnmf_padded_k(K), DNMF_E_SHAPE,
                 "dnmf_warp_gram_rhs: X=%d Y=%d Z=%d K=%d Kp=%d T=%d B=%d", X, Y, Z, K, Kp, T, B);
    GramParams p;
    p.vol = make_volume(X, Y, Z);
    DNMF_REQUIRE(ldf >= p.vol.P && a_frame_stride >= 0, DNMF_E_SHAPE, "dnmf_warp_gram_rhs: ldf=%ld < P=%ld", ldf,
                 p.vol.P);
    DNMF_REQUIRE(p.vol.P * Kp < (1L << 30), DNMF_E_UNSUPPORTED,
                 "dnmf_warp_gram_rhs: P*Kp=%ld does not fit 32-bit byte offsets", p.vol.P * Kp);
    DNMF_REQUIRE((reinterpret_cast<size_t>(Apk) & 15) == 0 && (a_frame_stride % 4) == 0 &&
                     (reinterpret_cast<size_t>(workspace) & 15) == 0,
                 DNMF_E_SHAPE, "dnmf_warp_gram_rhs: Apk / workspace must be 16-byte aligned");
    DNMF_REQUIRE(workspace_bytes >= dnmf_warp_gram_rhs_workspace(p.vol.P, K, B), DNMF_E_WORKSPACE,
                 "dnmf_warp_gram_rhs: workspace %zu < %zu bytes", workspace_bytes,
                 dnmf_warp_gram_rhs_workspace(p.vol.P, K, B));
    p.Apk = Apk, p.a_frame_stride = a_frame_stride, p.Kp = Kp, p.K = K;
    p.beta = beta, p.T = T, p.times = times, p.B = B;
    p.frames = frames, p.ldf = ldf, p.frame_ids = frame_ids;
    p.slab = static_cast<float *>(workspace);
    patch_shape(p.vol, p.lgy, p.lgz, p.npy, p.npz, p.npatch);
    choose_chunks(p.npatch, B, p.nchunks, p.chunk_len);
    hipStream_t st = (hipStream_t)stream;
    switch (Kp / 16) {
        case 1: return launch_gram<1>(p, G, r, bf16, st);
        case 2: return launch_gram<2>(p, G, r, bf16, st);
        case 3: return launch_gram<3>(p, G, r, bf16, st);
        case 4: return launch_gram<4>(p, G, r, bf16, st);
        case 5: return launch_gram<5>(p, G, r, bf16, st);
        case 6: return launch_gram<6>(p, G, r, bf16, st);
        case 7: return launch_gram<7>(p, G, r, bf16, st);
        case 8: return launch_gram<8>(p, G, r, bf16, st);
        default:
            return fail(DNMF_E_UNSUPPORTED, "dnmf_warp_gram_rhs: K=%d needs Kp=%d > 128 (not built yet)", K, Kp);
    }
}

int dnmf_warp_gram_rhs(const float *Apk, int Kp, int K, long a_frame_stride, int X, int Y, int Z, const float *beta,
                       int T, const int *times, int B, const float *frames, long ldf, const int *frame_ids, float *G,
                       float *r, void *workspace, size_t workspace_bytes, dnmf_stream_t stream) {
    return gram_entry(false, Apk, Kp, K, a_frame_stride, X, Y, Z, beta, T, times, B, frames, ldf, frame_ids, G, r,
                      workspace, workspace_bytes, stream);
}

int dnmf_warp_gram_rhs_bf16(const float *Apk, int Kp, int K, long a_frame_stride, int X, int Y, int Z,
                            const float *beta, int T, const int *times, int B, const float *frames, long ldf,
                            const int *frame_ids, float *G, float *r, void *workspace, size_t workspace_bytes,
                            dnmf_stream_t stream) {
    return gram_entry(true, Apk, Kp, K, a_frame_stride, X, Y, Z, beta, T, times, B, frames, ldf, frame_ids, G, r,
                      workspace, workspace_bytes, stream);
}

}  // extern "C"
