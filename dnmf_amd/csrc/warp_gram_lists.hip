// K3n -- fused warp + per-frame Gram matrix + right-hand side for COMPACT footprints, neuron by neuron.
//
// Reference: the two contractions of DeformableNMF.update_temporal (Demix/dNMF.py:141-142) on the warped
// footprints of spatial_pushforward (dNMF.py:69-87), as K3 / K3s.
//
// Why another kernel.  The reference's Gaussians underflow to an exact zero ~30 px from their centre, so at
// 512x512, K=100 a voxel sees 1.1 non-zero footprints on average and A_t^T A_t has ~2.6 GFLOP of non-zero products
// per 4000 frames, not 11.6 TFLOP.  K3s skips whole 16-neuron blocks but still evaluates 16x16 products per active
// block pair on the matrix pipe and spends most of its time on block bookkeeping.  Here nothing is padded to
// blocks: a wave walks over tiles of 256 voxels, looks up which neurons can reach the tile at all and evaluates
// only those, on the vector ALU -- no MFMA: there is no dense operand left to feed it.
//
// Exactness.  A product is skipped only when one factor is an exact zero, so every sum equals the dense kernel's
// up to the order of fp32 additions:
//   - bbox[k] is the bounding box of the non-zeros of footprint k (dnmf_pack_footprints_lists);
//   - a tile's neuron list (lists_tilemask_kernel, one bit per neuron) is every k whose bbox meets a box that
//     CONTAINS all taps of the tile's voxels: the range of the warped coordinate over the tile by interval
//     arithmetic on the ten monomials, widened by a margin far above the fp32 rounding of either evaluation.  A
//     neuron listed without need contributes exact zeros (its footprint is zero outside its bbox);
//   - G[k,l] can be non-zero for some warp only if a 2x2(x2) tap cell meets both bboxes, i.e. if per axis
//     k_lo - 1 <= l_hi and l_lo - 1 <= k_hi: a static pattern, independent of the warp (both footprints move
//     under the same map).  Pairs outside the pattern are never accumulated and come out as 0.
//
// Data flow.  Lane = voxel: 32 lanes along y for Z == 1 (a wave reads whole 128-byte lines of the frame; 16 for 3-D
// volumes), four voxels per lane along x; tiles (8 x 32 x 1, 8 x 16 x 2 or 4 x 16 x 4 voxels) are walked along x, so a
// lane's (y,z) and the monomials without x change only at the end of a tile row.  At is in the halo layout of common.hpp:
// a tap outside the volume reads a zero, no masks or clamps along x and y.  Per listed neuron the warped values a_k of
// the lane's voxels are NTAP FMAs on NTAP gathered taps.  Gathered straight from global memory those are 2 x NTAP/2
// eight-byte loads per voxel with addresses that differ from lane to lane: ~16 cycles of a CU's texture path per
// wave-instruction whatever its width (tools/gather_probe.hip), eight per neuron and tile, which kept that path 96 %
// busy and set the kernel's time.  For Z == 1 the taps of a tile lie in a region of 12 rows x 40 floats of the footprint
// image (lists_tilemask_kernel: the tile's tap box, first column aligned to 16 bytes): the wave copies that region of
// each listed neuron into LDS with two sixteen-byte loads per lane and gathers from LDS (2.5 cycles per instruction);
// tiles whose taps do not fit (a warp that scales a tile by more than ~10 %) and 3-D volumes keep the direct gathers.
// Then r_k += a_k.y and G_kl += a_k.a_l for the listed
// l >= k as per-lane partial sums.  While consecutive tiles have the same list the partial sums stay in registers;
// when the list changes: a fixed DPP tree over the 64 lanes and one LDS add by the last lane into the wave's private
// table of pattern slots.  LDS operations of one wave retire in order, so the sum is deterministic.  The table
// goes to a slab per (frame, chunk); the finish kernel adds the chunks in order and scatters the slots into dense
// G (K,K), r (K).
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "common.hpp"

namespace dnmf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));   // four floats at an 8-byte aligned address
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

#ifndef DNMF_K3N_NT
#define DNMF_K3N_NT 0   // frame values by non-temporal loads
#endif
// Coordinates of a voxel's taps.  0 (the product build): the reference's fp32 sequence -- the ten-term FMA chain,
// n = 2q/(S-1) - 1, u = ((n+1)/2)(S-1) -- which is what decides floor() at lattice coincidences and what K2, K1 and the
// dense kernels evaluate.  1 (a measured option, -DDNMF_K3N_DIRECT=1): u_d = q_d(x, y, z) evaluated directly, as a
// quadratic in x with coefficients the lane keeps per tile row: two FMAs per axis and voxel instead of ~20 instructions,
// 109 registers instead of 128, 3.24 ms against 3.45 per 4000 frames at 512x512, K=100.  The interpolated footprint
// value is a CONTINUOUS function of u, so G and r then differ from the faithful evaluation only by what ~1e-5 px of
// coordinate noise make (the reference's own round trip through n loses that much near the middle of an axis): up to
// 7e-6 of the largest entry on the sharp test footprints, inside the stated 2e-5 but three times what separates the
// faithful K3n from the dense kernel -- not worth 3 % of a sweep.
#ifndef DNMF_K3N_DIRECT
#define DNMF_K3N_DIRECT 0
#endif
#ifndef DNMF_K3N_EARLY1
#define DNMF_K3N_EARLY1 1
#endif
#ifndef DNMF_K3N_WAVES
#define DNMF_K3N_WAVES 4   // waves per SIMD the Z == 1 kernel is compiled for (128 registers)
#endif
#ifndef DNMF_K3N_DMA
#define DNMF_K3N_DMA 1   // 1: the regions of a list go from global memory straight into LDS (global_load_lds), all of them
#endif                   // requested ahead of the tile's coordinate arithmetic, through no registers (round 3: 3.14 -> 2.91 ms)
#ifndef DNMF_K3N_LATE_FRAMES
#define DNMF_K3N_LATE_FRAMES 1   // the wait in front of the staged taps covers the regions only, not the frame values
#endif
#ifndef DNMF_K3N_DMA2
#define DNMF_K3N_DMA2 1  // the same for the two groups of a long list (second launch)
#endif
#ifndef DNMF_K3N_WAVES_Z2
#define DNMF_K3N_WAVES_Z2 4   // the same for Z == 2
#endif
#ifndef DNMF_K3N_WAVES_Z2L
#define DNMF_K3N_WAVES_Z2L 3  // its long-list pass (two groups of four neurons' values in registers)
#endif
#ifndef DNMF_K3N_WAVES_Z3
#define DNMF_K3N_WAVES_Z3 3   // and for Z > 2 (direct gathers only)
#endif

constexpr int LISTS_NG = 4;      // neurons evaluated together (register slots); longer lists are cut into groups
constexpr int LISTS_LGV = 2;      // log2 of the voxels per lane (consecutive x positions, interleaved by lane)
constexpr int LISTS_VPL = 1 << LISTS_LGV;
constexpr int LISTS_MAXW = 4;    // 64-neuron words of a tile's list: K <= 256
constexpr long LISTS_ITEMS = 16384;  // target number of wave-sized work items per launch
constexpr int LISTS_MAX_SLOTS = 3800;  // 4 waves x 3800 words of LDS per workgroup
// Staged gathers (Z == 1): the taps of an 8 x 32 tile lie in a region of RR rows x RC floats of a footprint image; a wave
// copies that region of each listed neuron into LDS with sixteen-byte loads and gathers from there (see the kernel).
#ifndef DNMF_K3N_RR
#define DNMF_K3N_RR 12
#endif
constexpr int LISTS_RR = DNMF_K3N_RR, LISTS_RC = 40;                 // 9 tap rows + slack; 33 tap columns + alignment of the first + slack
constexpr int LISTS_REGION = LISTS_RR * LISTS_RC;            // floats per neuron: 1,920 bytes, 120 sixteen-byte pieces
static_assert(LISTS_RC % 4 == 0 && LISTS_REGION / 4 <= 128 && LISTS_REGION / 4 > 64, "stage_load moves two pieces per lane");

struct ListParams {
    const float *At;       // (K, halo layout)
    const int *bbox;       // (K,6) xlo,xhi,ylo,yhi,zlo,zhi; lo > hi for an all-zero footprint
    const int *pair_slot;  // (K,K) symmetric; pairs outside the pattern point at the trash slot nslot-1
    const unsigned long long *axis_masks;  // per axis and bound the neurons whose box starts / ends there (see below)
    int nslot;             // K rhs slots, then the pattern pairs, then one trash slot
    int K;
    Volume vol;
    HaloLayout hl;
    const float *beta;
    int T;
    const int *times;
    int B;
    const float *frames;
    long ldf;
    const int *frame_ids;
    float *slab;  // (B, tables, nslot): table c of a frame from chunk c; with two launches table nchunks + c from the second
    unsigned long long *tile_masks;  // (B, ntiles, NW): the neuron list of every tile of every frame
    int4 *tile_desc;       // (B, ntiles): x = the tile's tap region packed as first halo row << 16 | first float of a halo row,
                           // -1: the taps do not fit LISTS_RR x LISTS_RC (or Z > 1): direct gathers; y = the list's first
                           // four neurons, one per byte in ascending order (0xff past the end); z = its length
    int nchunks, chunk_len;
    int tables;            // nchunks, or 2 nchunks when the long-list tiles go in a launch of their own
    int lgx, lgy, lgz;  // tile = (LISTS_VPL << lgx) x (1 << lgy) x (1 << lgz) voxels, lgx + lgy + lgz = 6: 8 x 32 x 1 for
                        // Z == 1 (a wave reads whole 128-byte lines of a frame: with 16 voxels along y every line was
                        // fetched twice, by tiles 32 apart in the walk), 8 x 16 x 2, 4 x 16 x 4
    int ntx, nty, ntz, ntiles;     // tile q = (qy * ntz + qz) * ntx + qx: walked along x
    unsigned long long *counters;  // optional: [0] += (tile, neuron) evaluations, [1] += (tile, pair) sums
};

// ---- neuron lists of the tiles ---------------------------------------------------------------------------
// axis_masks: for axis d (0,1,2) two tables of S_d + 2 entries of NW 64-bit words:
//   LO_d[i + 1] = { k : bbox_lo_d[k] <= i },  i = -1 .. S_d   (entry 0 is the empty set)
//   HI_d[i]     = { k : bbox_hi_d[k] >= i },  i =  0 .. S_d+1 (the last two are empty)
// so the neurons whose box meets [a, b] along d are LO_d[min(b, S_d) + 1] & HI_d[max(a, 0)]: the list of a tile is the
// AND of three such pairs.  Built by lists_axis_masks_kernel from bbox.
__host__ __device__ inline long axis_masks_offset(const Volume &vol, int d, int hi_table) {
    const int S[3] = {vol.X, vol.Y, vol.Z};
    long o = 0;
    for (int e = 0; e < d; ++e) o += 2L * (S[e] + 2);
    return o + (hi_table ? S[d] + 2 : 0);
}
__host__ __device__ inline long axis_masks_entries(const Volume &vol) { return 2L * (vol.X + vol.Y + vol.Z + 6); }

#ifndef DNMF_K3N_TU_Z   // (the translation unit of the Z >= 2 instantiations takes only the templates)
__global__ __launch_bounds__(256) void lists_axis_masks_kernel(const int *__restrict__ bbox, int K, Volume vol, int NW,
                                                               unsigned long long *__restrict__ masks) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= axis_masks_entries(vol)) return;
    const int S[3] = {vol.X, vol.Y, vol.Z};
    long o = e;
    int d = 0;
    while (o >= 2L * (S[d] + 2)) o -= 2L * (S[d] + 2), ++d;
    const bool hi_table = o >= S[d] + 2;
    const int i = hi_table ? (int)(o - (S[d] + 2)) : (int)o - 1;
    for (int w = 0; w < NW; ++w) {
        unsigned long long m = 0;
        for (int j = 0; j < 64; ++j) {
            const int k = 64 * w + j;
            if (k >= K) break;
            const int lo = bbox[k * 6 + 2 * d], hi = bbox[k * 6 + 2 * d + 1];
            const bool in = lo <= hi && (hi_table ? hi >= i : lo <= i);
            m |= (unsigned long long)in << j;
        }
        masks[e * NW + w] = m;
    }
}

#endif  // DNMF_K3N_TU_Z

// One thread per (frame, tile): the tile's neuron list as NW 64-bit words.
template <int NW>
__global__ __launch_bounds__(256) void lists_tilemask_kernel(ListParams p) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)p.B * p.ntiles) return;
    const int b = (int)(id / p.ntiles), q = (int)(id - (long)b * p.ntiles);
    const Volume vol = p.vol;
    const int t = p.times ? p.times[b] : b;
    float bt[30];
    load_beta(p.beta, p.T, t, bt);
    const int qx = q % p.ntx, rest = q / p.ntx, qz = rest % p.ntz, qy = rest / p.ntz;
    const int tx = LISTS_VPL << p.lgx, ty = 1 << p.lgy, tz = 1 << p.lgz;
    const int S[3] = {vol.X, vol.Y, vol.Z};
    const int first[3] = {qx * tx, qy * ty, qz * tz}, size[3] = {tx, ty, tz};
    float lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) lo[d] = (float)first[d], hi[d] = (float)min(first[d] + size[d] - 1, S[d] - 1);
    const bool hasz = vol.Z > 1;
    unsigned long long m[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) m[w] = ~0ull;
    int ta[3] = {0, 0, 0}, tc[3] = {0, 0, 0};
    bool finite = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int a = 0, c = 0;  // tap range along d; Z == 1: the taps sit in slice 0
        if ((d < 2 || hasz) && !tap_range(bt, vol, d, lo, hi, hasz, a, c)) {
            // NaN coordinates: every tap of the tile is pulled into the halo and reads zeros
#pragma unroll
            for (int w = 0; w < NW; ++w) m[w] = 0;
            finite = false;
            continue;
        }
        ta[d] = a, tc[d] = c;
        const unsigned long long *LO = p.axis_masks + (axis_masks_offset(vol, d, 0) + (min(c, S[d]) + 1 < 0 ? 0 : min(c, S[d]) + 1)) * NW;
        const unsigned long long *HI = p.axis_masks + (axis_masks_offset(vol, d, 1) + min(max(a, 0), S[d] + 1)) * NW;
#pragma unroll
        for (int w = 0; w < NW; ++w) m[w] &= LO[w] & HI[w];
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) p.tile_masks[id * NW + w] = m[w];
    // the region of a footprint image that holds every tap of the tile (the gathers clamp their base corner into
    // [-HALO, S], the second corner is one further), origin pulled back so that the region stays inside the image
    // (Z == 2: a halo row interleaves the two slices, column y of the volume is floats 2 (y + HALO), 2 (y + HALO) + 1, and
    // the region holds both slices of its columns, so the z range of the taps does not matter)
    int2 reg = make_int2(-1, 0);
    if (vol.Z <= 2 && finite && p.hl.Xp >= LISTS_RR && p.hl.rowf >= LISTS_RC && p.hl.Xp < 32768 && p.hl.rowf < 65536) {
        const int ax = max(ta[0], -HALO), cx = min(tc[0], vol.X + 1), ay = max(ta[1], -HALO), cy = min(tc[1], vol.Y + 1);
        const int r0 = min(ax + HALO, p.hl.Xp - LISTS_RR), c0 = min(((ay + HALO) * vol.Z) & ~3, p.hl.rowf - LISTS_RC);
        if (cx >= ax && cy >= ay && cx + HALO - r0 < LISTS_RR && (cy + HALO) * vol.Z + vol.Z - 1 - c0 < LISTS_RC)
            reg = make_int2(r0, c0);
    }
    int nlist = 0;
    unsigned ids = 0xffffffffu;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        unsigned long long rem = m[w];
        while (rem) {
            if (nlist < 4) ids = (ids & ~(0xffu << (8 * nlist))) | ((unsigned)(64 * w + __builtin_ctzll(rem)) << (8 * nlist));
            ++nlist, rem &= rem - 1;
        }
    }
    // w: the tile's place, qx | (qy * ntz + qz) << 16 when both fit 16 bits (else -1: the kernel divides)
    const int place = (qx < 65536 && rest < 32768) ? (qx | rest << 16) : -1;
    p.tile_desc[id] = make_int4(reg.x < 0 ? -1 : (reg.x << 16 | reg.y), (int)ids, nlist, place);
}

// PASS 1: the tiles with at most LISTS_NG neurons; PASS 2: the other tiles, into tables of their own (the consumers sum
// a frame's tables in order), launched on a side stream so that the two run side by side: pass 2 alone is a chain of
// memory round trips per tile with nothing to hide them behind.  Two kernels because the compiler allocates registers for the union
// of all paths of one: with the staged long-list code inside, the short-list loop spilled.  PASS 0 is the one-kernel
// form (every tile, long lists by direct gathers) for launches whose waves have only a short run of tiles each, where the
// second launch costs more than it saves; the host picks (lists_passes).
//
// ZM = 1: Z == 1 (four taps).  ZM = 2: Z == 2 -- the two z-taps of a corner are the adjacent floats of a halo row, fetched
// as the pair (slice 0, slice 1) whose members take the weight of the tap they stand for (warp_recon_grad.hip has the
// derivation), from the staged region like Z == 1 (a region is 12 rows x 20 columns x 2 slices) or, for tiles without a
// region, as two sixteen-byte gathers per voxel and neuron.  ZM = 3: Z > 2, four eight-byte gathers (z-pair of a corner).
// For Z >= 2 the blend is hierarchical (z, then y, then x: s0 + w1 (s1 - s0) along x and y, whose weights add up to 1
// exactly) from four weights per voxel; eight pre-multiplied weights per voxel cost 16 more registers.
template <int ZM, int NW, int FAST, bool F32OFF, int PASS>
__global__ __launch_bounds__(256, (ZM == 1 ? DNMF_K3N_WAVES : (ZM == 2 ? (PASS == 2 ? DNMF_K3N_WAVES_Z2L : DNMF_K3N_WAVES_Z2) : DNMF_K3N_WAVES_Z3))) void warp_gram_lists_kernel(ListParams p) {
    constexpr bool LONGPASS = PASS == 2;
    extern __shared__ float s_tab[];
    constexpr bool HASZ = ZM > 1;
    constexpr int NPAIR = LISTS_NG * (LISTS_NG + 1) / 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long item = (long)blockIdx.x * 4 + wave;  // chunk-major: neighbouring waves work on the same part of At
    if (item >= (long)p.nchunks * p.B) return;      // whole wave leaves; no workgroup barrier below
    const int chunk = (int)(item / p.B);
    const int b = (int)(item - (long)chunk * p.B);
    const int t = p.times ? p.times[b] : b;
    const float *__restrict__ yb = p.frames + (long)(p.frame_ids ? p.frame_ids[b] : b) * p.ldf;
    const Volume vol = p.vol;
    const HaloLayout hl = p.hl;
    const int K = p.K;
    const size_t plane = (size_t)hl.Pp * 4u;  // bytes of one neuron's footprint image

    float bt[30];
    load_beta(p.beta, p.T, t, bt);

    float *tab = s_tab + (size_t)wave * p.nslot;
    for (int i = lane; i < p.nslot; i += 64) tab[i] = 0.0f;
    const unsigned tab_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float *)tab;  // LDS byte address
    // this wave's LISTS_NG staging regions (NTAP == 4 only; 16-byte aligned: the tables before them are padded)
    char *stage_lds = reinterpret_cast<char *>(s_tab + (((size_t)4 * p.nslot + 3) & ~(size_t)3)) +
                      (size_t)wave * (LISTS_NG * LISTS_REGION * 4);
    // piece e = lane + 64 j (j = 0, 1) of a region is row e / (RC/4), floats 4 (e % (RC/4)) .. +3
    constexpr int PPR = LISTS_RC / 4;
    const int piece_row[2] = {lane / PPR, (lane + 64) / PPR};
    const int piece_c4[2] = {lane - piece_row[0] * PPR, lane + 64 - piece_row[1] * PPR};
    const int4 *__restrict__ descs = p.tile_desc + (long)b * p.ntiles;

    const int lgx = p.lgx, lgz = p.lgz;
    const int lgy = p.lgy;
    const int lz = lane & ((1 << lgz) - 1), ly = (lane >> lgz) & ((1 << lgy) - 1), lx = lane >> (lgz + lgy);
    const int q_begin = chunk * p.chunk_len;
    const int q_end = min(q_begin + p.chunk_len, p.ntiles);
    const unsigned long long *__restrict__ masks = p.tile_masks + (long)b * p.ntiles * NW;
    unsigned long long n_eval = 0, n_pair = 0;  // wave-uniform
#ifdef DNMF_K3N_STAMPS
    // diagnostic build only (tools/k3n_stamps.py): wave cycles per section of the tile loop into counters[2..8], then
    // the number of non-empty tiles, of long-list tiles and of flushed runs
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define DNMF_STAMP(i)                                                     \
    {                                                                     \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
        st_acc[i] += now_ - st_last, st_last = now_;                      \
    }
#else
#define DNMF_STAMP(i)
#endif

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // one lane, one LDS add (an atomic builtin here is rewritten into a cross-lane reduction loop)
    auto add_slot = [&](int slot, float total_in_last_lane) {
        const unsigned addr = tab_lds + 4u * (unsigned)slot;
        if (lane == 63) asm volatile("ds_add_f32 %0, %1" : : "v"(addr), "v"(total_in_last_lane) : "memory");
    };
    // slot of a pair: a scalar load issued before the arithmetic that precedes its use
    auto pair_slot_of = [&](int k, int l) {
#ifdef DNMF_K3N_ABL_SLOT
        return K + ((max(k, 0) * 7 + max(l, 0)) & 255);   // timing ablation (tools/k3n_stamps.py): wrong sums, no table lookup
#else
        return __builtin_amdgcn_readfirstlane(p.pair_slot[max(k, 0) * K + max(l, 0)]);
#endif
    };
    // the lowest LISTS_NG set bits of a list (-1 past its end); `rem` loses them
    auto take_ids = [&](unsigned long long (&rem)[NW], int (&ks)[LISTS_NG]) {
#pragma unroll
        for (int i = 0; i < LISTS_NG; ++i) {
            int k = -1;
            bool found = false;
#pragma unroll
            for (int wd = 0; wd < NW; ++wd) {
                const bool take = !found && rem[wd] != 0;
                k = take ? 64 * wd + __builtin_ctzll(rem[wd]) : k;
                rem[wd] = take ? rem[wd] & (rem[wd] - 1) : rem[wd];
                found = found || take;
            }
            ks[i] = k;
        }
    };

    // partial sums of the current run of tiles with one and the same list of at most LISTS_NG neurons
    float acc_r[LISTS_NG], acc_p[NPAIR];
    int run_k[LISTS_NG];
    int run_n = 0;  // 0: nothing pending
    auto flush = [&]() {
        if (run_n == 0) return;
        auto go = [&](auto nn) {
            constexpr int N = decltype(nn)::value;
            int sl[N][N];
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = i; j < N; ++j) sl[i][j] = pair_slot_of(run_k[i], run_k[j]);
            float sr[N], sp[N][N];
            int e = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                sr[i] = wave_sum_last(acc_r[i]);
#pragma unroll
                for (int j = i; j < N; ++j) sp[i][j] = wave_sum_last(acc_p[e++]);
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                add_slot(run_k[i], sr[i]);
#pragma unroll
                for (int j = i; j < N; ++j) add_slot(sl[i][j], sp[i][j]);
            }
        };
        using std::integral_constant;
        switch (run_n) {
            case 1: go(integral_constant<int, 1>{}); break;
            case 2: go(integral_constant<int, 2>{}); break;
            case 3: go(integral_constant<int, 3>{}); break;
            default: go(integral_constant<int, 4>{}); break;
        }
        static_assert(LISTS_NG == 4, "the dispatch above lists the group sizes");
        run_n = 0;
    };

    float b2[30];
    double_beta(bt, b2);
    Monomials<HASZ> mono = monomials<HASZ>(0.0f, 0.0f, 0.0f);
    int row_of_c = -1;  // the tile row (qy, qz) `mono` belongs to
#if DNMF_K3N_DIRECT
    float hc0[3] = {0.f, 0.f, 0.f}, hc1[3] = {0.f, 0.f, 0.f}, hc2[3] = {0.f, 0.f, 0.f};
#endif
    unsigned long long prev[NW];
#pragma unroll
    for (int wd = 0; wd < NW; ++wd) prev[wd] = 0;
    int prev_ids = -1;   // PASS 1: the previous tile's list (a list of one to four neurons is never -1: neuron 255 in all
                         // four bytes would need four equal entries)

    // The lists and regions of 64 tiles at a time, one tile per lane (a per-tile load of these wave-uniform words would
    // put a full memory round trip in front of every tile); a tile then takes its words from that lane.
    for (int q0 = q_begin; q0 < q_end; q0 += 64) {
      unsigned my_lo[NW], my_hi[NW];
      int my_reg = -1;   // region origin packed as row << 16 | column (both below 65536: checked on the host), -1: none
      int my_ids = -1, my_n = 0;   // PASS 1: the list itself (at most four neurons: one per byte) and its length
      int my_place = -1;           // qx | tile row << 16 (-1: not packed)
      {
          const int ql = min(q0 + lane, q_end - 1);
          if (PASS != 1) {   // the short-list pass needs no masks: its lists fit the descriptor
#pragma unroll
              for (int wd = 0; wd < NW; ++wd) {
                  const unsigned long long m = masks[(long)ql * NW + wd];
                  my_lo[wd] = (unsigned)m, my_hi[wd] = (unsigned)(m >> 32);
              }
          }
          const int4 dsc = descs[ql];
          my_reg = ZM == 3 ? -1 : dsc.x, my_ids = dsc.y, my_n = dsc.z, my_place = dsc.w;
      }
      if (LONGPASS) {   // nothing for this pass among these 64 tiles?
          int myn = 0;
#pragma unroll
          for (int wd = 0; wd < NW; ++wd) myn += __builtin_popcount(my_lo[wd]) + __builtin_popcount(my_hi[wd]);
          if (__ballot(myn > LISTS_NG) == 0) continue;
      }
      const int q1 = min(q0 + 64, q_end);
      for (int q = q0; q < q1; ++q) {
        const int jl = q - q0;
        unsigned long long msk[NW];
        int n = 0;  // wave-uniform
        bool same = true;
        int ids = -1;
        if (PASS == 1) {
            n = __builtin_amdgcn_readlane(my_n, jl);
            if (n == 0 || n > LISTS_NG) continue;
            ids = __builtin_amdgcn_readlane(my_ids, jl);
            same = ids == prev_ids;
            prev_ids = ids;
        } else {
#pragma unroll
            for (int wd = 0; wd < NW; ++wd) {
                msk[wd] = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)my_hi[wd], jl) << 32) |
                          (unsigned)__builtin_amdgcn_readlane((int)my_lo[wd], jl);
                n += __builtin_popcountll(msk[wd]);
                same = same && msk[wd] == prev[wd];
            }
        }
        if (n == 0 || (PASS == 1 && n > LISTS_NG) || (PASS == 2 && n <= LISTS_NG)) continue;
        DNMF_STAMP(0)   // tile bookkeeping
#ifdef DNMF_K3N_STAMPS
        st_acc[7] += 1, st_acc[8] += n > LISTS_NG, st_acc[9] += (!same || n > LISTS_NG) && run_n != 0;   // tiles, long lists, flushes
#endif
        if (!same || n > LISTS_NG) flush();
        DNMF_STAMP(1)   // reductions of a finished run
        if (PASS != 1) {
#pragma unroll
            for (int wd = 0; wd < NW; ++wd) prev[wd] = n > LISTS_NG ? 0 : msk[wd];
        }

        // the tile's column and row in the walk: from its descriptor (integer divisions by run-time values cost the
        // scalar unit ~20 instructions each)
        const int place = __builtin_amdgcn_readlane(my_place, jl);
        const int qx = place >= 0 ? (place & 0xffff) : q % p.ntx, rest = place >= 0 ? (place >> 16) : q / p.ntx;
        if (rest != row_of_c) {
            const int qz = HASZ ? rest % p.ntz : 0, qy = HASZ ? rest / p.ntz : rest;
            mono = monomials<HASZ>(0.0f, (float)((qy << lgy) + ly), (float)((qz << lgz) + lz));
#if DNMF_K3N_DIRECT
#pragma unroll
            for (int d = 0; d < (HASZ ? 3 : 2); ++d) {   // q_d = hc0 + x (hc1 + x hc2) for this lane's (y, z)
                hc2[d] = bt[12 + d];
                hc1[d] = fmaf(bt[21 + d], mono.y, bt[3 + d]);
                hc0[d] = fmaf(bt[15 + d], mono.yy, fmaf(bt[6 + d], mono.y, bt[d]));
                if (HASZ) {
                    hc1[d] = fmaf(bt[24 + d], mono.z, hc1[d]);
                    hc0[d] = fmaf(bt[27 + d], mono.yz, fmaf(bt[18 + d], mono.zz, fmaf(bt[9 + d], mono.z, hc0[d])));
                }
            }
#endif
            row_of_c = rest;
        }
        const int qz = HASZ ? rest % p.ntz : 0, qy = HASZ ? rest / p.ntz : rest;
        const int y = (qy << lgy) + ly, z = (qz << lgz) + lz;
        const bool yz_in = y < vol.Y && z < vol.Z;
        // staged gathers for this tile?  (wave-uniform)
        int reg_r0 = -1, reg_c0 = 0;
        if (ZM != 3) {
            const int rg = __builtin_amdgcn_readlane(my_reg, jl);
            reg_r0 = rg < 0 ? -1 : rg >> 16, reg_c0 = rg & 0xffff;
        }
        // lists of more than two groups keep the direct gathers (and their offsets)
        const bool staged = ZM != 3 && reg_r0 >= 0 && n <= (PASS == 2 ? 2 : 1) * LISTS_NG;
        // byte offset of volume voxel (0,0,0) inside a staged region, and of the region inside a footprint image
        const float lds_origin = 4.0f * (float)((HALO - reg_r0) * LISTS_RC + HALO * (ZM == 2 ? 2 : 1) - reg_c0);
        const unsigned reg_goff = (unsigned)(reg_r0 * hl.row4 + reg_c0 * 4);

        // staged: the region of neuron k into staging slot i (two sixteen-byte pieces per lane, 120 in all) ...
        auto stage_load = [&](int k, f32x4 (&piece)[2]) {
            const char *__restrict__ Ak = reinterpret_cast<const char *>(p.At) + (size_t)k * plane + reg_goff;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned o = (unsigned)(piece_row[j] * hl.row4 + piece_c4[j] * 16);
                asm("" : "+v"(o));
                piece[j] = (j == 0 || lane + 64 < LISTS_REGION / 4) ? *reinterpret_cast<const f32x4 *>(Ak + o)
                                                                   : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto stage_store = [&](int i, const f32x4 (&piece)[2]) {
            char *dst = stage_lds + i * (LISTS_REGION * 4);
            *reinterpret_cast<f32x4 *>(dst + lane * 16) = piece[0];
            if (lane + 64 < LISTS_REGION / 4) *reinterpret_cast<f32x4 *>(dst + (lane + 64) * 16) = piece[1];
        };
        // The region of the list's first neuron is requested here, ahead of the coordinate arithmetic, which hides its
        // latency (3.10 -> 3.05 ms per 4000 frames at 512x512, K=100).  Eight registers in flight fit; the first TWO
        // neurons' regions (sixteen) spill: 3.98 ms.  Z == 1 only: with the third coordinate chain of Z == 2 in flight the
        // eight registers are not there (151 registers wanted against 131 without).
        constexpr bool DMA = DNMF_K3N_DMA && ZM != 3;
        constexpr bool EARLY = DNMF_K3N_EARLY1 && PASS == 1 && ZM == 1 && !DMA;
        // the short list (PASS 1: it came with the tile's descriptor; the one-kernel form reads it off the mask words)
        int ks[LISTS_NG];
#pragma unroll
        for (int i = 0; i < LISTS_NG; ++i) ks[i] = -1;
        if (PASS == 1) {
#pragma unroll
            for (int i = 0; i < LISTS_NG; ++i) ks[i] = i < n ? (int)(((unsigned)ids >> (8 * i)) & 0xffu) : -1;
        } else if (PASS == 0 && n <= LISTS_NG) {
            unsigned long long rem[NW];
#pragma unroll
            for (int wd = 0; wd < NW; ++wd) rem[wd] = msk[wd];
            take_ids(rem, ks);
        }
        f32x4 early[2];
        if (EARLY && staged) {
            stage_load((int)((unsigned)ids & 0xffu), early);
            __builtin_amdgcn_sched_barrier(0);
        }
        // LDS-DMA: neuron k's region into staging slot i without passing through registers (piece e = lane + 64 j lands at
        // byte 16 e of the slot: the DMA writes lane l's 16 bytes at base + 16 l)
        auto stage_dma = [&](int k, int i) {
            const char *__restrict__ Ak = reinterpret_cast<const char *>(p.At) + (size_t)k * plane + reg_goff;
            auto dst = (__attribute__((address_space(3))) char *)(stage_lds + i * (LISTS_REGION * 4));
            __builtin_amdgcn_global_load_lds(Ak + (unsigned)(piece_row[0] * hl.row4 + piece_c4[0] * 16), dst, 16, 0, 0);
            if (lane + 64 < LISTS_REGION / 4)
                __builtin_amdgcn_global_load_lds(Ak + (unsigned)(piece_row[1] * hl.row4 + piece_c4[1] * 16), dst + 1024, 16, 0, 0);
        };
        if (DMA && PASS != 2 && staged && n <= LISTS_NG) {
            // (the last tile's reads of the slots have long returned: their values went into its sums)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < LISTS_NG; ++i)
                if (i < n) stage_dma(ks[i], i);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- taps of this lane's four voxels: byte offsets of the two x-columns of the tap cell, weights ------------
        // Z == 1: off[.][dx] = byte offset of the (y, y+1) pair of x-column dx, w[.][dy + 2 dx] the four weights.
        // Z >= 2: off[.][0] = byte offset of the base corner's z-pair, w[.] = {weight of x-corner 1, of y-corner 1,
        // of the z-pair's members}.  A staged tile keeps the LDS byte offset of the base corner inside a region in off[.][0].
        unsigned off[LISTS_VPL][HASZ ? 1 : 2];
        float w[LISTS_VPL][4];
        float yv[LISTS_VPL];
        const int xt = qx << (lgx + LISTS_LGV);  // first x of the tile
        const float x0f = (float)(xt + lx);
        const bool full = xt + (LISTS_VPL << lgx) <= vol.X && (qy << lgy) + (1 << lgy) <= vol.Y && (qz << lgz) + (1 << lgz) <= vol.Z;
        // frame values: (scalar base + 32-bit lane offset) loads; the voxels of a lane are (1 << lgx) rows apart
        const unsigned yo0 = (unsigned)(((xt + lx) * vol.Y + min(y, vol.Y - 1)) * vol.Z + min(z, vol.Z - 1)) * 4u;
        const unsigned ystep = (unsigned)((vol.Y * vol.Z) << lgx) * 4u;
        auto taps = [&](auto full_tile) {
            constexpr bool FULL = decltype(full_tile)::value;
#pragma unroll
            for (int v = 0; v < LISTS_VPL; ++v) {
                const int x = xt + (v << lgx) + lx;
                float fx, fy, wx[2], wy[2];
#if DNMF_K3N_DIRECT
                const float xf = x0f + (float)(v << lgx);
                float ud[3];
#pragma unroll
                for (int d = 0; d < (HASZ ? 3 : 2); ++d) ud[d] = fmaf(xf, fmaf(xf, hc2[d], hc1[d]), hc0[d]);
                axis_taps_halo(ud[0], hl.xhi, fx, wx[0], wx[1]);
                axis_taps_halo(ud[1], hl.yhi, fy, wy[0], wy[1]);
                const float uz = HASZ ? ud[2] : 0.0f;
#else
                Monomials<HASZ> m = mono;
                m.x = x0f + (float)(v << lgx), m.xx = __fmul_rn(m.x, m.x), m.xy = __fmul_rn(m.x, m.y);
                if (HASZ) m.xz = __fmul_rn(m.x, m.z);
                axis_taps_halo(unnormalise(normalise_axis<FAST>(poly_a<HASZ>(b2, 0, m), vol, 0), vol.hx1), hl.xhi, fx, wx[0],
                               wx[1]);
                axis_taps_halo(unnormalise(normalise_axis<FAST>(poly_a<HASZ>(b2, 1, m), vol, 1), vol.hy1), hl.yhi, fy, wy[0],
                               wy[1]);
                const float uz = HASZ ? unnormalise(normalise_axis<FAST>(poly_a<HASZ>(b2, 2, m), vol, 2), vol.hz1) : 0.0f;
#endif
                const unsigned o0 = halo_offset<F32OFF>(fx, fy, hl, hl.origin4, hl.origin4f);
                const bool in = FULL || (yz_in && x < vol.X);   // a voxel beyond the volume: weights 0, frame value 0
                if constexpr (HASZ) {
                    float wzm[2];
                    unsigned zo = 0u;
                    if (ZM == 2) {   // the pair (0, 1): common.hpp, z_pair_weights
                        z_pair_weights(uz, wzm[0], wzm[1]);
                    } else {
                        int iz;
                        float wz[2];
                        axis_weights(uz, iz, wz[0], wz[1]);
                        const int izc = clamp_index(iz, vol.Z - 1);   // the pair (izc, izc + 1) inside the volume
                        const bool same = iz == izc;
                        wzm[0] = same ? wz[0] : (iz + 1 == izc ? wz[1] : 0.0f);
                        wzm[1] = same ? wz[1] : (iz == izc + 1 ? wz[0] : 0.0f);
                        zo = (unsigned)izc * 4u;
                    }
                    // (a voxel beyond the volume is not covered by the tile's region: it reads the region's first floats,
                    // times zero)
                    off[v][0] = staged ? (in ? (unsigned)fmaf(fx, (float)(4 * LISTS_RC), fmaf(fy, 8.0f, lds_origin)) : 0u) : o0 + zo;
                    w[v][0] = wx[1], w[v][1] = wy[1], w[v][2] = in ? wzm[0] : 0.0f, w[v][3] = in ? wzm[1] : 0.0f;
                } else {
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        off[v][dx] = o0 + (dx ? (unsigned)hl.row4 : 0u);
                        if (staged && dx == 0)
                            off[v][0] = in ? (unsigned)fmaf(fx, (float)(4 * LISTS_RC), fmaf(fy, 4.0f, lds_origin)) : 0u;
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy) w[v][dy + 2 * dx] = in ? __fmul_rn(wx[dx], wy[dy]) : 0.0f;
                    }
                }
                unsigned yo = in ? yo0 + (unsigned)v * ystep : 0u;
                asm("" : "+v"(yo));
#ifdef DNMF_K3N_ABL_FRAME   // timing ablation: no frame loads
                const float val = __builtin_bit_cast(float, yo);
#else
                const float *yp = reinterpret_cast<const float *>(reinterpret_cast<const char *>(yb) + yo);
                const float val = DNMF_K3N_NT ? __builtin_nontemporal_load(yp) : *yp;
#endif
                yv[v] = in ? val : 0.0f;
#ifdef DNMF_K3N_TAPS_BARRIER
                if (HASZ) __builtin_amdgcn_sched_barrier(0);   // one voxel's coordinate chains at a time (registers)
#endif
            }
        };
        if (full)
            taps(std::true_type{});
        else
            taps(std::false_type{});
        DNMF_STAMP(2)   // coordinates, weights, frame loads issued

        // z-pair members -> y -> x for Z >= 2: quad = the (y, z0), (y, z1), (y + 1, z0), (y + 1, z1) values of x-corner 0 / 1
        auto blend8 = [&](const float (&q0)[4], const float (&q1)[4], const float (&wv)[4]) {
            const float t00 = fmaf(wv[3], q0[1], wv[2] * q0[0]), t01 = fmaf(wv[3], q0[3], wv[2] * q0[2]);
            const float t10 = fmaf(wv[3], q1[1], wv[2] * q1[0]), t11 = fmaf(wv[3], q1[3], wv[2] * q1[2]);
            const float a0 = fmaf(wv[1], t01 - t00, t00), a1 = fmaf(wv[1], t11 - t10, t10);
            return fmaf(wv[0], a1 - a0, a0);
        };
        auto eval = [&](int k, float (&a)[LISTS_VPL]) {
            const char *__restrict__ Ak = reinterpret_cast<const char *>(p.At) + (size_t)k * plane;
#pragma unroll
            for (int v = 0; v < LISTS_VPL; ++v) {
                if constexpr (HASZ) {
                    float q[2][4];
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        unsigned o = off[v][0] + (dx ? (unsigned)hl.row4 : 0u);
                        asm("" : "+v"(o));
                        const char *src = Ak + o;
                        if constexpr (ZM == 2) {
                            const f32x4_a8 t = *reinterpret_cast<const f32x4_a8 *>(src);
                            q[dx][0] = t.x, q[dx][1] = t.y, q[dx][2] = t.z, q[dx][3] = t.w;
                        } else {
                            const f32x2_a4 t0 = *reinterpret_cast<const f32x2_a4 *>(src);
                            const f32x2_a4 t1 = *reinterpret_cast<const f32x2_a4 *>(src + hl.col4);
                            q[dx][0] = t0.x, q[dx][1] = t0.y, q[dx][2] = t1.x, q[dx][3] = t1.y;
                        }
                    }
                    a[v] = blend8(q[0], q[1], w[v]);
                } else {
                    float s = 0.0f;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        // the offsets are re-materialised as 32-bit values here so that the loads take the
                        // (scalar base + 32-bit vector offset) form; hoisted out of the neuron loop they become 64-bit pairs
                        unsigned o = off[v][e];
                        asm("" : "+v"(o));
                        const char *src = Ak + o;
                        const float s0 = *reinterpret_cast<const float *>(src);
                        const float s1 = *reinterpret_cast<const float *>(src + 4);
                        s = fmaf(s0, w[v][2 * e], s);
                        s = fmaf(s1, w[v][2 * e + 1], s);
                    }
                    a[v] = s;
                }
            }
        };
        // ... and the warped values of the lane's voxels from there
        auto eval_staged = [&](int i, float (&a)[LISTS_VPL]) {
            const char *src0 = stage_lds + i * (LISTS_REGION * 4);
#pragma unroll
            for (int v = 0; v < LISTS_VPL; ++v) {
                const char *src = src0 + off[v][0];
                if constexpr (HASZ) {   // Z == 2: four eight-byte reads (the z-pairs of the corners)
                    float q[2][4];
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const float2 t0 = *reinterpret_cast<const float2 *>(src + dx * (4 * LISTS_RC));
                        const float2 t1 = *reinterpret_cast<const float2 *>(src + dx * (4 * LISTS_RC) + 8);
                        q[dx][0] = t0.x, q[dx][1] = t0.y, q[dx][2] = t1.x, q[dx][3] = t1.y;
                    }
                    a[v] = blend8(q[0], q[1], w[v]);
                } else {
                    float s = 0.0f;
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const float s0 = *reinterpret_cast<const float *>(src + dx * (4 * LISTS_RC));
                        const float s1 = *reinterpret_cast<const float *>(src + dx * (4 * LISTS_RC) + 4);
                        s = fmaf(s0, w[v][2 * dx], s);
                        s = fmaf(s1, w[v][2 * dx + 1], s);
                    }
                    a[v] = s;
                }
            }
        };
        auto dot4 = [&](const float (&a)[LISTS_VPL], const float (&cc)[LISTS_VPL], float init) {
            float s = init;
#pragma unroll
            for (int v = 0; v < LISTS_VPL; ++v) s = fmaf(a[v], cc[v], s);
            return s;
        };

        n_eval += n, n_pair += n * (n + 1) / 2;
        if (PASS != 2 && n <= LISTS_NG) {
            // the usual case: the whole list in registers; sums join the pending run (same list) or start one
            const bool fresh = run_n == 0;
            auto go = [&](auto nn) {
                constexpr int N = decltype(nn)::value;
                float a[N][LISTS_VPL];
                if (staged && DMA) {
                    // The regions have landed: they were requested BEFORE the tile's LISTS_VPL frame values (one load each, issued
                    // by taps() of a full tile, none in between: sched_barrier after the requests, this statement a compiler
                    // barrier), and vector memory returns in order -- "at most LISTS_VPL outstanding" = every region is in LDS,
                    // while the frame values (from HBM, the longest latency of the tile) may still be on their way: they are not
                    // needed until the sums after the taps, where the compiler places its own wait.  (2.98 -> 2.92 ms per 4000
                    // frames of 512x512, K = 100; nothing at Z = 2.)
#if DNMF_K3N_LATE_FRAMES
                    if (full)
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LISTS_VPL) : "memory");
                    else
#endif
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        eval_staged(i, a[i]);
                        if (HASZ) __builtin_amdgcn_sched_barrier(0);
                    }
                } else if (staged) {
                    // the regions are requested two neurons at a time (four would hold 32 registers for the pieces)
#pragma unroll
                    for (int i0 = 0; i0 < N; i0 += 2) {
                        f32x4 piece[2][2];
#ifndef DNMF_K3N_ABL_STAGE   // timing ablation: taps from whatever the LDS holds
#pragma unroll
                        for (int i = i0; i < N && i < i0 + 2; ++i) {
                            if (EARLY && i == 0) {
                                piece[0][0] = early[0], piece[0][1] = early[1];
                                continue;
                            }
                            stage_load(ks[i], piece[i - i0]);
                        }
#pragma unroll
                        for (int i = i0; i < N && i < i0 + 2; ++i) stage_store(i, piece[i - i0]);
#endif
                    }
#ifdef DNMF_K3N_STAMPS
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
                    DNMF_STAMP(3)   // regions requested, arrived (with the frame values), stored
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        eval_staged(i, a[i]);
                        // Z >= 2: a neuron's taps are 32 values per lane; all four neurons' reads hoisted together spill
                        if (HASZ) __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < N; ++i) {   // Z == 1: the rows of all N neurons are requested together
                        eval(ks[i], a[i]);
                        if (HASZ) __builtin_amdgcn_sched_barrier(0);   // (32 values per neuron and lane)
                    }
                }
                int e = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    acc_r[i] = dot4(a[i], yv, fresh ? 0.0f : acc_r[i]);
#pragma unroll
                    for (int j = i; j < N; ++j, ++e) acc_p[e] = dot4(a[i], a[j], fresh ? 0.0f : acc_p[e]);
                }
            };
            using std::integral_constant;
            switch (n) {
                case 1: go(integral_constant<int, 1>{}); break;
                case 2: go(integral_constant<int, 2>{}); break;
                case 3: go(integral_constant<int, 3>{}); break;
                default: go(integral_constant<int, 4>{}); break;
            }
#pragma unroll
            for (int i = 0; i < LISTS_NG; ++i) run_k[i] = ks[i];
            run_n = n;
#ifdef DNMF_K3N_STAMPS
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            asm volatile("" ::"v"(acc_r[0]), "v"(acc_p[0]));
#endif
            DNMF_STAMP(4)   // taps from LDS (or direct gathers), per-lane sums
            continue;
        }
        if constexpr (PASS != 1) {
        if (PASS == 2 && staged) {
            // Five to eight neurons on a staged tile: two groups A (four) and B through the same staging slots.  A's
            // values stay in registers while its own sums are reduced like a finished run; then B's regions replace
            // A's in LDS and every B neuron is summed against A, the frame and the B neurons before it.  (With direct
            // gathers, every neuron of B evaluated twice and a table lookup in front of every group of sums these
            // tiles -- 4.8 % of all at 512x512, K=100 -- took 17.6 % of the kernel's time.)
            unsigned long long rem[NW];
#pragma unroll
            for (int wd = 0; wd < NW; ++wd) rem[wd] = msk[wd];
            int kA[LISTS_NG], kB[LISTS_NG];
            take_ids(rem, kA);
            take_ids(rem, kB);
            const int nB = n - LISTS_NG;
            float aA[LISTS_NG][LISTS_VPL], aB[LISTS_NG][LISTS_VPL];
            if (DMA && DNMF_K3N_DMA2) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < LISTS_NG; ++i) stage_dma(kA[i], i);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int i0 = 0; i0 < LISTS_NG; i0 += 2) {
                    f32x4 piece[2][2];
#pragma unroll
                    for (int i = i0; i < i0 + 2; ++i) stage_load(kA[i], piece[i - i0]);
#pragma unroll
                    for (int i = i0; i < i0 + 2; ++i) stage_store(i, piece[i - i0]);
                }
            }
#pragma unroll
            for (int i = 0; i < LISTS_NG; ++i) {
                eval_staged(i, aA[i]);
                if (HASZ) __builtin_amdgcn_sched_barrier(0);
            }
            {
                int e = 0;
#pragma unroll
                for (int i = 0; i < LISTS_NG; ++i) {
                    acc_r[i] = dot4(aA[i], yv, 0.0f);
#pragma unroll
                    for (int j = i; j < LISTS_NG; ++j, ++e) acc_p[e] = dot4(aA[i], aA[j], 0.0f);
                    run_k[i] = kA[i];
                }
                run_n = LISTS_NG;
                flush();
            }
            if (DMA && DNMF_K3N_DMA2) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // group A's values are in registers
#pragma unroll
                for (int i = 0; i < LISTS_NG; ++i)
                    if (i < nB) stage_dma(kB[i], i);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int i0 = 0; i0 < LISTS_NG; i0 += 2) {
                    if (i0 >= nB) break;   // wave-uniform
                    f32x4 piece[2][2];
#pragma unroll
                    for (int i = i0; i < i0 + 2; ++i) stage_load(max(kB[i], 0), piece[i - i0]);   // past the list: neuron 0, unused
#pragma unroll
                    for (int i = i0; i < i0 + 2; ++i) stage_store(i, piece[i - i0]);
                }
            }
#pragma unroll
            for (int j = 0; j < LISTS_NG; ++j) {
                if (j >= nB) break;   // wave-uniform
                eval_staged(j, aB[j]);
                int sc[LISTS_NG], sb[LISTS_NG];
#pragma unroll
                for (int i = 0; i < LISTS_NG; ++i) sc[i] = pair_slot_of(kA[i], kB[j]);
#pragma unroll
                for (int i = 0; i <= j; ++i) sb[i] = pair_slot_of(kB[i], kB[j]);
                float tc[LISTS_NG], tb[LISTS_NG];
#pragma unroll
                for (int i = 0; i < LISTS_NG; ++i) tc[i] = wave_sum_last(dot4(aA[i], aB[j], 0.0f));
#pragma unroll
                for (int i = 0; i <= j; ++i) tb[i] = wave_sum_last(dot4(aB[i], aB[j], 0.0f));
                const float tr = wave_sum_last(dot4(aB[j], yv, 0.0f));
#pragma unroll
                for (int i = 0; i < LISTS_NG; ++i) add_slot(sc[i], tc[i]);
#pragma unroll
                for (int i = 0; i <= j; ++i) add_slot(sb[i], tb[i]);
                add_slot(kB[j], tr);
            }
            DNMF_STAMP(6)   // a long-list tile, after its coordinates
            continue;
        }
        // longer lists, 3-D volumes, tiles without a region: groups of LISTS_NG neurons with direct gathers, every group
        // against itself and against every later group; reduced and added tile by tile
        unsigned long long rem1[NW];
#pragma unroll
        for (int wd = 0; wd < NW; ++wd) rem1[wd] = msk[wd];
        for (int g1 = 0; g1 < n; g1 += LISTS_NG) {
            int kA[LISTS_NG];
            float aA[LISTS_NG][LISTS_VPL];
            take_ids(rem1, kA);
#pragma unroll
            for (int i = 0; i < LISTS_NG; ++i)
                if (kA[i] >= 0) {
                    eval(kA[i], aA[i]);
                } else {
#pragma unroll
                    for (int v = 0; v < LISTS_VPL; ++v) aA[i][v] = 0.0f;
                }
#pragma unroll
            for (int i = 0; i < LISTS_NG; ++i) {
                if (kA[i] < 0) continue;
                add_slot(kA[i], wave_sum_last(dot4(aA[i], yv, 0.0f)));
#pragma unroll
                for (int j = i; j < LISTS_NG; ++j)
                    if (kA[j] >= 0) add_slot(pair_slot_of(kA[i], kA[j]), wave_sum_last(dot4(aA[i], aA[j], 0.0f)));
            }
            unsigned long long rem2[NW];
#pragma unroll
            for (int wd = 0; wd < NW; ++wd) rem2[wd] = rem1[wd];
            for (int g2 = g1 + LISTS_NG; g2 < n; g2 += LISTS_NG) {
                int kB[LISTS_NG];
                take_ids(rem2, kB);
#pragma unroll
                for (int j = 0; j < LISTS_NG; ++j) {
                    if (kB[j] < 0) continue;
                    float aB[LISTS_VPL];
                    eval(kB[j], aB);
                    int sl[LISTS_NG];
#pragma unroll
                    for (int i = 0; i < LISTS_NG; ++i) sl[i] = pair_slot_of(kA[i], kB[j]);
                    float sp[LISTS_NG];
#pragma unroll
                    for (int i = 0; i < LISTS_NG; ++i) sp[i] = wave_sum_last(dot4(aA[i], aB, 0.0f));
#pragma unroll
                    for (int i = 0; i < LISTS_NG; ++i)
                        if (kA[i] >= 0) add_slot(sl[i], sp[i]);
                }
            }
        }
        DNMF_STAMP(6)   // a long-list tile, after its coordinates
        }
      }
    }
    flush();

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float *out = p.slab + ((long)b * p.tables + (LONGPASS ? p.nchunks : 0) + chunk) * p.nslot;
    for (int i = lane; i < p.nslot; i += 64) out[i] = tab[i];
    if (p.counters && lane == 0) {
        atomicAdd(&p.counters[0], n_eval);
        atomicAdd(&p.counters[1], n_pair);
#ifdef DNMF_K3N_STAMPS
        DNMF_STAMP(5)
        for (int i = 0; i < 10; ++i) atomicAdd(&p.counters[2 + i], st_acc[i]);
#endif
    }
}

#ifndef DNMF_K3N_TU_Z
// G[b] (K,K), r[b] (K) <- ordered sum of the chunk tables of frame b
__global__ __launch_bounds__(256) void gram_lists_finish_kernel(const float *__restrict__ slab, int nchunks, int nslot,
                                                                const int *__restrict__ pair_slot, int K,
                                                                float *__restrict__ G, float *__restrict__ r) {
    const int b = blockIdx.x;
    const float *src = slab + (long)b * nchunks * nslot;
    for (int e = threadIdx.x; e < K * K + K; e += blockDim.x) {
        const int slot = e < K * K ? pair_slot[e] : e - K * K;
        float s = 0.0f;
        if (slot != nslot - 1)
            for (int c = 0; c < nchunks; ++c) s += src[(long)c * nslot + slot];
        if (e < K * K)
            G[(long)b * K * K + e] = s;
        else
            r[(long)b * K + (e - K * K)] = s;
    }
}

// ---- layout ----------------------------------------------------------------------------------------------
__global__ void lists_init_kernel(int *bbox, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K * 6) bbox[i] = (i & 1) ? -1 : 0x7fffffff;
}

// At[k][halo(p)] = A[p][k] through a 32x32 LDS tile (the border of At is zeroed beforehand); bbox[k] grows over the
// non-zeros
__global__ __launch_bounds__(256) void lists_transpose_kernel(const float *__restrict__ A, long P, int K, Volume vol,
                                                              HaloLayout hl, float *__restrict__ At,
                                                              int *__restrict__ bbox) {
    __shared__ float tile[32][33];
    const long p0 = (long)blockIdx.x * 32;
    const int k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const long pp = p0 + i;
        const int k = k0 + tx;
        tile[i][tx] = (pp < P && k < K) ? A[pp * K + k] : 0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i;
        const long pp = p0 + tx;
        const bool in = k < K && pp < P;
        const float v = in ? tile[tx][i] : 0.0f;
        int x = 0, y = 0, z = 0;
        if (in) {
            voxel_xyz(pp, vol, x, y, z);
            At[(long)k * hl.Pp + (long)(x + HALO) * hl.rowf + (long)(y + HALO) * vol.Z + z] = v;
        }
        // the box grows by what the 32 lanes that hold neuron k found, one set of atomics per half wave with a non-zero
        // (one per non-zero voxel kept this kernel at 2.9 ms per call at 512x512, K=100: 76 GB/s)
        const bool nz = v != 0.0f;
        const unsigned long long any = __ballot(nz);
        if (any == 0) continue;   // wave-uniform
        int lo[3] = {nz ? x : 0x7fffffff, nz ? y : 0x7fffffff, nz ? z : 0x7fffffff};
        int hi[3] = {nz ? x : -1, nz ? y : -1, nz ? z : -1};
#pragma unroll
        for (int off = 16; off > 0; off >>= 1)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                lo[d] = min(lo[d], __shfl_xor(lo[d], off, 32));
                hi[d] = max(hi[d], __shfl_xor(hi[d], off, 32));
            }
        if (tx == 0 && hi[0] >= 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) atomicMin(&bbox[k * 6 + 2 * d], lo[d]), atomicMax(&bbox[k * 6 + 2 * d + 1], hi[d]);
        }
    }
}

// pattern of G and its slots: one block, thread k owns row k.  Slots: [0,K) rhs, then (k,k), then (k,l>k) in order.
__global__ __launch_bounds__(256) void lists_pairs_kernel(const int *__restrict__ bbox, int K, int *__restrict__ pair_slot,
                                                          int *__restrict__ nslot_out) {
    __shared__ int cnt[256];
    __shared__ int base[257];
    const int k = threadIdx.x;
    auto meets = [&](int a, int c) {
        bool ok = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int alo = bbox[a * 6 + 2 * d], ahi = bbox[a * 6 + 2 * d + 1];
            const int clo = bbox[c * 6 + 2 * d], chi = bbox[c * 6 + 2 * d + 1];
            ok = ok && alo <= ahi && clo <= chi && alo - 1 <= chi && clo - 1 <= ahi;
        }
        return ok;
    };
    int n = 0;
    if (k < K) {
        n = 1;  // (k,k)
        for (int l = k + 1; l < K; ++l) n += meets(k, l) ? 1 : 0;
    }
    cnt[k] = n;
    __syncthreads();
    if (k == 0) {
        int s = K;
        for (int i = 0; i < 256; ++i) base[i] = s, s += cnt[i];
        base[256] = s;  // trash slot
        *nslot_out = s + 1;
    }
    __syncthreads();
    if (k < K) {
        const int trash = base[256];
        int s = base[k];
        pair_slot[k * K + k] = s++;
        for (int l = k + 1; l < K; ++l) {
            const int v = meets(k, l) ? s++ : trash;
            pair_slot[k * K + l] = v;
            pair_slot[l * K + k] = v;
        }
    }
}

static void lists_tile_shape(const Volume &vol, int &lgx, int &lgy, int &lgz, int &ntx, int &nty, int &ntz, int &ntiles) {
    lgz = vol.Z == 1 ? 0 : (vol.Z == 2 ? 1 : 2);
    lgy = vol.Z == 1 ? 5 : 4;
    lgx = 6 - lgy - lgz;
    const int tx = LISTS_VPL << lgx, ty = 1 << lgy, tz = 1 << lgz;
    ntx = (vol.X + tx - 1) / tx;
    ntz = (vol.Z + tz - 1) / tz;
    nty = (vol.Y + ty - 1) / ty;
    ntiles = ntx * nty * ntz;
}

// DNMF_LISTS_CHUNKS=n in the environment (1 .. 64) asks for n chunks per frame instead: the parity tests use it to give a
// wave of a small problem the long runs of tiles (and of equal lists) it has at the bench size.
static void lists_choose_chunks(int ntiles, int B, int &nchunks, int &chunk_len) {
    long want = (LISTS_ITEMS + B - 1) / B;  // wave-sized work items: several rounds of four waves per SIMD
    if (const char *e = getenv("DNMF_LISTS_CHUNKS")) {
        const long n = strtol(e, nullptr, 10);
        if (n >= 1 && n <= 64 && n <= want) want = n;   // never more tables than the workspace was sized for
    }
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    if (want > ntiles) want = ntiles;
    chunk_len = (int)((ntiles + want - 1) / want);
    nchunks = (ntiles + chunk_len - 1) / chunk_len;
}

static int lists_words(int K) { return K <= 64 ? 1 : (K <= 128 ? 2 : 4); }

// One kernel or two?  A second launch has its fixed cost per wave (table, slab, the scan for its tiles): it pays when a
// wave has a long run of tiles -- 512x512, K=100: 205 tiles per wave at 4000 frames (3.6 ms against 3.7), 25 at 400
// frames (0.89 ms against 0.51 in one kernel); 256x256x1000, K=50: 16 tiles per wave, 0.92 against 0.58.  The density
// of the footprints does not decide it: 512x512x4000 with K=200 (45 % of the tiles list more than four neurons) takes
// 9.1 ms in two launches and 12.2 in one.
// DNMF_LISTS_PASSES=1|2 in the environment overrides the choice (the parity tests run both forms on small problems).
static int lists_passes(int chunk_len) {
    if (const char *e = getenv("DNMF_LISTS_PASSES")) {
        if (e[0] == '1' && e[1] == 0) return 1;
        if (e[0] == '2' && e[1] == 0) return 2;
    }
    return chunk_len < 100 ? 1 : 2;
}

#endif  // DNMF_K3N_TU_Z

// The stream the second pass runs on and the two events of its fork / join, one set per device, made on first use and
// kept (the only state this file holds), with the mutex that serialises the fork / join of one device -- shared by every
// instantiation and both translation units: the events are shared, and a wait takes whatever was last recorded on its
// event when it is enqueued.
struct SideStream {
    std::mutex lock;
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
SideStream &side_stream();   // locks nothing; callers hold .lock around the fork / join
#ifndef DNMF_K3N_TU_Z
SideStream &side_stream() {
    static SideStream per_device[64];
    static std::mutex create;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    SideStream &ss = per_device[dev];
    std::lock_guard<std::mutex> hold(create);
    if (!ss.stream) {
        hipStream_t s = nullptr;
        hipEvent_t a = nullptr, b = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&a, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&b, hipEventDisableTiming) == hipSuccess)
            ss.stream = s, ss.fork = a, ss.join = b;
    }
    return ss;
}
#endif

template <int ZM, int NW, int FAST, bool F32OFF>
static void launch_lists_passes(const ListParams &p, unsigned nwg, size_t lds, hipStream_t st) {
    if (p.tables == p.nchunks) {
        hipLaunchKernelGGL((warp_gram_lists_kernel<ZM, NW, FAST, F32OFF, 0>), dim3(nwg), dim3(256), lds, st, p);
        return;
    }
    // fork: the second pass on the side stream behind the lists, join before anything that follows on `st`
    // (one host thread at a time through the fork / join of a device)
    SideStream &ss = side_stream();
    std::lock_guard<std::mutex> hold(ss.lock);
    const bool forked = ss.stream && hipEventRecord(ss.fork, st) == hipSuccess &&
                        hipStreamWaitEvent(ss.stream, ss.fork, 0) == hipSuccess;
    if (forked) {
        hipLaunchKernelGGL((warp_gram_lists_kernel<ZM, NW, FAST, F32OFF, 2>), dim3(nwg), dim3(256), lds, ss.stream, p);
        const bool recorded = hipEventRecord(ss.join, ss.stream) == hipSuccess;
        hipLaunchKernelGGL((warp_gram_lists_kernel<ZM, NW, FAST, F32OFF, 1>), dim3(nwg), dim3(256), lds, st, p);
        if (!recorded || hipStreamWaitEvent(st, ss.join, 0) != hipSuccess)
            (void)hipStreamSynchronize(ss.stream);   // the join could not be enqueued: wait for the side stream here
    } else {   // no side stream to be had: one after the other
        hipLaunchKernelGGL((warp_gram_lists_kernel<ZM, NW, FAST, F32OFF, 1>), dim3(nwg), dim3(256), lds, st, p);
        hipLaunchKernelGGL((warp_gram_lists_kernel<ZM, NW, FAST, F32OFF, 2>), dim3(nwg), dim3(256), lds, st, p);
    }
}

template <int ZM, int NW>
static void launch_lists_t(const ListParams &p, unsigned nwg, size_t lds, hipStream_t st) {
    const long nthreads = (long)p.B * p.ntiles;
    hipLaunchKernelGGL((lists_tilemask_kernel<NW>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, p);
    if (p.vol.fastdiv && p.hl.f32off)
        launch_lists_passes<ZM, NW, 1, true>(p, nwg, lds, st);
    else if (p.vol.fastdiv)
        launch_lists_passes<ZM, NW, 1, false>(p, nwg, lds, st);
    else
        launch_lists_passes<ZM, NW, 0, false>(p, nwg, lds, st);
}

// The Z >= 2 instantiations are compiled in warp_gram_lists_z.hip (this file again, under DNMF_K3N_TU_Z) with
// -fno-slp-vectorize: the SLP vectoriser packs their blends into v_pk_* operations that need their operands in register
// pairs, and the kernels then want 200 registers instead of 150 (19.9 ms per 4000 frames at 512x512x2 with the spills
// against 6.6); the Z == 1 kernels, tuned with the vectoriser on, lose 15 % without it.
void launch_lists_z(const ListParams &p, unsigned nwg, size_t lds, hipStream_t st, int nw);
#ifdef DNMF_K3N_TU_Z
void launch_lists_z(const ListParams &p, unsigned nwg, size_t lds, hipStream_t st, int nw) {
    if (p.vol.Z > 2) {
        if (nw == 1) launch_lists_t<3, 1>(p, nwg, lds, st);
        else if (nw == 2) launch_lists_t<3, 2>(p, nwg, lds, st);
        else launch_lists_t<3, 4>(p, nwg, lds, st);
    } else {
        if (nw == 1) launch_lists_t<2, 1>(p, nwg, lds, st);
        else if (nw == 2) launch_lists_t<2, 2>(p, nwg, lds, st);
        else launch_lists_t<2, 4>(p, nwg, lds, st);
    }
}
#endif

}  // namespace dnmf

#ifndef DNMF_K3N_TU_Z
extern "C" {

size_t dnmf_lists_axis_masks_bytes(int X, int Y, int Z, int K) {
    if (X <= 0 || Y <= 0 || Z <= 0 || K <= 0 || K > 64 * dnmf::LISTS_MAXW) return 0;
    return (size_t)dnmf::axis_masks_entries(dnmf::make_volume(X, Y, Z)) * dnmf::lists_words(K) * sizeof(unsigned long long);
}

int dnmf_pack_footprints_lists(const float *A, int X, int Y, int Z, int K, float *At, int *bbox, int *pair_slot,
                               int *nslot, void *axis_masks, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(A && At && bbox && pair_slot && nslot && axis_masks, DNMF_E_NULL, "dnmf_pack_footprints_lists: NULL buffer");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && K > 0, DNMF_E_SHAPE, "dnmf_pack_footprints_lists: X=%d Y=%d Z=%d K=%d", X, Y, Z,
                 K);
    DNMF_REQUIRE(K <= 64 * LISTS_MAXW, DNMF_E_UNSUPPORTED, "dnmf_pack_footprints_lists: K=%d > %d", K, 64 * LISTS_MAXW);
    const Volume vol = make_volume(X, Y, Z);
    const HaloLayout hl = make_halo_layout(X, Y, Z);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(At, 0, (size_t)K * hl.Pp * sizeof(float), st);
    DNMF_REQUIRE(e == hipSuccess, (int)e, "dnmf_pack_footprints_lists: hipMemsetAsync: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(lists_init_kernel, dim3((K * 6 + 255) / 256), dim3(256), 0, st, bbox, K);
    hipLaunchKernelGGL(lists_transpose_kernel, dim3((unsigned)((vol.P + 31) / 32), (unsigned)((K + 31) / 32)), dim3(256), 0,
                       st, A, vol.P, K, vol, hl, At, bbox);
    hipLaunchKernelGGL(lists_pairs_kernel, dim3(1), dim3(256), 0, st, bbox, K, pair_slot, nslot);
    hipLaunchKernelGGL(lists_axis_masks_kernel, dim3((unsigned)((axis_masks_entries(vol) + 255) / 256)), dim3(256), 0, st, bbox,
                       K, vol, lists_words(K), static_cast<unsigned long long *>(axis_masks));
    return check_launch("dnmf_pack_footprints_lists");
}

// workspace = slot tables (B, nchunks, nslot) floats, then the tile lists (B, ntiles, NW) 64-bit words
static size_t lists_slab_bytes(int nslot, int B) {
    long want = (dnmf::LISTS_ITEMS + B - 1) / B;
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    return ((size_t)B * (size_t)(2 * want) * (size_t)nslot * sizeof(float) + 255) / 256 * 256;   // up to two tables per chunk
}

// the tiles' list words, rounded up so that the 16-byte descriptors behind them are aligned
static size_t lists_masks_bytes(int K, int B, int ntiles) {
    return ((size_t)B * ntiles * dnmf::lists_words(K) * sizeof(unsigned long long) + 15) / 16 * 16;
}

size_t dnmf_warp_gram_rhs_lists_workspace(int nslot, int K, int X, int Y, int Z, int B) {
    using namespace dnmf;
    if (nslot <= 0 || B <= 0 || K <= 0 || K > 64 * LISTS_MAXW || X <= 0 || Y <= 0 || Z <= 0) return 0;
    int lgx, lgy, lgz, ntx, nty, ntz, ntiles;
    lists_tile_shape(make_volume(X, Y, Z), lgx, lgy, lgz, ntx, nty, ntz, ntiles);
    return lists_slab_bytes(nslot, B) + lists_masks_bytes(K, B, ntiles) + (size_t)B * ntiles * sizeof(int4);
}

int dnmf_warp_gram_rhs_lists(const float *At, const int *bbox, const int *pair_slot, const void *axis_masks, int nslot, int K,
                             int X, int Y, int Z, const float *beta, int T, const int *times, int B, const float *frames,
                             long ldf, const int *frame_ids, float *G, float *r, void *workspace, size_t workspace_bytes,
                             unsigned long long *counters, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(At && bbox && pair_slot && axis_masks && beta && frames && workspace && (!G == !r), DNMF_E_NULL,
                 "dnmf_warp_gram_rhs_lists: NULL buffer");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && K > 0 && T > 0 && B > 0 && nslot > K, DNMF_E_SHAPE,
                 "dnmf_warp_gram_rhs_lists: X=%d Y=%d Z=%d K=%d T=%d B=%d nslot=%d", X, Y, Z, K, T, B, nslot);
    DNMF_REQUIRE(K <= 64 * LISTS_MAXW, DNMF_E_UNSUPPORTED, "dnmf_warp_gram_rhs_lists: K=%d > %d", K, 64 * LISTS_MAXW);
    DNMF_REQUIRE(nslot <= LISTS_MAX_SLOTS, DNMF_E_UNSUPPORTED,
                 "dnmf_warp_gram_rhs_lists: %d pattern slots > %d (footprints overlap too much for this kernel)", nslot,
                 LISTS_MAX_SLOTS);
    ListParams p;
    p.vol = make_volume(X, Y, Z);
    p.hl = make_halo_layout(X, Y, Z);
    DNMF_REQUIRE(ldf >= p.vol.P, DNMF_E_SHAPE, "dnmf_warp_gram_rhs_lists: ldf=%ld < P=%ld", ldf, p.vol.P);
    DNMF_REQUIRE(p.hl.Pp < (1L << 29) && p.hl.row4 < (1 << 23) && p.hl.Xp < (1 << 23), DNMF_E_UNSUPPORTED,
                 "dnmf_warp_gram_rhs_lists: volume %dx%dx%d too large for 32-bit tap offsets", X, Y, Z);
    p.At = At, p.bbox = bbox, p.pair_slot = pair_slot, p.nslot = nslot, p.K = K;
    p.axis_masks = static_cast<const unsigned long long *>(axis_masks);
    p.beta = beta, p.T = T, p.times = times, p.B = B;
    p.frames = frames, p.ldf = ldf, p.frame_ids = frame_ids;
    p.slab = static_cast<float *>(workspace);
    p.tile_masks = reinterpret_cast<unsigned long long *>(static_cast<char *>(workspace) + lists_slab_bytes(nslot, B));
    p.counters = counters;
    lists_tile_shape(p.vol, p.lgx, p.lgy, p.lgz, p.ntx, p.nty, p.ntz, p.ntiles);
    lists_choose_chunks(p.ntiles, B, p.nchunks, p.chunk_len);
    p.tables = p.nchunks * lists_passes(p.chunk_len);
    DNMF_REQUIRE(workspace_bytes >= dnmf_warp_gram_rhs_lists_workspace(nslot, K, X, Y, Z, B), DNMF_E_WORKSPACE,
                 "dnmf_warp_gram_rhs_lists: workspace %zu < %zu bytes", workspace_bytes,
                 dnmf_warp_gram_rhs_lists_workspace(nslot, K, X, Y, Z, B));
    hipStream_t st = (hipStream_t)stream;
    const long nitems = (long)p.nchunks * B;
    const unsigned nwg = (unsigned)((nitems + 3) / 4);
    const int nw = lists_words(K);
    p.tile_desc = reinterpret_cast<int4 *>(reinterpret_cast<char *>(p.tile_masks) + lists_masks_bytes(K, B, p.ntiles));
    // four slot tables (padded to 16 bytes), then for Z <= 2 the four waves' staging regions
    const size_t lds = (((size_t)4 * nslot + 3) & ~(size_t)3) * sizeof(float) +
                       (Z <= 2 ? (size_t)4 * LISTS_NG * LISTS_REGION * sizeof(float) : 0);
    if (Z > 1) {
        launch_lists_z(p, nwg, lds, st, nw);
    } else {
        if (nw == 1) launch_lists_t<1, 1>(p, nwg, lds, st);
        else if (nw == 2) launch_lists_t<1, 2>(p, nwg, lds, st);
        else launch_lists_t<1, 4>(p, nwg, lds, st);
    }
    if (G)
        hipLaunchKernelGGL(gram_lists_finish_kernel, dim3((unsigned)B), dim3(256), 0, st, p.slab, p.tables, nslot,
                           pair_slot, K, G, r);
    return check_launch("dnmf_warp_gram_rhs_lists");
}

int dnmf_warp_gram_rhs_lists_chunks(int X, int Y, int Z, int B) {
    using namespace dnmf;
    if (X <= 0 || Y <= 0 || Z <= 0 || B <= 0) return 0;
    const Volume vol = make_volume(X, Y, Z);
    int lgx, lgy, lgz, ntx, nty, ntz, ntiles, nchunks, chunk_len;
    lists_tile_shape(vol, lgx, lgy, lgz, ntx, nty, ntz, ntiles);
    lists_choose_chunks(ntiles, B, nchunks, chunk_len);
    return nchunks * lists_passes(chunk_len);
}

}  // extern "C"
#endif  // DNMF_K3N_TU_Z
