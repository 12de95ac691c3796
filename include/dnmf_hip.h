/*
 * dnmf_hip.h -- C ABI of libdnmf_hip.so: the MI355X (gfx950) kernels of the deformable-NMF hot path.
 *
 * The reference (mathdiane/dNMF) has no FFI or plugin interface: its hot path is stock torch / numpy
 * calls inside Demix/dNMF.py.  Each entry point below therefore replaces a group of those calls and
 * cites them (paths relative to the reference root).  The Python mirror of the reference classes
 * (dnmf_amd/Demix/dNMF.py) is the only caller; INTEGRATION.md shows the ctypes stub a maintainer of
 * the reference would add to call the same functions from the original file.
 *
 * Conventions
 *   - plain C, no torch types; every pointer is a DEVICE pointer unless it says "host";
 *   - the caller owns every buffer (torch allocates them); the library keeps no state between calls
 *     other than the thread-local text behind dnmf_last_error();
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     every call only enqueues work on it and returns -- no host synchronisation, no allocation;
 *   - return value: 0 = ok, negative = argument error (DNMF_E_*), positive = hipError_t;
 *   - layouts are the reference's: a volume is (X,Y,Z) row-major with voxel index
 *     p = (x*Y + y)*Z + z, P = X*Y*Z; footprints A are (P,K) (= torch (X,Y,Z,K) contiguous,
 *     Demix/dNMF.py:39-40); beta is (10,3,T) contiguous (Demix/dNMF.py:24-27); traces C are (K,T)
 *     with row stride ldc (Demix/dNMF.py:130); a frame is P floats;
 *   - Z == 1 means "z pinned to slice 0" (the reference divides 0/0 there; see oracle/dnmf_oracle.py);
 *   - HALO LAYOUT: the two single-channel images the hot kernels gather from -- the reconstruction images S
 *     (read by K2) and the neuron-major footprints At (read by K3n) -- carry a zero border of DNMF_HALO voxels
 *     around x and y: voxel (x,y,z) lives at (x + DNMF_HALO)*row + (y + DNMF_HALO)*Z + z with
 *     row = dnmf_halo_row(Y,Z) = (Y + 2*DNMF_HALO)*Z rounded up to a multiple of 32 floats (rows start on 128-byte
 *     lines), and an image has dnmf_halo_voxels(X,Y,Z) = (X + 2*DNMF_HALO)*row floats.  A trilinear tap outside the
 *     volume then reads a zero, which is what grid_sample's zero padding (Demix/dNMF.py:57) contributes, without
 *     per-corner bounds tests.  The kernels that WRITE such images (dnmf_recon_image*, dnmf_pack_footprints_lists)
 *     write the border (and the alignment excess of every row) too.
 */
#ifndef DNMF_HIP_H
#define DNMF_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DNMF_ABI_VERSION 6

#define DNMF_OK 0
#define DNMF_E_NULL (-1)      /* required pointer is NULL */
#define DNMF_E_SHAPE (-2)     /* non-positive or inconsistent size */
#define DNMF_E_UNSUPPORTED (-3) /* valid request outside what this build handles (e.g. K too large) */
#define DNMF_E_WORKSPACE (-4) /* workspace too small */

typedef void *dnmf_stream_t;

#define DNMF_HALO 2
/* Floats of one row / of one whole halo-layout image of an X x Y x Z volume (0 on bad sizes). */
int dnmf_halo_row(int Y, int Z);
long dnmf_halo_voxels(int X, int Y, int Z);

/* ABI version of the loaded library (DNMF_ABI_VERSION). */
int dnmf_version(void);
/* Text of the last error raised on the calling thread ("" if none). */
const char *dnmf_last_error(void);
/* "file:hash;file:hash;..." of the sources this library was compiled from (dnmf_amd/build.py computes the hashes and
 * passes them at compile time; "" for a build made without it).  bench.py reports counter data from profiles/ only when
 * the hashes recorded with them equal these. */
const char *dnmf_build_stamp(void);

/* ---- packed footprints ---------------------------------------------------------------------
 * The Gram kernel reads footprints from a zero-padded copy with row length Kp = 16*ceil((K+1)/16)
 * floats (16-byte aligned rows; one spare column carries the frame so that A_t^T y falls out of the
 * same matrix product).  dnmf_padded_k returns Kp (0 if K < 1). */
int dnmf_padded_k(int K);
/* Apk (P,Kp) <- A (P,K), pad columns zeroed.  Replaces nothing in the reference (layout change). */
int dnmf_pack_footprints(const float *A, long P, int K, float *Apk, int Kp, dnmf_stream_t stream);

/* ---- K1: materialised warp ---------------------------------------------------------------------
 * ExponentialFP.forward up to A_t (Demix/dNMF.py:54-57): q = basis.beta_t, n = 2q/(S-1)-1, zero-padded
 * trilinear gather of all K channels (torch grid_sample, align_corners=True).
 *   times  (B) int32 frame indices into beta's last axis
 *   A_t    (B,K,X,Y,Z) or NULL;  grid (X,Y,Z,3,B) normalised coordinates or NULL */
int dnmf_warp_gather(const float *A, int X, int Y, int Z, int K, const float *beta, int T,
                     const int *times, int B, float *A_t, float *grid, dnmf_stream_t stream);

/* ---- reconstruction image ------------------------------------------------------------------------
 * S[b,p] = sum_k C[k,times[b]] * A[p,k]  (fp32 MFMA).  Because the trilinear gather is linear in the
 * footprints, A_tC of Demix/dNMF.py:58 equals the gather of this single image; K2 consumes it.
 *   Apk (P,Kp) packed footprints, P = X*Y*Z; C (K,T) row stride ldc;
 *   S (B, halo layout) row stride lds >= dnmf_halo_voxels(X,Y,Z) floats, border written as zeros */
int dnmf_recon_image(const float *Apk, int X, int Y, int Z, int K, int Kp, const float *C, long ldc, const int *times,
                     int B, float *S, long lds, dnmf_stream_t stream);

/* ---- K2: fused warp + reconstruction + loss + d loss / d beta --------------------------------------
 * One mini-batch of update_motion (Demix/dNMF.py:186-190): forward (dNMF.py:54-58), F.mse_loss
 * (dNMF.py:188, mean over B*P) and its autograd gradient w.r.t. beta[:,:,times]; also reg of
 * dNMF.py:60-61 (gradient-free in the reference).
 *   S       recon images in the halo layout (zero border), frame b at S + s_ids[b]*lds (s_ids NULL -> b)
 *   frames  video frames, frame b at frames + frame_ids[b]*ldf (frame_ids NULL -> b)
 *   gout    NULL, or (B,P) upstream gradient d L / d A_tC of an arbitrary loss: then grad receives
 *           its chain through the warp unscaled, `frames` may be NULL and loss / frame_loss are
 *           meaningless (this is the backward of the autograd node the Python surface exposes)
 *   recon   (B,P) A_tC or NULL
 *   grad    (10,3,T): columns times[b] are INCREMENTED by the gradient (autograd .grad semantics);
 *           times must not contain duplicates
 *   norm_frames  number of frames the mean of the loss runs over (0 -> B): lets one launch evaluate many
 *           mini-batches of norm_frames frames each
 *   loss    (1) sum over b of frame_loss;  frame_loss (B) per-frame sum of squares / (norm_frames*P) or NULL
 *   reg     (B) or NULL
 *   workspace: dnmf_warp_recon_grad_workspace(X,Y,Z,B) bytes */
size_t dnmf_warp_recon_grad_workspace(int X, int Y, int Z, int B);
int dnmf_warp_recon_grad(const float *S, long lds, const int *s_ids, const float *frames, long ldf,
                         const int *frame_ids, const float *gout, int X, int Y, int Z, const float *beta,
                         int T, const int *times, int B, int norm_frames, float *recon, float *grad, float *loss,
                         float *frame_loss, float *reg, void *workspace, size_t workspace_bytes,
                         dnmf_stream_t stream);

/* ---- K3: fused warp + per-frame Gram + rhs ---------------------------------------------------------
 * The two large contractions of update_temporal (Demix/dNMF.py:141-142) on the warped footprints of
 * spatial_pushforward (dNMF.py:69-87) without materialising A_t:
 *   G[b] = A_t^T A_t (K,K),  r[b] = A_t^T y_b (K),  A_t = warp of A by beta[:,:,times[b]].
 * fp32 MFMA (v_mfma_f32_16x16x4_f32), upper triangle only, symmetric fill.
 *   Apk      packed footprints; a_frame_stride 0 = one A for all frames, else floats between the
 *            per-frame packed A of consecutive b (static update_temporal on an explicit A_t)
 *   beta     NULL = no warp: every voxel takes its own footprint row with weight 1 (the static update_temporal of
 *            Demix/dNMF.py:139-149 on an explicit A_t; a trilinear re-sampling at the identity is not exact in fp32)
 *   times    (B) or NULL (-> 0..B-1); frames / ldf / frame_ids as in K2
 *   G (B,K,K), r (B,K); workspace: dnmf_warp_gram_rhs_workspace(P,K,B) bytes */
size_t dnmf_warp_gram_rhs_workspace(long P, int K, int B);
int dnmf_warp_gram_rhs(const float *Apk, int Kp, int K, long a_frame_stride, int X, int Y, int Z,
                       const float *beta, int T, const int *times, int B, const float *frames, long ldf,
                       const int *frame_ids, float *G, float *r, void *workspace, size_t workspace_bytes,
                       dnmf_stream_t stream);

/* K3b: the same contraction on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16): warped footprints and frame are
 * evaluated in fp32 as above, rounded to bf16 (nearest even) and accumulated in fp32.  Reduced precision that the
 * reference does not have (it contracts in float64); offered for BASELINE config 5.  Same arguments, workspace and
 * results layout as dnmf_warp_gram_rhs. */
int dnmf_warp_gram_rhs_bf16(const float *Apk, int Kp, int K, long a_frame_stride, int X, int Y, int Z,
                            const float *beta, int T, const int *times, int B, const float *frames, long ldf,
                            const int *frame_ids, float *G, float *r, void *workspace, size_t workspace_bytes,
                            dnmf_stream_t stream);

/* ---- K3s: the same contraction with exact-zero block skipping ---------------------------------------------
 * For footprints that are exactly zero over most of the volume (the reference's Gaussians underflow to 0 in
 * fp32 beyond ~30 px; multiplicative updates keep zeros).  Neurons are taken in the order `order` (K ints, sorted
 * channel -> neuron; the caller sorts them along a space-filling curve) and cut into blocks of 16; a product in
 * which either factor is an exact zero is not evaluated, which changes no sum.
 * dnmf_pack_footprints_sparse: Aps (P,Ks) <- A[:, order], Ks = dnmf_sparse_k(K) = 16*ceil(K/16), zero padded;
 *   row_mask[p] bit b = row p has a non-zero among channels 16b..16b+15.  K <= 128.
 * dnmf_warp_gram_rhs_sparse: arguments as dnmf_warp_gram_rhs; G, r come out in the ORIGINAL neuron order.
 *   counters: NULL, or 2 x uint64 that are INCREMENTED by the MFMA instructions issued and by the (block, k-step)
 *   gathers executed (the work that was not skipped; bench.py's roofline uses them). */
int dnmf_sparse_k(int K);
int dnmf_pack_footprints_sparse(const float *A, long P, int K, const int *order, float *Aps, int Ks,
                                unsigned char *row_mask, dnmf_stream_t stream);
size_t dnmf_warp_gram_rhs_sparse_workspace(long P, int K, int B);
int dnmf_warp_gram_rhs_sparse(const float *Aps, int Ks, int K, const int *order, const unsigned char *row_mask,
                              int X, int Y, int Z, const float *beta, int T, const int *times, int B,
                              const float *frames, long ldf, const int *frame_ids, float *G, float *r,
                              void *workspace, size_t workspace_bytes, unsigned long long *counters,
                              dnmf_stream_t stream);

/* The same with a local block table: a wave keeps accumulators only for the (at most four) blocks it is working
 * with and adds finished tiles into its slab, so four waves fit on a SIMD.  Same arguments and results. */
size_t dnmf_warp_gram_rhs_sparse_lt_workspace(long P, int K, int B);
int dnmf_warp_gram_rhs_sparse_lt(const float *Aps, int Ks, int K, const int *order, const unsigned char *row_mask,
                                 int X, int Y, int Z, const float *beta, int T, const int *times, int B,
                                 const float *frames, long ldf, const int *frame_ids, float *G, float *r,
                                 void *workspace, size_t workspace_bytes, unsigned long long *counters,
                                 dnmf_stream_t stream);

/* ---- K3n: the same contraction neuron by neuron, for compact footprints ---------------------------------------
 * When every footprint is non-zero only inside a small box (the reference's Gaussians: ~61 px wide), almost all of
 * A_t^T A_t is a sum of products with an exact zero.  This kernel evaluates, per tile of 256 voxels, only the neurons
 * whose box the tile's taps can reach, on the vector ALU; sums are those of dnmf_warp_gram_rhs up to the order of
 * fp32 additions and are deterministic.  K <= 256.
 * dnmf_pack_footprints_lists: At (K, halo layout) <- A (P,K) transposed, border zeroed; bbox (K,6) int32 =
 *   xlo,xhi,ylo,yhi,zlo,zhi of the non-zeros of each footprint; pair_slot (K,K) int32 and *nslot (one int32, DEVICE) =
 *   the static pattern of G: slots [0,K) hold r, the following ones the pairs (k,l) whose boxes can meet under one tap
 *   cell, the last one (nslot-1) collects everything else; axis_masks (dnmf_lists_axis_masks_bytes(X,Y,Z,K) bytes) =
 *   per axis and coordinate the set of neurons whose box starts at or before / ends at or after it, from which the
 *   neuron list of a tile is three pairs of lookups.  The caller reads *nslot back once to size the workspace.
 * dnmf_warp_gram_rhs_lists: other arguments and results as dnmf_warp_gram_rhs (G dense (B,K,K), r (B,K); both NULL:
 *   the slot tables are left at the start of the workspace for dnmf_mu_temporal_slots);
 *   nslot <= 3800 (DNMF_E_UNSUPPORTED beyond: the footprints overlap too much, use K3 / K3s);
 *   workspace: dnmf_warp_gram_rhs_lists_workspace(nslot,K,X,Y,Z,B) bytes (slot tables, then the tile lists);
 *   counters: NULL, or 2 x uint64 INCREMENTED by the (tile, neuron) evaluations and the (tile, pair) sums done.
 *   Long videos: the tiles with more than four neurons are evaluated by a second launch on a side stream the library
 *   keeps (forked from `stream` behind the lists, joined back to it before the call's last kernel): ordering on
 *   `stream` is as if everything ran there. */
size_t dnmf_lists_axis_masks_bytes(int X, int Y, int Z, int K);
int dnmf_pack_footprints_lists(const float *A, int X, int Y, int Z, int K, float *At, int *bbox, int *pair_slot,
                               int *nslot, void *axis_masks, dnmf_stream_t stream);
size_t dnmf_warp_gram_rhs_lists_workspace(int nslot, int K, int X, int Y, int Z, int B);
int dnmf_warp_gram_rhs_lists_chunks(int X, int Y, int Z, int B); /* chunk tables per frame the launch will write */
int dnmf_warp_gram_rhs_lists(const float *At, const int *bbox, const int *pair_slot, const void *axis_masks, int nslot,
                             int K, int X, int Y, int Z, const float *beta, int T, const int *times, int B,
                             const float *frames, long ldf, const int *frame_ids, float *G, float *r, void *workspace,
                             size_t workspace_bytes, unsigned long long *counters, dnmf_stream_t stream);

/* Reconstruction image from the K3n layout: S (halo layout) as dnmf_recon_image, summing per tile of 4 x 64 voxels only the neurons
 * whose box meets the tile (static lists); bound by writing S.  K <= 256.  At and S 16-byte aligned, lds a multiple of 4
 * floats (dnmf_halo_voxels is a multiple of 32). */
int dnmf_recon_image_lists(const float *At, const int *bbox, int K, int X, int Y, int Z, const float *C, long ldc,
                           const int *times, int B, float *S, long lds, dnmf_stream_t stream);
/* The same with skip_empty != 0: tiles no neuron's box meets are left alone instead of being zeroed -- for a caller that
 * keeps S between calls and knows those tiles hold zeros already (an earlier call with skip_empty == 0 and the same bbox on
 * the same rows of S wrote them). */
int dnmf_recon_image_lists_ex(const float *At, const int *bbox, int K, int X, int Y, int Z, const float *C, long ldc,
                              const int *times, int B, float *S, long lds, int skip_empty, dnmf_stream_t stream);

/* One group of mini-batches of the fused motion epoch with the reconstruction images kept in the last-level cache:
 * dnmf_recon_image_lists and dnmf_warp_recon_grad (its frames / frame_ids / times / norm_frames / grad / frame_loss / reg
 * arguments, no upstream gradient, A_tC not returned) alternate over pieces of `chunk` frames that share ONE buffer of
 * `chunk` images, so S_t = A.C_t never makes the round trip through HBM (Demix/dNMF.py:58 + 186-190 for B frames).
 * Same kernels and sums as the two calls on all B frames (K2's finish kernel runs once at the end).  norm_frames > 0 is
 * required; B may exceed 65535; workspace: dnmf_motion_grad_lists_workspace(X,Y,Z,chunk,B) bytes. */
size_t dnmf_motion_grad_lists_workspace(int X, int Y, int Z, int chunk, int B);
int dnmf_motion_grad_lists(const float *At, const int *bbox, int K, const float *C, long ldc, const float *frames, long ldf,
                           const int *frame_ids, int X, int Y, int Z, const float *beta, int T, const int *times, int B,
                           int norm_frames, float *grad, float *frame_loss, float *reg, int chunk, void *workspace,
                           size_t workspace_bytes, dnmf_stream_t stream);

/* ---- K4: multiplicative update of the traces --------------------------------------------------------
 * C <- C * (r + gamma*nbr) / (G C + 2 gamma C + 1e-32)  (Demix/dNMF.py:143-148, looped at dNMF.py:172-173)
 * on the hoisted G, r.  Arithmetic in fp64 like the reference's numpy code.
 *   G (T,K,K) symmetric, r (T,K)
 *
 * dnmf_mu_temporal: `iters` rounds without the neighbour term (gamma None or 0: frames independent, the
 *   whole loop runs in registers); C (K,T) fp32 row stride ldc, in/out, rounded to fp32 once at the end
 *   as dNMF.py:177 does.
 * dnmf_mu_temporal_step: ONE round with the neighbour term on an fp64 state, Cin -> Cout (both (K,T),
 *   row stride ldc, distinct buffers).  Neighbours of the first / last frame are replicated
 *   (dNMF.py:145) unless c_left / c_right (K doubles: the adjacent frame owned by the neighbouring
 *   T-shard) are given. */
int dnmf_mu_temporal(const float *G, const float *r, float *C, long ldc, int K, int T, int iters,
                     dnmf_stream_t stream);
/* dnmf_mu_temporal when G has a known pattern (the static pattern of K3n): nbr (K,NN) int32 lists for every row the
 * columns that can be non-zero in ascending order, padded with columns that cannot; NN in {8,16,32}.  Bit-identical
 * results (the skipped terms are exact zeros), NN instead of K terms per row and round. */
int dnmf_mu_temporal_nbr(const float *G, const float *r, float *C, long ldc, int K, int T, int iters, const int *nbr,
                         int NN, dnmf_stream_t stream);
/* The same straight from the slot tables of K3n: call dnmf_warp_gram_rhs_lists with G = r = NULL (the tables then stay
 * in its workspace as (T, nchunks, nslot) floats, nchunks = dnmf_warp_gram_rhs_lists_chunks(X,Y,Z,T)) and hand the
 * workspace in as `slab`.  Bit-identical to finishing into a dense G first; saves writing and re-reading (T,K,K). */
int dnmf_mu_temporal_slots(const float *slab, int nchunks, int nslot, const int *pair_slot, float *C, long ldc, int K,
                           int T, int iters, const int *nbr, int NN, dnmf_stream_t stream);
int dnmf_mu_temporal_step(const float *G, const float *r, const double *Cin, double *Cout, long ldc, int K,
                          int T, double gamma, const double *c_left, const double *c_right,
                          dnmf_stream_t stream);

/* ---- K5 / K6: multiplicative update of the footprints -----------------------------------------------
 * DeformableNMF.update_spatial (Demix/dNMF.py:151-160).
 * dnmf_spatial_accum: A1[p,k] = sum_t Y[t,p] C[k,t] (fp32 MFMA, dNMF.py:154) and Cs = C C^T (dNMF.py:153) over
 *   the T frames given: frame t at Y + frame_ids[t]*ldy (NULL -> t), trace column times[t] (NULL -> t);
 *   Ct: NULL, or the same traces frame-major, Ct[c*ldct + k] = C[k*ldc + c] for every column c the call touches: the
 *   matrix operands are then read in 64-byte runs instead of 16 rows x 16 bytes per instruction (the kernel's limit);
 *   accumulate != 0 adds to A1 / Cs instead of overwriting (frame chunks).  With the T axis sharded the caller
 *   sums A1 (P,K) and Cs (K,K) over ranks (RCCL all-reduce) before dnmf_mu_spatial.
 * dnmf_mu_spatial: A <- A * A1 / (A Cs + gamma D + 1e-32) in place (dNMF.py:155-159); D (P,K) or NULL. */
int dnmf_spatial_accum(const float *Y, long ldy, const int *frame_ids, const float *C, long ldc, const float *Ct,
                       long ldct, const int *times, int T, long P, int K, float *A1, float *Cs, int accumulate,
                       dnmf_stream_t stream);
int dnmf_mu_spatial(float *A, const float *A1, const float *Cs, const float *D, double gamma, long P, int K,
                    dnmf_stream_t stream);

/* ---- C1: frame-summed accumulators across the GPUs of a node -----------------------------------------
 * With the T axis sharded over ranks the two sums over frames of update_spatial (A1 = Y_i C^T, dNMF.py:154, and
 * Cs = C C^T, dNMF.py:153) are completed by ONE RCCL all-reduce each per update; nothing else on the path
 * communicates.  The communicator is the only state the library ever holds, behind an opaque handle the caller owns.
 * RCCL is bound at run time (dlopen of the librccl already in the process -- torch's -- else the system one), so
 * the library loads on hosts without it and these calls then return DNMF_E_UNSUPPORTED.
 *   dnmf_comm_unique_id: rank 0 fills `id` (HOST, DNMF_COMM_ID_BYTES) and hands it to the other ranks out of band
 *     (the host mirror broadcasts it over the caller's torch.distributed group);
 *   dnmf_comm_init: collective over the nranks processes, each with its own current HIP device;
 *   dnmf_allreduce_sum_f32: buf (count floats, device) <- sum over ranks, in place, enqueued on `stream`;
 *   dnmf_comm_destroy: releases the communicator (NULL is accepted).
 * Positive return values of these four are ncclResult_t codes (text in dnmf_last_error). */
#define DNMF_COMM_ID_BYTES 128
typedef void *dnmf_comm_t;
int dnmf_comm_unique_id(void *id_host);
int dnmf_comm_init(dnmf_comm_t *comm, const void *id_host, int nranks, int rank);
int dnmf_allreduce_sum_f32(dnmf_comm_t comm, float *buf, size_t count, dnmf_stream_t stream);
int dnmf_comm_destroy(dnmf_comm_t comm);

/* ---- K7: registered video ---------------------------------------------------------------------------
 * ExponentialFP.image_iwarp over the frames of spatial_pushforward (Demix/dNMF.py:81-83, 89-91, 95-103): every
 * lattice point takes the value of the voxel whose warped position ((n+1)/2 * sz, the reference's scaling
 * there) is nearest (float64 distances on the fp32 positions, ties to the lowest voxel index).  The search runs in
 * a window around the back-mapped lattice point whose radius follows from a lower bound of the warp's stretch, so
 * it returns what an exhaustive search returns; lattice points whose window would be too large (a folding or
 * violent warp, points far outside the warped image) are searched exhaustively.
 *   out (B,P) row stride ldo;  workspace: dnmf_image_iwarp_workspace(X,Y,Z,B) bytes (one flag per lattice point, sixteen constants of the
 *   frame's warp and one counter of marked points per frame), 8-byte aligned;
 *   exhaustive != 0: every lattice point by the exhaustive search (the checker of the window search);
 *   fallback_count: NULL, or one uint64 INCREMENTED by the lattice points that took the exhaustive search. */
size_t dnmf_image_iwarp_workspace(int X, int Y, int Z, int B);
int dnmf_image_iwarp(const float *frames, long ldf, const int *frame_ids, int X, int Y, int Z, const float *beta,
                     int T, const int *times, int B, float *out, long ldo, void *workspace, size_t workspace_bytes,
                     int exhaustive, unsigned long long *fallback_count, dnmf_stream_t stream);

/* ---- Adam on beta for one epoch of mini-batches ------------------------------------------------------
 * update_motion steps the caller's torch.optim.Adam once per mini-batch on the whole (10,3,T) tensor
 * (Demix/dNMF.py:186-191; optimiser built at demo.py:42): columns outside the mini-batch get a zero gradient
 * but still move once they have moment history.  Columns are independent, so an epoch of `nsteps` steps in
 * which frame t belongs to mini-batch frame_step[t] (0-based, <0 = none) is evaluated per column:
 *   phase 0: the frame_step[t] zero-gradient steps before its mini-batch (then run K2 on all frames),
 *   phase 1: the step with grad[:, :, t] and the zero-gradient steps after it.
 * beta, exp_avg, exp_avg_sq (10,3,T) are the optimiser's own tensors, updated in place; step0 = steps taken
 * before this epoch.  torch.optim.Adam with amsgrad off, weight_decay 0, maximize off: the step with a gradient is
 * torch's fp32 arithmetic literally; a run of zero-gradient steps is evaluated in closed form (m b1^i, v b2^i, the
 * increments of p summed in double, ending when m underflows), within 1e-6 of the displacement of the step-by-step
 * fp32 evaluation.
 *   order  NULL, or (T) a permutation of the frames sorted by frame_step: thread r of the launch takes frame
 *          order[r], so the lanes of a wave walk through the same window of steps (speed only).
 *   workspace: dnmf_adam_epoch_workspace(nsteps) bytes (the step-dependent scalars of the epoch, rebuilt by each call) */
size_t dnmf_adam_epoch_workspace(int nsteps);
int dnmf_adam_epoch(float *beta, const float *grad, float *exp_avg, float *exp_avg_sq, int T, long step0,
                    const int *frame_step, const int *order, int nsteps, double lr, double beta1, double beta2,
                    double eps, int phase, void *workspace, size_t workspace_bytes, dnmf_stream_t stream);

/* ---- synthetic input: the render loop of the simulator ------------------------------------------------
 * WUtils/Simulator.py:66-73 (generate_video) with simulate_cell (:197-212): frame t0+t receives, neuron by
 * neuron in index order, fp32(traces[k,t] * exp(-|v - positions[k,:,t]|^2 / (2 shape_std))) added in fp32.
 *   positions (K,3,T_total) fp32, traces (K,T_total) fp64, out (T,P) fp32 row stride ldo (frames t0..t0+T-1)
 *   amp_max >= max(traces): bounds the window outside which a term is exactly 0 after the fp32 cast */
int dnmf_render_frames(const float *positions, const double *traces, int K, int T_total, int t0, int T, int X,
                       int Y, int Z, double shape_std, double amp_max, float *out, long ldo,
                       dnmf_stream_t stream);

/* ---- K5 / K6, list form (compact footprints) ---------------------------------------------------------------------
 * Reference: DeformableNMF.update_spatial, Demix/dNMF.py:151-160 -- A1 = Y_i C^T is only ever used multiplied by A, which is
 * an exact zero outside footprint k's non-zero box and stays zero under the update.  Tiles of 4 x-rows x 64 positions of the
 * (y,z) plane list the neurons whose box (bbox of dnmf_pack_footprints_lists) meets them; the sums exist only for (tile,
 * listed neuron) pairs, in a compact buffer A1c of `total` floats (1 KiB per pair) -- also what the ranks all-reduce.
 * tables: int[2 ntiles + 2 + 32 ntiles], ntiles = dnmf_spatial_lists_tiles: list lengths, entry offsets (tables[2 ntiles] =
 * total, -1 when a tile lists more than 32 neurons: use the dense kernels), an overflow counter, the lists.
 * dnmf_spatial_accum_lists: A1c and Cs = C C^T (float64 sums) over the T frames given (rows frame_ids[b] or b of Y, trace
 * columns times[b] or b).  dnmf_mu_spatial_lists: A1c in = the (all-reduced) sums, out = the new footprint values; A (P,K) is
 * updated in place at the listed entries (every other entry is and stays zero); At = the neuron-major halo copy of A. */
long dnmf_spatial_lists_tiles(int X, int Y, int Z);
int dnmf_spatial_lists_setup(const int *bbox, int K, int X, int Y, int Z, int *tables, dnmf_stream_t stream);
size_t dnmf_spatial_accum_lists_workspace(int X, int Y, int Z, long total, int T);
int dnmf_spatial_accum_lists(const float *Y, long ldy, const int *frame_ids, const float *C, long ldc, const int *times, int T, int X,
                             int Yd, int Z, int K, const int *tables, long total, float *A1c, float *Cs, void *workspace,
                             size_t workspace_bytes, dnmf_stream_t stream);
int dnmf_mu_spatial_lists(float *A, const float *At, float *A1c, const float *Cs, const float *D, double gamma, int X, int Y, int Z,
                          int K, const int *tables, dnmf_stream_t stream);

/* ---- K8: position initialiser ----------------------------------------------------------------------
 * Reference: Demix/MotionCorrect.py -- MotionCorrect.motion_correct_pwrigid :260-328 -> tile_and_correct_3d :1518-1608
 * (per frame: rigid shift, then one shift per patch inside rigid +- max_deviation_rigid; register_translation_3d :648-797 with
 * the matrix-multiply upsampled DFT :498-614) and MotionCorrect.apply_shifts_points :351-371.  The module is an orphan of
 * the reference tree, cannot be imported in the build container and has no fixtures: parity is pinned only against the
 * restatement oracle/motion_oracle.py ("parity unpinned").
 *
 * dnmf_register_patches_grid: the patch grid of sliding_window_3d :1190-1221 (windows of strides + overlaps every `strides`,
 * the last one flush with the end).  Returns the number of patches NP (0: the windows do not fit), dims[3] = patches per
 * axis, starts (NP,3) = first voxel of each patch in the reference's order (x outermost); dims / starts may be NULL (host
 * pointers).
 * dnmf_register_patches: frames (>= B rows of ldf floats, voxel p = (x Y + y) Z + z; row frame_ids[b] or b), tmpl (X Y Z)
 * -> rigid_shifts (B,3) as register_translation_3d returns them, patch_shifts (B,NP,3) with the signs of
 * tile_and_correct_3d's total_shifts (-x, -y, +z), i.e. what the class stores in x/y/z_shifts_els.  strides, overlaps,
 * max_shifts: 3 host ints each.  All device buffers fp32. */
int dnmf_register_patches_grid(int X, int Y, int Z, const int *strides, const int *overlaps, int *dims, int *starts);
size_t dnmf_register_patches_workspace(int X, int Y, int Z, const int *strides, const int *overlaps, int B);
int dnmf_register_patches(const float *frames, long ldf, const int *frame_ids, int B, const float *tmpl, int X, int Y, int Z,
                          const int *strides, const int *overlaps, const int *max_shifts, int max_deviation_rigid,
                          int upsample_factor, float add_to_movie, float *rigid_shifts, float *patch_shifts, void *workspace,
                          size_t workspace_bytes, dnmf_stream_t stream);
/* Rigid correction of the frames against a template (tile_and_correct_3d :1518-1574 with max_deviation_rigid == 0, the
 * pass motion_correct_batch_rigid :1770-1877 makes to build the template a piecewise-rigid pass starts from):
 * rigid_shifts (B,3) as register_translation_3d returns them (the reference's shifts_rig holds their negatives, :1574);
 * corrected (NULL or B rows of ldc floats): every frame moved by its shift through the phases of its spectrum
 * (apply_shifts_dft :1028-1157, 3-D branch), minus add_to_movie; border_nan: what goes where the shift brought in voxels
 * from the other side -- 0 nothing (False of the reference), 1 NaN (True), 2 the frame's smallest value ('min'), 3 the
 * nearest row / column / slice inside ('copy'); tsum / tcount (both NULL, or P
 * floats / ints): += the finite corrected values and their number per voxel -- the nanmean of
 * tile_and_correct_wrapper :2057 is tsum / tcount. */
size_t dnmf_rigid_correct_workspace(int X, int Y, int Z, int B);
int dnmf_rigid_correct(const float *frames, long ldf, const int *frame_ids, int B, const float *tmpl, int X, int Y, int Z,
                       const int *max_shifts, int upsample_factor, float add_to_movie, int border_nan, float *rigid_shifts,
                       float *corrected, long ldc, float *tsum, int *tcount, void *workspace, size_t workspace_bytes,
                       dnmf_stream_t stream);
/* apply_shifts_points :351-371: points (K,3), patch_shifts (T,NP,3) as above, centers (NP,3) = patch start + strides / 2 ->
 * out (K,3,T): out[k,0/1,t] = p - (s[t] - s[0]), out[k,2,t] = p + (s[t] - s[0]) with s the shifts of the patch whose
 * centre is nearest to point k. */
int dnmf_apply_shifts_points(const float *points, int K, const float *patch_shifts, int T, int NP, const float *centers, float *out,
                             dnmf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DNMF_HIP_H */
