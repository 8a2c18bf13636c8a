/* aewavenet.h — C ABI of the MI355X-native WaveNet-autoencoder training hot path.
 *
 * The reference (hrbigelow/ae-wavenet) is pure Python on torch ATen; it has no FFI.  Its
 * drop-in boundary is the nn.Module surface (autoencoder_model.py:206-259,
 * mfcc_inverter.py:89-108, driven by chassis.py:151-171).  This header is the C ABI the
 * build adds *underneath* that surface: plain pointers, sizes and POD descriptors, no torch
 * types.  Each entry point / op lists the reference ATen call sites it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative AEW_E_* on argument errors, or a
 *     positive hipError_t; nothing throws across the ABI
 *   - the library never allocates or frees device memory; the caller owns all buffers
 *   - all work is enqueued on the hipStream_t passed in (stream-ordered, no host sync)
 *   - activations are CHANNELS-LAST: tensor[b][t][c], c fastest.  bf16 is stored as uint16
 *   - "row" means a time index t; a buffer is addressed as
 *         ptr + b*batch_stride + row*row_pitch + c            (element units)
 *   - descriptors carry everything an op computes WITH; kernel-shape choices live in a tuning record (aew_tuning_t): the
 *     aew_set_* switches edit the process-wide one, aew_run_plan_tuned takes a caller's own
 *     (kernel shape / schedule choices for A-B measurements and bisecting; results are the same under every setting
 *     unless a switch says otherwise).  They default to the production choice, are not thread-safe, and take effect
 *     for launches / graph captures made afterwards; a caller that never touches them gets a stateless library
 *
 * The hot path is a static LAUNCH PLAN: an array of aew_op_t records executed in order by
 * aew_run_plan().  Almost every record is one of two GEMM forms over *row-affine segments*:
 *
 *   NT ("forward/dgrad"):  C[b][m][n]  = sum_s sum_k A_s[b][m*step_s + off_s][k] * W[n][K_s + k]
 *   TN ("wgrad")        :  dW[b][n][K_s + k] = sum_m G[b][m*gstep + goff][n] * A_s[b][m*step_s + off_s][k]
 *
 * Rows outside a segment's [row_lo,row_hi) read as zero.  With this one form:
 *   dilated causal conv (wavenet.py:100-101)      = 3 segments  x[t], x[t+dil], cond[t+lead]
 *   strided encoder conv (wave_encoder.py:39)      = f segments  x[t*s + tap]
 *   ConvTranspose1d upsampler (wavenet.py:154)     = f/s segments per output phase, x[q - j]
 *   every dgrad                                    = the transposed segment table
 *   every wgrad                                    = TN over the same segments
 */
#ifndef AEWAVENET_H
#define AEWAVENET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AEW_ABI_VERSION 20
#define AEW_MAX_SEGS 32

/* error codes (negative; positive values are hipError_t) */
#define AEW_E_ARG      (-1)   /* malformed descriptor */
#define AEW_E_UNSUP    (-2)   /* unsupported op / dtype / flag combination */
#define AEW_E_ALIGN    (-3)   /* pointer or pitch violates the documented alignment */

/* element types */
#define AEW_BF16 0
#define AEW_F32  1

/* ---------------------------------------------------------------------------------------
 * A-operand segment: rows m*row_step + row_off of a channels-last buffer, k_len channels.
 * k_len must be a multiple of the kernel's K tile (64 for bf16, 32 for f32); the buffer
 * must hold k_len (zero-padded) channels per row.  16-byte alignment of ptr and of
 * row_pitch*sizeof(elem) is required.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const void* ptr;
    int64_t batch_stride;
    int32_t row_pitch;
    int32_t row_step;
    int32_t row_off;
    int32_t row_lo, row_hi;      /* valid source rows [lo,hi); others read as zero */
    int32_t k_len;
} aew_seg_t;

/* generic channels-last view used by epilogues */
typedef struct {
    void* ptr;
    int64_t batch_stride;
    int32_t row_pitch;
    int32_t row_step;            /* view row = m*row_step + row_off */
    int32_t row_off;
    int32_t row_lo, row_hi;      /* rows of the *view* that exist: stores outside are dropped,
                                    loads outside read as zero */
    int32_t dtype;               /* AEW_BF16 | AEW_F32 */
} aew_view_t;

/* NT epilogues */
#define AEW_EPI_STORE     0   /* flag-driven store, see AEW_EF_* */
#define AEW_EPI_GATED     1   /* a=tanh(f+bf), s=sigmoid(g+bg); N is (16 filt | 16 gate)-interleaved.
                                 out0 = z = a*s, out1 = dz/df = s(1-a^2), out2 = dz/dg = a s(1-s)
                                 (bf16)                                   wavenet.py:100-102 */
#define AEW_EPI_RES_SKIP  2   /* n<n_split: out0[m]=acc+aux0[m] (residual, wavenet.py:108-109)
                                 n>=n_split: out1[m][n-n_split] (+)= acc (skip sum, :103,:357);
                                 with AEW_EF_OUT2_RELU also out2=relu(out1) bf16 (:359)          */
#define AEW_EPI_DFG       3   /* acc=dz; aux0=dz/df, aux1=dz/dg (from GATED); out0 = (dfilt|dgate)-
                                 interleaved bf16: dz*aux0, dz*aux1    (backward of wavenet.py:102) */

/* AEW_EPI_STORE flags */
#define AEW_EF_BIAS        (1u << 0)  /* val += bias[b*bias_bs + n]                           */
#define AEW_EF_RELU        (1u << 1)  /* val = max(val,0)          (after bias)               */
#define AEW_EF_OUT1_PRE    (1u << 2)  /* out1 = val  (before the aux add; encoder relu out)    */
#define AEW_EF_ADD_AUX0    (1u << 3)  /* val += aux0[view row of m][n]  (zero outside range)   */
#define AEW_EF_MUL_POS1    (1u << 4)  /* val *= (aux1[m][n] > 0)                               */
#define AEW_EF_OUT1_POS1   (1u << 5)  /* out1 = val * (aux1[m][n] > 0)  (out0 keeps val)       */
#define AEW_EF_ACCUM       (1u << 6)  /* RES_SKIP: out1 += acc instead of =                    */
#define AEW_EF_OUT2_RELU   (1u << 7)  /* RES_SKIP: out2 = relu(new out1) as bf16              */
#define AEW_EF_RELU_POST   (1u << 9)  /* val = max(val,0) AFTER the aux0 add (sum of partial GEMMs, then relu) */
#define AEW_EF_OUT2_COPY   (1u << 10) /* out2 = the value stored to out0, in out2's dtype (bf16 copy of an fp32
                                         activation for the bf16 backward GEMMs)                */
#define AEW_EF_COUNT_ZERO  (1u << 8)  /* atomically add #(out0==0) into counter (enc_az metric,
                                         wave_encoder.py:46)                                   */

typedef struct {
    int32_t dtype;               /* operand type AEW_BF16 | AEW_F32 */
    int32_t impl;                /* 0 = tiled MFMA kernel, 1 = scalar check kernel, 2 = full-N MFMA kernel with
                                    loader / consumer waves (aew_fn.hip; falls back to 0 when the shape is not
                                    covered).  Same results bit for bit                            */
    int32_t M;                   /* output rows per batch                                     */
    int32_t N;                   /* output columns to store: multiple of 8 (bf16) / 4 (f32)   */
    int32_t N_pad;               /* rows of W (multiple of 128 for bf16, 64 for f32)          */
    int32_t batch;
    int32_t n_segs;
    int32_t K_total;             /* sum of k_len = row length of W                            */
    aew_seg_t seg[AEW_MAX_SEGS];
    const void* W;               /* packed [N_pad][K_total], same dtype as the operands       */
    int32_t epi;
    uint32_t flags;
    aew_view_t out0, out1, out2; /* views are indexed with the GEMM row m                     */
    aew_view_t aux0, aux1;
    const float* bias;           /* fp32 [*, N_pad]                                           */
    int64_t bias_bs;             /* per-batch stride of bias (0 = shared)                     */
    int32_t n_split;             /* RES_SKIP column boundary (multiple of the N tile)         */
    int32_t reserved;
    unsigned long long* counter; /* AEW_EF_COUNT_ZERO target                                  */
    /* Fused gated layer (AEW_EPI_GATED only; wavenet.py:100-109).  With W2 != NULL the op also computes the
     * residual 1x1 of the layer from the z tile it has just produced:
     *     out3[b][m][n2] = sum_k z[b][m][k] * W2[n2][k] + aux0[b][m][n2]        n2 < N2, k < N_pad / 2
     * (z = the bf16 values written to out0, so the result equals GATED followed by a STORE | ADD_AUX0 GEMM over
     * out0; impl 0 / 1 execute it exactly that way, impl 2 keeps the z tile in LDS).                        */
    const void* W2;              /* packed [N2_pad][N_pad / 2] bf16                                          */
    int32_t N2, N2_pad;          /* N2 multiple of 8, N2_pad multiple of 128                                 */
    aew_view_t out3;
    /* Split-K of the exact fp32 kernel (ABI 19; AEW_F32, impl 0, AEW_EPI_STORE only; wave_encoder.py:39, vqema_bn.py:131).
     * k_split = S in {2, 4}: the concatenated K axis is cut into S contiguous ranges of K_total / S channels (a multiple of
     * 32), each a k-ascending fmaf chain of its own on its own workgroup; the partial sums meet in the FIXED order
     *     S = 2: p0 + p1          S = 4: (p0 + p1) + (p2 + p3)         (plain fp32 adds, then bias / relu / ...)
     * whichever workgroup arrives last does the combine and the epilogue - the order of arrival does not enter the result.
     * This IS the canonical summation order of the op (oracle/exact_chain.c: aewo_conv_cl ksplit), chosen so that the
     * 180-block launches of the encoder (232-560 rows in all) become 720 blocks with a quarter of the serial K loop each.
     * ksplit_ws: device fp32 [S][rows_pad][N_pad], rows_pad = M * batch rounded up to 32; ksplit_tickets: device uint32
     * [tiles of 16 rows x 16 channels rounded up], ZERO before the first launch (the combine resets them).  0 / 1 = off.
     *
     * AEW_BF16, impl 0: a HINT for launches of very few workgroups - the upsampler / encoder data gradients,
     * 16-112 blocks of 64 x 64 on 256 CUs, each a lone block's LDS fill latency long (wavenet.py:275, wave_encoder.py:39
     * backward).  k_split = S in 2..8 with K_total a multiple of 64 * S: the launch becomes S copies of its tile grid, copy s
     * contracting K tiles [s, s + 1) * K_total / 64 / S; the partial accumulators meet in ascending order of s (a fixed
     * order: results are deterministic, but NOT those of the unsplit launch - bf16 operands, fp32 accumulation, the same
     * tolerance class) in the workgroup that arrives last.  Honoured only where the launcher takes its 64 x 64 shape under
     * the tuning record in force (else the whole K axis is contracted by one workgroup, the slabs stay unused);
     * aew_gemm_nt_small_split() states the rule and the sizes: ksplit_ws = ws_bytes bytes (16-byte aligned),
     * ksplit_tickets = n_tickets uint32, zero before the first launch.                                                   */
    int32_t k_split;
    int32_t pad2_;
    float* ksplit_ws;
    uint32_t* ksplit_tickets;
} aew_gemm_nt_t;

typedef struct {
    int32_t dtype;
    int32_t impl;
    int32_t Mc;                  /* contraction rows per batch                                */
    int32_t batch;
    int32_t N;                   /* real columns of G                                         */
    int32_t N_pad;               /* multiple of the tile (128 bf16 / 64 f32); G has N_pad cols */
    aew_seg_t g;                 /* G operand; k_len ignored                                  */
    int32_t n_segs;
    int32_t K_total;
    aew_seg_t seg[AEW_MAX_SEGS];
    float* out;                  /* fp32 [batch][N_pad][K_total] per-batch partial sums       */
    int64_t out_batch_stride;
    /* Grouped form only (AEW_OP_GEMM_TN_GROUP: the contraction runs over the batch elements in order inside one
     * block).  With snap_out != NULL the running column sum_{b' <= b} sum_m G[b'][m][n] * A[b'][m][snap_k] is
     * written after every batch element b:  snap_out[b * snap_bs + n].  With a constant-one channel at snap_k
     * (the pad channel R of x, aew_base_gather_t.ones_channel) that is the running column sum of G - the per-batch
     * bias / speaker-embedding gradients come out of the weight-gradient GEMM for free (aew_spk_bwd_t.colsum_running). */
    float* snap_out;
    int64_t snap_bs;
    int32_t snap_k;              /* column of the concatenated K axis; -1 (with snap_out): no A column - the running
                                    column sums of G themselves, from an all-ones MFMA operand in the blocks of the
                                    first k tile (layers without a spare pad channel for the ones: R a multiple of 128) */
    int32_t pad_;
    /* Grouped form only.  colsum_out != NULL: colsum_out[n] = sum_b sum_m G[b][m][n] for n < N - the bias gradient of
     * the layer whose weight gradient this is (its G operand is the gradient of the layer's pre-activation) - written
     * by the blocks of the first k tile from an extra all-ones MFMA operand: no separate column-sum launch, one
     * summation order (deterministic, no atomics).                                                                  */
    float* colsum_out;
    /* Grouped form only.  grp_splits > 0: the contraction is cut like a stand-alone TN op's - one block per (output
     * tile, batch element, chunk of grp_rows rows), grp_splits chunks per batch element, partial sums to slab
     * b * grp_splits + chunk of `out` (out_batch_stride apart; the unpack sums them).  For matrices with few output
     * tiles and a long contraction (the upsampler weights), where one block per tile would be a lone latency-bound
     * chain.  snap_out / colsum_out need grp_splits = 0 (one block sees every row).                              */
    int32_t grp_splits, grp_rows;
} aew_gemm_tn_t;

/* ---------------------------------------------------------------------------------------
 * Grouped weight gradients: the output tiles of SEVERAL TN descriptors as one launch, each tile contracting over
 * the WHOLE time axis and all batch elements (k ascending, b ascending) in one block.  One fp32 result per
 * descriptor (desc.out[N_pad][K_total], no split-K slabs to write and sum), and enough tiles to fill the chip
 * without splitting (20 layers x (4 x 7 + 3 x 2) tiles in the decoder).  Descriptors and the tile map live in
 * DEVICE memory (built once per plan).  tile_map[p] = desc << 22 | chunk << 12 | tile (tile = nt * (K_total / 128) + kt;
 * chunk = 0 unless the descriptor is split, see aew_gemm_tn_t.grp_splits), or -1: block p idles.  Workgroup p runs on XCD p % 8, so the builder places tiles that share operands on one XCD.
 * bf16 only.  Same products as aew_gemm_tn_t ops; the summation order differs (one chain instead of slabs).
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const aew_gemm_tn_t* descs;  /* device                                                      */
    const int32_t* tile_map;     /* device                                                      */
    int32_t n_descs, n_blocks;
    int32_t tile;                /* 128: 128 x 128 output tiles, 4 waves, three blocks per CU
                                    256: 256 x 256 tiles (half the operand bytes staged per FLOP), 8 waves, one block
                                         per CU; tile = nt * ((K_total / 128 + 1) / 2) + kt in 256-column units, halves
                                         beyond N_pad / K_total are skipped
                                    384 (ABI 20): EIGHT waves of 64 x 64 on a 128 (k) x 256 (n) tile - or 256 (k) x 128 (n)
                                         where K_total is a multiple of 256 and N_pad is not - two blocks per CU on a
                                         3 x 24 KiB ring (the NT kernels' shape: 1.5 x the MFMAs per staged byte of the
                                         128-tile, four waves per SIMD); tile = nt * nkt + kt in that grid; no split
                                         descriptors (grp_splits = 0), no cursor; results bit-identical to tile = 128 */
    int32_t cursor_stride;       /* words per descriptor in `cursors` (>= 64)                                          */
    /* Optional row cursor (ABI 18; tile = 128 only; non-NULL = pace this launch): device
     * [n_descs][cursor_stride] uint32, ZERO before every launch (epoch / slack: aew_set_tn_cursor); word i of a descriptor's row = the epoch tile i is
     * about to issue.  The output tiles of one descriptor share their operand columns; the cursor keeps them within
     * `slack` epochs (of `epoch` 32-row stages) of each other as they walk down the rows, so that what one tile
     * fetched is still in its XCD's L2 when its siblings read it.  Performance only: a tile whose wait times out (a
     * sibling not resident, or on another XCD) runs free from there on; descriptors with > 64 tiles are not paced.
     * NULL = off.                                                                                                    */
    uint32_t* cursors;
} aew_gemm_tn_group_t;

/* ---------------------------------------------------------------------------------------
 * table-driven strided copy / convert / reduce (weight pack, gradient unpack, NCL<->NLC)
 *   dst[sum_d i_d*ds_d] = cvt( sum_{r<red_n} src[sum_d i_d*ss_d + r*red_stride] )
 * Tiled form (tr_a > 0; red_n = 1, no accumulate, f32 source): dims[3] is the dim the SOURCE is (nearly) contiguous in,
 * dims[2] the one the DESTINATION is contiguous in (ds[2] = 1); a block moves one tr_a x tr_b tile of that plane through
 * LDS, reading along dims[3] and writing along dims[2] (transposing weight packs: both sides coalesced).  Same result
 * as the element-wise forms.  Blocks per record: dims[0] * dims[1] * ceil(dims[2] / tr_b) * ceil(dims[3] / tr_a).
 * Interleave forms (tr_a = -k, k = 2..4 taps of a conv weight; f32 source, red_n = 1, no accumulate, 16-byte aligned):
 *   tr_b = 1  [..][k][c] -> [..][c][k]:  dims[2] = k, dims[3] = c (ss[3] = 1, ds[2] = 1, ds[3] = k), f32 destination
 *   tr_b = 2  [..][c][k] -> [..][k][c]:  dims[2] = c, dims[3] = k (ss[3] = 1, ss[2] = k, ds[2] = 1), f32 or bf16
 * a thread permutes W = 4 (8 for bf16) channels of all taps in registers; blocks: ceil(dims[0] * dims[1] * c / W / 256).
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const void* src;
    void* dst;
    int32_t dims[4];
    int64_t ss[4], ds[4];
    int32_t src_dtype, dst_dtype;
    int32_t red_n;
    int32_t accumulate;          /* dst += (f32 dst only) */
    int64_t red_stride;
    float scale;
    int32_t first_block;         /* filled by the host: first block of this record (1024 elements per block, or one
                                    tile in the tiled form)                                                       */
    int32_t tr_a, tr_b;          /* tr_a > 0: tiled form, tile extents along dims[3] / dims[2] (tr_a * tr_b <= 4096,
                                    tr_b * (tr_a | 1) <= 4160); tr_a < 0: interleave forms; 0 = element-wise      */
} aew_copy_rec_t;

typedef struct {
    const aew_copy_rec_t* recs;  /* DEVICE array                                              */
    const int32_t* block_rec;    /* DEVICE array: record index of every block                 */
    int32_t n_blocks;
    int32_t n_recs;
} aew_copy_table_t;

/* ---------------------------------------------------------------------------------------
 * small ops
 * ------------------------------------------------------------------------------------- */
typedef struct {                 /* nearest code  (vqema_bn.py:135-142, vq_bn.py:39-41)        */
    const float* ze;             /* [Q][d_pitch] fp32, Q = B*N_e                              */
    const float* emb;            /* [K][d]                                                    */
    int32_t Q, K, d, d_pitch;
    int32_t metric;              /* 0 scaled_l2, 1 sq_l2                                      */
    int64_t* ind;                /* [Q]                                                       */
    float* dist;                 /* [Q]                                                       */
    float* zq;                   /* [Q][d_pitch] (pad channels zeroed)                        */
    void* scratch; int32_t n_split;   /* optional: 8*Q*n_split bytes.  The K codes are then scanned by n_split
                                         blocks per query (partial minima into scratch) and a second tiny kernel
                                         combines them in ascending split order: same result bit for bit, 4x
                                         shorter critical path at Q = 232, K = 4096                           */
} aew_vq_nearest_t;

typedef struct {                 /* z_sum/n_sum, deterministic order (vqema_bn.py:172-188)     */
    const float* ze; const int64_t* ind;
    int32_t Q, K, d, d_pitch;
    float* z_sum; float* n_sum;  /* [K][d], [K]                                               */
    float* hist;                 /* optional ind_hist accumulator [K] (util.py:107-123)       */
} aew_vq_stats_t;

typedef struct {                 /* EMA + optional codebook refresh (vqema_bn.py:190-195,216) */
    float* numer; float* denom; const float* z_sum; const float* n_sum;
    float* emb;                  /* written when update_codebook != 0; 2 = only rows with denom > 0 (with
                                    gamma = 0, gamma_comp = 1 that is the centroid step of Lloyd's k-means,
                                    autoencoder_model.py:171-199)                              */
    int32_t K, d, update_codebook;
    float gamma, gamma_comp;
    const uint32_t* guard;       /* ABI 20, optional: device word; non-zero = the op does nothing (see aew_adam_t.guard) */
} aew_vq_ema_t;

typedef struct {                 /* d(ze) = d(zq) + gscale*gamma * d(min_dist)/d(ze)          */
    const float* ze; const float* emb; const int64_t* ind; const float* dzq;
    int32_t Q, d, d_pitch, metric;
    float coef;                  /* vq_gamma * upstream loss gradient                         */
    float demb_coef;             /* upstream loss gradient (weight of the l2 term, vq_bn.py:78) */
    float* dze;                  /* [Q][d_pitch]                                              */
    float* demb;                 /* optional [K][d], pre-zeroed: d/d(emb) of sum (sg(ze)-emb)^2 */
    const float* gmul;           /* optional device scalar: upstream d(L)/d(loss) of THIS backward call; multiplies
                                    coef and demb_coef at run time (a captured graph stays valid for any value)  */
} aew_vq_bwd_t;

typedef struct {                 /* jitter gather (wavenet.py:330-336) fp32 -> bf16           */
    const float* src; int64_t src_bs; int32_t src_pitch;   /* [B][N][C]                       */
    const int64_t* jitter; int32_t jit_pitch;              /* [B][N]                          */
    uint16_t* dst; int64_t dst_bs; int32_t dst_pitch;      /* [B][N][C_pad] bf16              */
    int32_t B, N, C, C_pad;
    int32_t take_compat;         /* 1: reproduce torch.take flattening (SURVEY C-1)           */
} aew_lc_gather_t;

typedef struct {                 /* transpose of the gather: dsrc[b][j][c] += d[b][t][c]       */
    const float* d; int64_t d_bs; int32_t d_pitch;
    const int64_t* jitter; int32_t jit_pitch;
    float* dsrc; int64_t dsrc_bs; int32_t dsrc_pitch;
    int32_t B, N, C;
    int32_t take_compat;
} aew_lc_scatter_t;

typedef struct {                 /* per-(batch,layer) gated bias incl. speaker term           */
    /* bias[b][l][pack(co)] = conv_bias_l[co] + sum_j V_l[co][C_lc + j] * gc[b][j]
       gc[b] = Wspk[:, voice[b]] + bspk             (wavenet.py:135-139 folded)   */
    const float* params;         /* flat fp32 parameter buffer                                */
    const int64_t* voice;        /* [B]                                                       */
    const int64_t* off_bias_sig; const int64_t* off_bias_gate;  /* DEVICE [L] offsets or -1   */
    const int64_t* off_proj_sig; const int64_t* off_proj_gate;  /* DEVICE [L] offsets         */
    int64_t off_spk_w, off_spk_b;                              /* -1 if no bias               */
    int32_t B, L, D, D_pad, C_lc, G, n_speakers;
    float* bias;                 /* [B][L][2*D_pad]                                           */
    float* gc;                   /* [B][G] saved for backward                                 */
} aew_spk_bias_t;

typedef struct {                 /* backward of the above from per-batch column sums of dfg; B <= 16, G <= 16
                                    (AEW_E_UNSUP beyond: the kernel holds both in registers)  */
    const float* params; const int64_t* voice;
    const int64_t* off_bias_sig; const int64_t* off_bias_gate;
    const int64_t* off_proj_sig; const int64_t* off_proj_gate;
    int64_t off_spk_w, off_spk_b;
    int32_t B, L, D, D_pad, C_lc, G, n_speakers;
    const float* colsum;         /* [B][L][2*D_pad]                                           */
    const float* gc;             /* [B][G]                                                    */
    float* grads;                /* flat fp32 gradient buffer (same offsets as params)        */
    int32_t colsum_running;      /* layers l < colsum_running: colsum[b][l] holds the sum over batch elements 0..b
                                    (aew_gemm_tn_t.snap_out); the per-batch value is colsum[b] - colsum[b-1]  */
    int32_t layer_range;         /* 0: all L layers.  Else first layer | count << 16: this op covers layers [first, first +
                                    count) only (data parallel with two grouped wgrad launches: the upper layers' bias /
                                    projection gradients right after the first group; the speaker-embedding sums of the
                                    parts add up in `grads`)                                                       */
    /* ABI 20, deterministic form (aew_tuning_t.deterministic != 0 and both pointers set): the speaker-embedding sums -
     * one term per (layer, filt | gate) block - go to det_scratch ([count][2][chunks][16][16] floats, count = layers of
     * this op) and the block that draws the last ticket (det_tickets[0]: zeroed once by the caller, left zero by every
     * launch) adds them layer ascending, filt before gate, batch element ascending; without it they are fp32 atomics. */
    float* det_scratch; uint32_t* det_tickets;
} aew_spk_bwd_t;

typedef struct {                 /* base layer as a column gather (wavenet.py:348-351)        */
    const float* wav; int32_t wav_pitch; int32_t wav_off;   /* [B][n_wav] float-encoded ints  */
    const float* W; const float* bias;                      /* [R][Q] (k=1), [R] or NULL      */
    const float* Wt;             /* optional transposed copy [Q][R_pad] (pad columns zero): a row of x
                                    is then ONE contiguous read instead of R strided ones            */
    int32_t B, T, R, R_pad, Q;
    uint16_t* x; int64_t x_bs; int32_t x_pitch;             /* bf16 [B][T][R_pad]             */
    uint16_t* onehot; int64_t oh_bs; int32_t oh_pitch;      /* optional bf16 [B][T][Q_pad]    */
    int32_t Q_pad;
    int32_t ones_channel;        /* 1: write 1.0 into pad channel R (requires R < R_pad).  The padded
                                    weight rows/cols are zero, so the constant propagates through the
                                    residual adds untouched and column R of every gated-layer wgrad
                                    GEMM equals the column sums of dfg (= the gated bias gradients)   */
} aew_base_gather_t;

typedef struct {                 /* fused log-softmax + NLL (+ gradient)  wavenet.py:543-547  */
    const float* logits; int64_t bs; int32_t pitch;         /* fp32 [B][w][Q_pad]             */
    const float* wav; int32_t wav_pitch; int32_t tgt_off;   /* target[b][u] = wav[b][tgt_off+u] */
    int32_t B, w, Q, Q_pad;      /* positions u in [0, w-1) carry loss; u = w-1 is dropped    */
    float* nll;                  /* fwd: [B][w] per-position nll (0 at u=w-1)                 */
    float* ptgt;                 /* fwd: [B][w] probability of the target (chassis.py:266-270)*/
    uint16_t* dlogits; int64_t dl_bs; int32_t dl_pitch;     /* bwd: bf16 [B][w][Q_pad]        */
    float scale;                 /* bwd: d(loss)/d(nll term)                                  */
    int32_t backward;
    const float* gmul;           /* optional device scalar: upstream d(L)/d(loss), multiplies scale at run time */
    float* peak; int32_t* amax;  /* fwd, optional: [B][w] peak log-probability max_c log p(c) and its (lowest) class per
                                    position - what aew_vq_diag_t reduces (vqema_bn.py:261-263) without reading the
                                    logits again                                               */
} aew_softmax_nll_t;

typedef struct {                 /* MFCC + delta + delta-delta front-end on the device (mfcc.py:39-76: librosa.feature.mfcc
                                    and librosa.feature.delta on the host in the reference's DataLoader, data.py:230).
                                    Frame f of the call covers y[f*hop - win/2, f*hop + win/2), y = left_pad zeros + wav,
                                    reflected at both ends; |DFT|^2 -> mel -> dB (floor max - 80 over the whole call)
                                    -> DCT-II -> trim -> Savitzky-Golay derivatives (width 9).  Tables come from the host. */
    const float* wav; int64_t wav_bs; int32_t n;            /* [B][n] samples (float-encoded)                  */
    int32_t B, win, hop, n_bins, n_mels, n_mfcc;            /* win <= 1024, n_bins = win/2 + 1, n_mels <= 128, n_mfcc <= 16 */
    int32_t left_pad, trim_left, trim_right, n_frames;      /* n_frames = 1 + (left_pad + n) / hop, before trimming */
    const float* window;         /* [win]                                                      */
    const float* twiddle;        /* [win][2]  cos, sin (2 pi j / win)                          */
    const float* melw;           /* [n_mels][n_bins]                                           */
    const float* dct;            /* [n_mfcc][n_mels]                                           */
    const float* sg;             /* [2][81]: per derivative 9 interior taps, 4x9 left-edge and 4x9 right-edge rows */
    float* scratch;              /* B * n_frames * (n_mels + 1 + n_mfcc) floats                */
    float* out; int64_t out_bs; int32_t out_pitch;          /* [B][3*n_mfcc][out_pitch]; frames n_frames - trims (>= 9) */
} aew_mfcc_t;

typedef struct {                 /* per-step diagnostics of the reference's loss module as fused reductions
                                    (vqema_bn.py:155-160 and :251-264, util.py:98-105; chassis.py:180-185 reads them):
                                    out[0..1] min / max ||ze||, [2..3] min / max ||emb||, [4] entropy (bits) of the
                                    normalised index histogram, [5] codes used this step, [6] mean, [7] unbiased std
                                    of the peak log-probability over the positions that carry loss, [8] number of
                                    distinct arg-max classes.  Groups with a NULL pointer are skipped (zeros).   */
    const float* ze; int32_t Q, d, d_pitch;
    const float* emb; int32_t K;
    const float* hist;           /* accumulated index histogram [K] (aew_vq_stats_t.hist)      */
    const float* n_sum;          /* this step's counts [K]                                     */
    const float* logits; int64_t bs; int32_t pitch;         /* fp32 [B][w][>= n_quant]; u = w-1 is dropped */
    int32_t B, w, n_quant;       /* n_quant <= 256                                             */
    void* scratch;               /* >= 2 KiB, written by the op (per-slice partial results; with `logits` the
                                    class bins).  NULL (not with `logits`): one block does everything            */
    float* out;                  /* [12]                                                       */
    const float* peak; const int32_t* amax;   /* instead of `logits`: the per-position arrays aew_softmax_nll_t wrote
                                                 ([B][w], u = w-1 dropped); out[6..8] come from them               */
} aew_vq_diag_t;

typedef struct {                 /* first two moments of a channels-last view: the gradient statistics run() reports
                                    every step (autoencoder_model.py:252-257 mel_grad_sd / bn_grad_sd,
                                    mfcc_inverter.py:100-106 mel_grad_sd / mel_grad_mean), over x[b][m][0:cols],
                                    b < batch, m < rows.  out[0] mean, out[1] unbiased std (torch.std), out[2] sum,
                                    out[3] sum of squares.  One block, fp64 accumulation in a fixed order.       */
    aew_view_t x; int32_t rows, cols, batch;
    float* out;                  /* [4] */
} aew_moments_t;

typedef struct {                 /* column sums over rows: out[b][n] (+)= sum_m x[b][m][n]     */
    aew_seg_t x; int32_t dtype; int32_t M, N, batch;
    float* out; int64_t out_bs; int32_t accumulate;
    /* ABI 20, deterministic form (aew_tuning_t.deterministic != 0 and both pointers set): the row chunks write their
     * partial sums to det_scratch (aew_colsum_det_size floats) instead of adding atomically; the block that draws the
     * last ticket of its output (det_tickets: zeroed once by the caller, left zero by every launch) adds them in a
     * fixed order - batch element ascending, row chunk ascending - and is the only writer of that output.          */
    int32_t pad_;
    float* det_scratch; uint32_t* det_tickets;
} aew_colsum_t;

typedef struct {                 /* v_i = scale_i * sum(x_i[0:n_i]);  out[1+i] = v_i;
                                    out[0] = sum_i (clamp_i ? post_scale_i*max(v_i, clamp_min_i) : v_i)
                                    (loss scalar; the clamp is SGVB's free-nats, vae_bn.py:113-116) */
    const float* x[4]; int32_t n[4]; float scale[4]; int32_t clamp[4]; float clamp_min[4];
    float post_scale[4]; int32_t n_terms;
    float* out;                  /* [5] */
    const float* post_scale_dev[4];   /* if non-NULL, post_scale_i is read from device memory at run time
                                         (values that change every step, e.g. the KL anneal weight, must not
                                         be frozen into a captured graph)                                    */
} aew_reduce_t;

typedef struct {                 /* fused Adam over a flat fp32 buffer (torch.optim.Adam defaults,
                                    checkpoint.py:49-50)                                      */
    float* p; const float* g; float* m; float* v;
    int64_t n;
    float lr, beta1, beta2, eps;
    float bc1, bc2;              /* 1-beta1^t, 1-beta2^t                                      */
    float grad_scale;            /* g is multiplied by this first (e.g. 1/world for a mean)   */
    int32_t pad_;
    const uint32_t* guard;       /* ABI 20, optional: device word read at launch time; non-zero = the update is a no-op.
                                    The engine points it at the STICKY word of its chained launches (aew_nt_chain_t.sticky):
                                    a step in which a hand-off wait gave up does not reach the parameters.              */
} aew_adam_t;

typedef struct { void* ptr; int64_t bytes; } aew_zero_t;

typedef struct {                 /* VAE reparameterisation (vae_bn.py:44-53) and its backward */
    const float* lin; int32_t lin_pitch;   /* [Q][2d pitch]: mu | log_sigma_sq               */
    const float* eps;                       /* [Q][d]                                        */
    int32_t Q, d, d_pitch;
    float* sample;                          /* [Q][d_pitch]                                  */
    float* kl_terms;                        /* fwd: [Q] per-row sum of 1+ls-mu^2-sigma^2     */
    const float* dsample; float kl_coef;    /* bwd: d(loss)/d(sample); d(loss)/d(KL)          */
    const float* kl_value; float free_nats; /* bwd: if kl_value != NULL the KL gradient is gated by
                                               [kl_value[0] >= free_nats] (torch.clamp backward)   */
    float* dlin;                            /* bwd: [Q][2d pitch]                            */
    int32_t backward;
    const float* kl_coef_dev;               /* if non-NULL, kl_coef is read from device memory at run time  */
    const float* gmul;                      /* optional device scalar: upstream d(L)/d(loss), multiplies kl_coef */
} aew_vae_t;

typedef struct {                 /* AE norm term (ae_bn.py:36-38): | ||ze|| - 1 | per row      */
    const float* ze; int32_t Q, d, d_pitch;
    float* term;                 /* fwd: [Q]                                                  */
    const float* dze_in; float coef; float* dze;  /* bwd: dze = dze_in + coef*sign*ze/||ze||   */
    int32_t backward;
    const float* gmul;           /* optional device scalar: upstream d(L)/d(loss), multiplies coef              */
} aew_ae_norm_t;

typedef struct {                 /* time-jitter indices on the device (jitter.py:13-33).  out[b][t] = t - 1 + X[b][t],
                                    X in {0,1,2}; X = 1 for t < 2.  mode 0 = what the reference does at HEAD:
                                    X iid with P = [p, 1-2p, p] (its conditional table is indexed [p1][p1], so the
                                    "no three in a row" rule never fires); mode 1 = the documented rule:
                                    P(X_t | X_t-2, X_t-1) = [p, s, p] except (2,1) -> [0, s/(p+s), p/(p+s)].
                                    Randomness: counter-based, u(b,t) = mix64(seed, step, b, t) / 2^53 in [0,1);
                                    X = (u >= c0) + (u >= c1) on the cumulative table (oracle/jitter_rng.py is the
                                    same arithmetic in numpy: bit-identical output).                                  */
    int64_t* out; int32_t out_pitch;
    int32_t B, n;
    float p;
    int32_t mode;
    uint64_t seed, step;
} aew_jitter_t;

/* ---------------------------------------------------------------------------------------
 * Chained NT launch (ABI 19): a run of DEPENDENT bf16 NT ops - the gated stack's G1 / G2 pairs (wavenet.py:100-109,
 * 354-357: the layer loop) or its backward's dz / dx pairs - as ONE kernel launch.  The tiles of all stages form one
 * grid in stage order; a tile of stage s starts once the tiles of earlier stages that produce the rows it reads have
 * published them (tile-granular hand-off through device-scope counters), so a stage's first tiles run in the slots
 * its predecessor's last tiles leave free and the launch pays the fill / drain of a dependent launch once instead of
 * once per op (profiles/r04_notes.md 13: ~10 us each, 84 times per step).
 *   producer tile : out0 stored write-through (sc1), every wave drains its stores, one lane adds 1 to the counter of
 *                   its (stage, batch element, row tile) - device scope
 *   consumer tile : one wave polls the counters of the producer row tiles it reads (relaxed device-scope loads, bounded
 *                   spin), workgroup barrier, then its activation and epilogue operands with device-scope (sc1) loads
 * No assumption on workgroup -> XCD placement.  The bounded spin relies on workgroups being dispatched in index
 * order (every producer of a resident tile has been dispatched): a wait that times out sets *err and the tile runs on.
 * Same kernel bodies, tile shape (256 x 128) and summation order as the stand-alone launches of the default shape
 * (k_gemm_nt_bf16<.., 4> / k_gemm_nt_bf16_win), so results are bit-identical to the serial plan.
 *
 * In a plan the chain op sits IN FRONT of its n_ops stage ops (plain AEW_OP_GEMM_NT records).  When chaining is on
 * (aew_tuning_t.nt_chain, default 1) and the call is not in per-op timing mode, aew_run_plan launches the chain and
 * skips the n_ops records; otherwise the chain op does nothing and the records run as stand-alone launches (what the
 * timing pass and the CPU plan interpreter execute) - the two forms read and write the same buffers.
 * The stage table is built on the HOST by aew_nt_chain_build from the same n_ops descriptors (dependencies derived
 * from the segment / view records; anything it cannot prove safe is refused) and uploaded by the caller.
 * ------------------------------------------------------------------------------------- */
#define AEW_CHAIN_MAXDEP 3
typedef struct {
    int32_t cnt_base;            /* first counter of the producer stage                                       */
    int32_t n_mt;                /* its row tiles per batch element                                           */
    int32_t need;                /* arrivals per row tile = its N tiles                                       */
    int32_t d_lo, d_hi;          /* consumer rows [m_first, m_last] read producer rows [m_first + d_lo, m_last + d_hi] ... */
    int32_t c_lo, c_hi;          /* ... clipped to [c_lo, c_hi] (rows the producer writes and the consumer can read) */
    int32_t bm;                  /* producer tile rows                                                        */
} aew_chain_dep_t;

typedef struct {
    aew_gemm_nt_t g;
    int32_t kind;                /* kernel body: 0 plain K loop, 1 one-window (d <= 16), 2 one-window (d <= 64)   */
    int32_t first_block;         /* multiple of 8                                                             */
    int32_t n_blocks;
    int32_t n_mt, n_nt;
    int32_t cnt_base;            /* counters [batch][n_mt] of this stage                                      */
    int32_t publish;             /* a later stage waits for this one                                          */
    int32_t n_deps;
    aew_chain_dep_t dep[AEW_CHAIN_MAXDEP];
} aew_nt_stage_t;

typedef struct {
    const aew_nt_stage_t* stages;    /* device                                                                */
    const uint16_t* block_stage;     /* device: stage of blocks [8 i, 8 i + 8)                                */
    uint32_t* counters;              /* device, 16-byte aligned, n_counters + 8 words rounded up to 16 bytes, zeroed by the launch:
                                        the counters, then [n_counters] timeout flag (stage + 1 of a wait that gave up),
                                        [+1] tiles that found a producer unfinished, [+2] the longest wait in polls */
    int32_t n_stages, n_blocks, n_counters;
    int32_t set;                     /* 0: GATED / STORE bodies (forward), 1: DFG / STORE bodies (backward)    */
    int32_t n_ops;                   /* stage ops that follow this op in the plan                              */
    int32_t spin_max;                /* polls before a wait gives up (0: default 1 << 18)                      */
    int32_t flags;                   /* 4 = consumers also issue an agent-scope acquire fence (A/B; the sc1 loads make it redundant);
                                        2 = the caller zeroes `counters` itself before the launch (a plan with several chains:
                                        one AEW_OP_ZERO for all of them) */
    int32_t built_window;            /* ABI 20: aew_tuning_t.nt_window the stage table was built under (it decides which stages
                                        take the one-window body, i.e. their summation order).  A launch under a record that
                                        disagrees runs the stage ops one by one instead - the chain never executes another
                                        kernel than the serial plan would                                                */
    uint32_t* sticky;                /* ABI 20, optional device word that NO launch clears: a wait that gives up also stores
                                        (stage + 1) here (atomic max).  aew_adam_t.guard / aew_vq_ema_t.guard read it.    */
} aew_nt_chain_t;

/* Host logic only.  descs[0..n): the stage descriptors in execution order.  Fills stages_out[n] (host memory; `g` copied
 * in), block_stage_out[cap_blocks / 8] and *n_blocks / *n_counters / *set.  force != 0: chain also launches the stand-alone
 * launcher would run on its small-launch shapes (tests).  Returns 0, AEW_E_UNSUP if the run cannot be chained (a
 * descriptor outside the default bf16 shapes, a dependency the builder cannot express, a buffer reused inside the run),
 * AEW_E_ARG on malformed input. */
int aew_nt_chain_build(const aew_gemm_nt_t* descs, int n, aew_nt_stage_t* stages_out, uint16_t* block_stage_out,
                       int cap_blocks, int* n_blocks, int* n_counters, int* set, int force);
/* Producer row tiles consumer tile (first row m0 of the 256-row tile) of `stage` waits for through dependency `dep`:
 * [*t_lo, *t_hi] (empty if *t_lo > *t_hi).  The kernel uses the same arithmetic; exported for tests. */
int aew_nt_chain_dep_tiles(const aew_nt_stage_t* stage, int dep, int m0, int* t_lo, int* t_hi);

/* Split-K hint for a small bf16 launch (aew_gemm_nt_t.k_split): *k_split = the largest S in 2..8 the launcher would honour
 * for g under the calling thread's tuning record with S * blocks <= target_blocks and at least four 64-channel K tiles per
 * range, or 1 (leave g alone); *ws_bytes / *n_tickets = what ksplit_ws / ksplit_tickets must then hold.  Host-only.   */
int aew_gemm_nt_small_split(const aew_gemm_nt_t* g, int target_blocks, int* k_split, int64_t* ws_bytes, int* n_tickets);

/* ---------------------------------------------------------------------------------------
 * plan
 * ------------------------------------------------------------------------------------- */
enum {
    AEW_OP_GEMM_NT = 1, AEW_OP_GEMM_TN, AEW_OP_COPY_TABLE, AEW_OP_VQ_NEAREST, AEW_OP_VQ_STATS,
    AEW_OP_VQ_EMA, AEW_OP_VQ_BWD, AEW_OP_LC_GATHER, AEW_OP_LC_SCATTER, AEW_OP_SPK_BIAS,
    AEW_OP_SPK_BWD, AEW_OP_BASE_GATHER, AEW_OP_SOFTMAX_NLL, AEW_OP_COLSUM, AEW_OP_REDUCE,
    AEW_OP_ADAM, AEW_OP_ZERO, AEW_OP_VAE, AEW_OP_AE_NORM, AEW_OP_JITTER, AEW_OP_VQ_DIAG, AEW_OP_MFCC,
    AEW_OP_MOMENTS, AEW_OP_GEMM_TN_GROUP, AEW_OP_NT_CHAIN
};

/* Lanes.  A plan is a sequential program; `lane` lets the caller mark ops that are OFF the
 * critical chain (weight gradients, bias column sums, weight packing) so that they execute on side
 * HIP streams / parallel hipGraph branches and fill the tails of the chain's kernels:
 *   lane 0 (main)      runs after the preceding lane-0 ops; it waits for preceding side ops only if
 *                      `join` != 0.  The end of the plan is an implicit join.
 *   lane k = 1..5      runs after ALL lane-0 ops that precede it in the plan and after the preceding ops
 *                      of the SAME lane; with `join` != 0 also after the preceding ops of every other side
 *                      lane.  Ops on different side lanes are otherwise unordered (they may run concurrently).
 * Any serial execution in plan order is a valid schedule (that is what the timing mode and
 * aew_set_lanes(0) do), so a lane assignment is correct iff it respects every true dependency. */
typedef struct {
    int32_t kind;
    int32_t tag;                 /* caller-defined label, reported by the timing interface    */
    int32_t lane;                /* 0 main, 1..5 side lanes                                    */
    int32_t join;                /* 1: wait for the preceding ops of all (other) side lanes first;
                                    10+k (lane-0 ops): wait for the preceding ops of side lane k only */
    union {
        aew_gemm_nt_t nt; aew_gemm_tn_t tn; aew_copy_table_t copy; aew_vq_nearest_t vqn;
        aew_vq_stats_t vqs; aew_vq_ema_t vqe; aew_vq_bwd_t vqb; aew_lc_gather_t lcg;
        aew_lc_scatter_t lcs; aew_spk_bias_t spk; aew_spk_bwd_t spkb; aew_base_gather_t base;
        aew_softmax_nll_t sm; aew_colsum_t cs; aew_reduce_t red; aew_adam_t adam; aew_zero_t zero;
        aew_vae_t vae; aew_ae_norm_t aen; aew_jitter_t jit; aew_vq_diag_t diag; aew_mfcc_t mfcc;
        aew_moments_t mom; aew_gemm_tn_group_t tng; aew_nt_chain_t chain;
    } u;
} aew_op_t;

/* Library / build identification. */
int aew_abi_version(void);
/* sizeof(aew_op_t) etc. so the binding can verify its struct mirrors. */
int aew_sizeof(int which);      /* 0 op, 1 gemm_nt, 2 gemm_tn, 3 seg, 4 view, 5 copy_rec, 6 actor, 7 sampler, 8 tuning, 9 nt_stage, 10 nt_chain */

/* Execute ops[0..n) in order on `stream` (a hipStream_t).  Returns at the first error and
 * writes the failing index to *fail_index if non-NULL. */
int aew_run_plan(const aew_op_t* ops, int n, void* stream, int* fail_index);

/* hipGraph form of a plan: capture once (on a private stream; pointers and scalars inside the
 * ops are frozen), replay with one launch per step on any stream.  `exec` is an opaque handle. */
int aew_graph_capture(const aew_op_t* ops, int n, void** exec, int* fail_index);
int aew_graph_launch(void* exec, void* stream);
int aew_graph_destroy(void* exec);

/* Per-op timing: while enabled, aew_run_plan brackets every op with HIP events on `stream`;
 * aew_timing_read synchronises the stream and returns elapsed ms per executed op (in
 * execution order since the last enable) and its tag.  aew_timing_enable(1): every op is its own launch (the stage ops of
 * an AEW_OP_NT_CHAIN run one by one, the chain op itself is an empty interval); (2): a chain runs as the ONE launch it
 * is in the untimed plan and carries the time, its stage ops are empty intervals.  One interval per op either way. */
/* =======================================================================================
 * Tuning context (ABI 17).  Every aew_set_* switch below edits ONE process-wide instance of this record; a caller that
 * wants its own settings - two engines with different shapes in one process, an A/B that must not leak - passes a record
 * to aew_run_plan_tuned / aew_graph_capture_tuned instead, which applies to that call only (thread-local for its duration;
 * a captured graph keeps the settings it was captured under).  aew_tuning_default fills in the library defaults,
 * aew_tuning_get the current process-wide values.  None of the fields changes a result beyond what the individual
 * switch documents (bit-identical shapes; the one-window kernel's summation order).
 * The tn_* fields that decide a TN op's split-K plan (tn_fold_rows, tn_target_blocks, tn_small_tiles, tn_small_target,
 * tn_big, tn_big_target) are read from the PROCESS-WIDE record also inside a *_tuned call: plan construction
 * (aew_tn_slabs) sized the slab buffers under it, and a launch must write exactly those slabs.  Set them with
 * aew_tuning_set / the aew_set_tn_* calls BEFORE building a plan.
 * ======================================================================================= */
typedef struct {
    int32_t nt_wave_rows;
    int32_t nt_pipe;
    int32_t nt_rows192;
    int32_t nt_small_tiles;
    int32_t nt_small_n64;
    int32_t nt_small_w8;
    int32_t nt_small_deep;
    int32_t nt_window;
    int32_t nt_mem128;
    int32_t nt_deep;
    int32_t nf_loaders;
    int32_t nf_deep;
    int32_t fn_enable;
    int32_t fn_ring3;
    int32_t tn_safe;
    int32_t tn_big;
    int32_t tn_big_target;
    int32_t tn_fold_rows;
    int32_t tn_target_blocks;
    int32_t tn_small_tiles;
    int32_t tn_small_target;
    int32_t lanes;
    int32_t tn_cursor_epoch;     /* ABI 18: aew_set_tn_cursor */
    int32_t tn_cursor_slack;
    int32_t nt_chain;            /* ABI 19: 1 (default) AEW_OP_NT_CHAIN ops launch their chain, 0 their stage ops run one by one */
    int32_t deterministic;       /* ABI 20: 1 (default) every sum of the training step has ONE order: column sums and the
                                    speaker-embedding gradient use their ticketed forms where the descriptor carries
                                    det_scratch / det_tickets; 0 = fp32 atomics (A/B of the cost)                   */
    int32_t tn_mfma32;           /* ABI 20: 1 = the grouped weight-gradient launch (tile = 128, no cursor) on v_mfma_f32_32x32x16_bf16:
                                    half the MFMA instructions per stage; same products, 16-row instead of 32-row partial sums
                                    (equal to fp32 rounding, not bit for bit, to the 16 x 16 x 32 form).  0 (default)            */
    int32_t reserved_[5];
} aew_tuning_t;
int aew_tuning_default(aew_tuning_t* out);
int aew_tuning_get(aew_tuning_t* out);
int aew_tuning_set(const aew_tuning_t* in);          /* replaces the process-wide instance (values are clamped like the setters') */
int aew_run_plan_tuned(const aew_op_t* ops, int n, void* stream, int* fail_index, const aew_tuning_t* tuning);
int aew_graph_capture_tuned(const aew_op_t* ops, int n, void** exec_out, int* fail_index, const aew_tuning_t* tuning);
/* aew_nt_chain_build under `tuning` (NULL: the process-wide record) - pass the record the plan will RUN under (aew_run_plan_tuned) and
 * store its nt_window in aew_nt_chain_t.built_window. */
int aew_nt_chain_build_tuned(const aew_gemm_nt_t* descs, int n, aew_nt_stage_t* stages_out, uint16_t* block_stage_out,
                             int cap_blocks, int* n_blocks, int* n_counters, int* set, int force, const aew_tuning_t* tuning);

/* Box fingerprint (measurement aid): sustained rate of THIS device on a pure bf16 MFMA loop (out[0], TFLOP/s) and on a
 * device-to-device copy of copy_bytes (out[1], TB/s read + written), HIP events on `stream`.  scratch: device memory,
 * 16-byte aligned, >= 2 * copy_bytes (>= 1 MiB each; contents are overwritten).  Synchronises the stream. */
int aew_probe_box(void* scratch, int64_t scratch_bytes, int64_t copy_bytes, void* stream, float* out);

int aew_timing_enable(int on);
int aew_timing_read(float* ms, int32_t* tags, int capacity, int* count);

/* Tile / wave shape of the bf16 NT kernel.  Same results bit for bit; tuning / A-B aid (profiles/r01_notes.md).
 *   64  (default) 256x128 tiles, 8 waves of 64x64, K tiles of 32; per launch the 192x128 or 64x128 shapes where
 *       the cost model prefers them (aew_set_nt_rows192, aew_set_nt_small_tiles)
 *   128 256x128 tiles, 4 waves of 128x64          256 256x256 tiles, 8 waves of 128x64 (N_pad % 256 == 0)
 *   0   128x128 tiles with K tiles of 64          1   256x128 tiles with K tiles of 64 */
int aew_set_nt_wave_rows(int rows);
/* Shapes 128 / 256 only: 1 (default) the register-double-buffered loop, 2 K tiles of 64, 0 the plain loop. */
int aew_set_nt_pipe(int on);
/* impl = 2 descriptors (full-N kernels, aew_fn.hip): 1 (default) use them where the shape is covered, 0 run every
 * impl = 2 descriptor on the tiled kernels instead (A/B and bisecting aid; same results). */
int aew_set_fn(int on);
int aew_set_fn_ring3(int min_k_tiles);   /* plain full-N GEMMs of >= this many 64-channel K tiles take the three-stage
                                            operand ring (default 16; 0 = never).  Same results either way.       */
/* Which kernel a GEMM_NT descriptor dispatches to under the current settings: 0 k_gemm_nt_bf16 (256- / 192-row tiles),
 * 1 k_gemm_nt_bf16_p64 (64-row tiles, small launches), 2 k_fn, 3 k_gemm_nt_f32, 4 the scalar check kernel, 5 an A/B
 * shape.  Measurement aid: lets a caller group per-op times by kernel the way a rocprofv3 kernel trace does. */
int aew_nt_kernel(const aew_gemm_nt_t* g);
/* Default shape only: memory-bound plain launches (DFG epilogue, or K_total <= 256: wavenet.py:103-109 and the backward of
 * :100-102) as 128-row tiles - 0 the 256- / 192-row tiles, 1 128 x 128 tiles with K tiles of 32 (4 waves, three blocks
 * per CU), 2 128 x 128 tiles with K tiles of 64 (two blocks per CU).  Bit-identical results. */
int aew_set_nt_mem128(int mode);
/* Default shape only (A/B): deep operand rings at one block per CU - 1 256 x 128 tiles on a 6-stage ring, 2 256 x 256
 * tiles (8 waves of 128 x 64) on a 5-stage ring where N_pad % 256 == 0 and as 1 elsewhere, 3 as 2 with every other launch
 * on its default kernel.  0 (default) off.  Same results as the kernel each launch replaces, up to the order in which the
 * one-window kernel interleaves its taps. */
int aew_set_nt_deep(int mode);
/* Default shape only: bf16 NT launches of <= n 256x128 tiles use 64x128 tiles instead (0 = never). */
int aew_set_nt_small_tiles(int n);
/* ... and of those, launches of <= max_blocks blocks (default 256: one block per CU) run a 5-stage operand ring
 * instead of 2 stages: their K loop is DMA latency, not MFMA (0 = never; same results). */
int aew_set_nt_small_n64(int max_blocks);   /* launches of <= this many 64 x 128 blocks run as 64 x 64 tiles (twice the
                                               blocks, 2/3 of the operand bytes per block; default 256, 0 = never)  */
int aew_set_nt_small_deep(int max_blocks);
/* ... as 8 waves of 16 rows x 64 channels (default) or 2 waves of 64 x 64 per block (A/B; same results). */
int aew_set_nt_small_waves(int waves);
/* fp32 NT kernel: launches of <= max_blocks blocks (default 256 = one per CU) run with a 12-14 stage operand ring
 * (the whole LDS of the CU; 16-row tiles if that fits the limit, else 32-row tiles), larger ones with 5 stages and
 * three blocks per CU.  0 = always the 5-stage shape.  The chains themselves - one per output, k ascending - do not
 * change (bit-identical results). */
int aew_set_nf_deep(int max_blocks);
/* fp32 NT kernel: 1 (default) four dedicated loader waves per block issue the LDS-DMA, 0 the four compute waves stage
 * their own operands (A/B; same results). */
int aew_set_nf_loaders(int on);
/* default NT shape only: 192 x 128 tiles (8 waves of 48x64) instead of 256 x 128 — 0 never, 1 (default) where the
 * per-CU cost model prefers them, 2 always.  Results are bit-identical either way. */
int aew_set_nt_rows192(int mode);
/* default NT shape only: descriptors whose first two segments are the SAME buffer at row offsets d <= max_dist apart
 * (the two taps of a dilated convolution, wavenet.py:100-101, and of its input gradient) stage ONE LDS window of
 * 256 + d rows per K tile and issue both taps' MFMAs from it (k_gemm_nt_bf16_win).  Default 64 (the largest
 * supported), 0 = never.  Results agree with the two-segment kernel to fp32 accumulation order, not bit for bit. */
int aew_set_nt_window(int max_dist);
/* 0 (default): ignore aew_op_t.lane - every op on the caller's stream, in plan order.  1: side-lane ops on private
 * streams (branches of a captured graph).  Same results either way (the atomically accumulated bias sums up to order). */
int aew_set_lanes(int on);
/* Row cursor of the grouped weight-gradient launch: a launch whose descriptor carries progress words
 * (aew_gemm_tn_group_t.cursors) is paced.  epoch = 32-row stages per epoch, slack = epochs a tile may run ahead of the slowest
 * tile of its matrix: (0, -) = the defaults 4 / 2 (the process default); (3..64, 1..8) explicit; (-1, -) = never pace. */
int aew_set_tn_cursor(int epoch, int slack);

/* Number of fp32 partial slabs a TN op writes into `out` (depends on the split heuristic). */
int aew_tn_slabs(const aew_gemm_tn_t* g);
/* Validate one descriptor of a grouped launch (aew_gemm_tn_group_t.descs lives in device memory, so the launcher
 * cannot): 0 or AEW_E_*.  Host logic only. */
int aew_tn_group_check(const aew_gemm_tn_t* g);
/* What the deterministic form of a column-sum op needs (aew_colsum_t.det_scratch / det_tickets): *floats of scratch,
 * *tickets words (zeroed once by the caller).  Host logic only; the launcher uses the same arithmetic. */
int aew_colsum_det_size(const aew_colsum_t* c, int64_t* floats, int32_t* tickets);
/* 1 if the op's batch loop is folded into one slab per split (short contractions). */
int aew_tn_fold(const aew_gemm_tn_t* g);
/* Contraction length (rows x batch) up to which TN ops fold the batch; default 4096. */
int aew_set_tn_fold_rows(int rows);
/* Split-K target of the TN ops (blocks per launch, default 512).  Changes aew_tn_slabs(): set it
 * before building a plan. */
int aew_set_tn_target_blocks(int n);
/* TN ops whose output has <= max_tiles 128x128 tiles are split to ~target_blocks blocks only (fewer
 * slabs).  Like the call above it changes aew_tn_slabs(): set before building a plan. */
int aew_set_tn_small(int max_tiles, int target_blocks);
/* 1: read TN fragments with a scalar LDS gather instead of ds_read_b64_tr_b16 (debug aid). */
int aew_set_tn_safe(int on);
/* 256 x 256 output tiles for large bf16 wgrads (k_gemm_tn_bf16_big): on = 0 / 1, target_blocks > 0 sets the split-K
 * block target.  Changes the slab count: call before plans are built (aew_tn_slabs follows it). */
int aew_set_tn_big(int on, int target_blocks);

/* =======================================================================================
 * Autoregressive sampler — replaces WaveNet.forward_test (wavenet.py:367-531: the per-sample Python
 * loop over base_layer / conv_layers / post1 / post2 / softmax / torch.multinomial with rolling
 * per-layer buffers, driven by InferenceChassis, chassis.py:283-349).
 *
 * One PERSISTENT kernel; every wavefront is an ACTOR that owns a fixed slice of one layer's weights
 * (MFMA A-fragments resident in its VGPRs for the whole generation) and processes the work items
 * (t, b) — time step t of stream-batch b, 16 streams per batch — in the same global order
 * t = 0..n_steps-1, b = 0..n_batches-1.  Actors exchange 16-stream activation rows through global
 * memory and signal with monotone sequence flags: flag = seq(t, b) = t*n_batches + b + 1 once item
 * (t, b) is published.  Per layer l (dilation d):
 *   EARLY(l,p)  pre-activations of channel pair-tile p from h_l(t-d) and cond(t)  (off the critical path)
 *   LATE(l,p)   adds the h_l(t) taps, gates: z_l = tanh(filt)*sigmoid(gate)          (wavenet.py:100-103)
 *   RES(l,q)    h_{l+1}(t) = h_l(t) + W_res z_l                                      (wavenet.py:108-110)
 *   SKIP(l,q)   skip sum += W_skp z_l                                                (wavenet.py:104,458)
 * then POST1 / POST2 (wavenet.py:461-462) and SAMPLE (softmax + inverse-CDF draw from a counter RNG in
 * place of torch.multinomial, wavenet.py:463-466; writes the base-layer row h_0(t+1), wavenet.py:452).
 * With n_batches > 1 the layers work on different stream-batches at the same time (a pipeline).
 * Every spin is bounded: if a producer never arrives the kernel sets status[0] and all actors exit.
 * ===================================================================================== */
#define AEW_ACT_NONE   (-1)
#define AEW_ACT_EARLY  0
#define AEW_ACT_LATE   1
#define AEW_ACT_RES    2
#define AEW_ACT_SKIP   3
#define AEW_ACT_POST1  4
#define AEW_ACT_POST2  5
#define AEW_ACT_SAMPLE 6

/* 16-stream activation buffer.  Entry of stream-batch b at time t (BYTES): ptr + b*bstride + (t % ring)*entry.
 * layout 0 (rows): stream i's channels are contiguous, row at + i*pitch.
 * Rows that several actors write are stored PRODUCER-CONTIGUOUS instead, so that every actor writes whole cache
 * lines of its own and a consumer's K tile (32 channels x 16 streams) is one 1 KiB block:
 * layout 1 (h_l, relu(post1)): [32-channel group][stream][32 ch]  -> channel c of stream i at (c/32)*1024 + i*64 + (c%32)*2
 * layout 2 (z_l):              [16-channel pair ][stream][16 ch]  -> (c/16)*512 + i*32 + (c%16)*2 */
typedef struct {
    void*   ptr;
    int64_t bstride;
    int64_t entry;
    int64_t pitch;
    int32_t ring;                /* >= 1 */
    int32_t layout;
} aew_sbuf_t;

/* wait until flags[j*flag_stride] >= seq(t - lag, b) for j < n (skipped while t < lag) */
typedef struct {
    const uint32_t* flags;
    int32_t n;                   /* <= 32 */
    int32_t lag;
} aew_wait_t;

typedef struct {
    int32_t role;                /* AEW_ACT_* */
    int32_t layer, index;        /* informational */
    int32_t nt;                  /* 16-channel output tiles this actor owns: 1 or 2                       */
    int32_t nk;                  /* K tiles (32 channels) of the register-resident weights `w`             */
    int32_t nk2;                 /* EARLY: K tiles of the streamed cond weights `w2`                       */
    int32_t dil;                 /* EARLY: dilation d                                                      */
    int32_t pad;
    aew_wait_t wait[2];
    uint32_t* flag;              /* this actor's sequence flag                                             */
    /* weight fragments, bf16: blob[k][n(2)][lane(64)][8] = W[n*16 + (lane&15)][k*32 + (lane>>4)*8 + j]
     * (the MFMA 16x16x32 A operand); SAMPLE: `w` = base-layer table [Q][in0-row] bf16 (bias folded in)  */
    const void* w;
    const void* w2;
    /* fp32 bias: EARLY per stream [n_streams][bias_pitch] (speaker term folded in), lane reads
     * bias[stream*bias_pitch + n*16 + (lane>>4)*4 .. +3]; POST1/POST2 shared (bias_pitch = 0)            */
    const float* bias;
    int64_t bias_pitch;
    /* role        in0                  in1                      out
     * EARLY       h_l (read at t-d)    cond (ring = n_steps)    partial pre-acts (ring 2; raw lane layout)
     * LATE        h_l (t)              partial pre-acts         z_l     (ptr at this pair's channels)
     * RES         z_l                  h_l (t), own channels    h_{l+1} (t), own channels
     * SKIP        z_l                  skip sum (NULL: layer 0) skip sum, own channels (fp32)
     * POST1       skip sum (fp32)      -                        relu(post1) bf16, own channels
     * POST2       relu(post1)          -                        logits fp32, own channels
     * SAMPLE      logits               -                        h_0 (written at t+1)                        */
    aew_sbuf_t in0, in1, out;
    aew_sbuf_t out2;             /* POST2: optional copy of the logits (ring = n_steps), ptr NULL = off     */
    int32_t n_quant;             /* SAMPLE: Q (multiple of 16, <= 256); streams [index*4, index*4+4)        */
    int32_t row_bytes;           /* SAMPLE: bytes of one base-table row (= h_0 row)                         */
} aew_actor_t;

typedef struct {
    const aew_actor_t* actors;   /* DEVICE array [n_slots]; slot L runs as workgroup L (XCD L % 8)          */
    int32_t n_slots;
    int32_t n_batches;           /* stream-batches of 16 streams                                          */
    int32_t n_steps;             /* T: positions 0..T-1; position 0 is always forced                       */
    int32_t flag_stride;         /* uint32 words between consecutive flags                                 */
    int32_t kr_max;              /* 12 or 16: compiled bound on nk of EARLY / LATE                         */
    int32_t spin_max;            /* polls before an actor gives up (0 = default)                           */
    uint32_t* flags;             /* [n_slots * flag_stride]; zeroed by aew_sampler_run                      */
    uint32_t* status;            /* [4] device words: abort flag, slot, t, b of the first give-up          */
    const int32_t* forced;       /* [n_streams][n_steps]: value fed at position t if >= 0, else the draw   */
    int32_t* wav_out;            /* [n_streams][n_steps] the sequence that was fed (forced or drawn)       */
    uint64_t seed;               /* u(stream, t) = mix64-counter uniform, see oracle/jitter_rng.py         */
    int32_t nap_eighths;         /* 0..7: after publishing an item an actor sleeps through this many eighths of the
                                    running average of its own item period before it polls again (fewer pollers
                                    on the memory-side path); 0 = poll at once                                */
    int32_t pad;
    uint64_t* prof;              /* optional [n_slots][4]: s_memtime cycles (100 MHz) each actor spent waiting,
                                    loading + computing, publishing; [3] = items.  NULL = off               */
} aew_sampler_t;

/* Enqueue one generation on `stream`.  Returns AEW_E_UNSUP if the device cannot keep n_slots
 * wavefront-workgroups resident at once (the actors wait on each other). */
int aew_sampler_run(const aew_sampler_t* s, void* stream);

/* Human-readable string for a return code. */
const char* aew_strerror(int code);

/* Device self-test of the MFMA / LDS-transpose lane mappings this library relies on.
 * scratch: >= 1 MiB device buffer.  Returns 0 if the hardware behaves as assumed. */
int aew_selftest(void* scratch, int64_t scratch_bytes, void* stream, int32_t* detail /*[8]*/);

#ifdef __cplusplus
}
#endif
#endif /* AEWAVENET_H */
