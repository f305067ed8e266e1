/* libavsiam_hip.so - C ABI of the MI355X (gfx950) kernels behind the AVSiam pre-training hot path.
 *
 * The reference (GenjiB/AVSiam) has no native code and no FFI: its hot path is torch.nn calls inside
 * src/models/cav_mae_base.py.  Each entry point below names the reference operation it replaces (file:line
 * relative to /root/reference).  The binding a maintainer adds on the reference side is the ctypes loader of
 * INTEGRATION.md (avsiam_amd/_lib.py is that loader).
 *
 * Conventions
 *  - all pointers are DEVICE pointers; bf16 tensors are raw uint16_t bits; row-major; leading dimensions in elements
 *  - kernels are enqueued on `stream` and never synchronise, allocate or take ownership
 *  - return 0 on success, -1 runtime/launch error, -2 bad argument; avs_last_error() describes the failure
 *  - re-entrant per stream.  Global state: the thread-local error string, and the process-wide TUNING KNOBS below (avs_tuning_set):
 *    plain integers the launchers read; they are written only through this ABI (the host binding sets them once at load from the
 *    AVSIAM_* environment - the library itself never reads the environment and initialises nothing lazily), before kernels are queued
 */
#ifndef AVSIAM_HIP_H
#define AVSIAM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* avs_stream_t;
typedef uint16_t avs_bf16;

const char* avs_last_error(void);
int avs_abi_version(void);
int avs_device_cu_count(void);

/* ---- tuning knobs (no reference counterpart: the reference leaves kernel selection to cuBLAS / cuDNN heuristics).  name:
 *   "gemm_tile" 0 auto | 128 | 256          "gemm_persistent" 0 | 1          "gemm_nt8" 0 | 1 (8-phase 256^2 kernels)
 *   "nt_tile_h" 0 auto | 256 | 224 | 240    "nt_grid" cap of the persistent nt grid, 0 = none
 *   "cu_reserve" compute units EVERY persistent kernel (nt / fp8 nt / tn8 / tn8f grids and split factors) leaves free, a multiple of 8
 *                keeps the XCDs balanced; 0 on one GPU, 8 when a gradient all-reduce overlaps the backward (src/traintest_cavmae_base.py:58-59
 *                is DDP's overlap; RCCL's kernels need CUs WHILE a GEMM runs)
 *   "gemm_ring" 0 | 1 | 2 (default): small forward / input-gradient GEMMs (at most one 128 x 128 workgroup per CU) on the two-buffer kernel | the
 *               4-slot LDS-DMA ring kernel | the ring kernel, and 64 x 128 half-height tiles when even those fill less than half the CUs
 *   "nt_big_min" forward / input-gradient GEMMs with at least this many 256 x 256 output tiles run the persistent 256^2 kernels, smaller ones the
 *               128 x 128 kernels; 0 (default) = half the persistent CU slots
 *   "ln_dma" 0 | 1    "ln_rpw" 0 auto | 4 | 8 | 16    "attn_ring" 0 (default) | 1 (attention forward / dQ with K/V tiles by LDS-DMA ring: same bits, not faster)
 *   "det" 0 (default) | 1: every reduction into a parameter gradient has one writer per element and a fixed order - weight-gradient GEMMs
 *               without a split of their token rows, column sums / vector-matrix product / LayerNorm slab reduce as one block per column group,
 *               atomics-free positional scatter and un-shuffle token sums: two runs of a step give the same bits (a debugging mode, slower;
 *               the reference gets the same from torch.use_deterministic_algorithms).  The host also keeps the step on ONE stream and takes the
 *               fc1 bias gradient by avs_colsum_bf16 instead of the GEMM epilogue's atomics (EngineOptions.deterministic)
 * avs_persistent_cu_slots(): the CUs a persistent grid fills now (device CUs - cu_reserve). */
int avs_tuning_set(const char* name, int value);
int avs_tuning_get(const char* name, int* value);
int avs_persistent_cu_slots(void);

/* ---- LayerNorm with per-row modality affine (Block.norm1/_a/_v, norm2/_a/_v: src/models/cav_mae_base.py:120-122,
 * 135-137,151-152,169-170,190-191; final norms :492,495,563,566,631).  x fp32 [rows,D] -> y bf16.  row_mod (0/1 per
 * row, may be NULL) picks (g0,b0) or (g1,b1); out_map (may be NULL) redirects output row r to y[out_map[r]]; y is bf16
 * (GEMM operand) or fp32 when y_f32 (final norms that feed the fp32 residual stream / the token mean). */
int avs_layernorm_ws_floats(int rows, int D);
int avs_layernorm_fwd(const float* x, const float* g0, const float* b0, const float* g1, const float* b1,
                      const uint8_t* row_mod, const int* out_map, void* y, int y_f32, float* mean, float* rstd, int rows,
                      int D, float eps, avs_stream_t stream);
/* the same, also writing y8 = e4m3(clamp(y * q8, +-448)) (bf16 output only; y8 may be NULL): the fp8 operand of the forward GEMM that
 * consumes this LayerNorm in the fp8-forward mode, without a quantising pass of its own.  q8_dev (may be NULL): a device
 * quantisation record (see avs_fp8_scale_update) - the scale is then read from q8_dev[0] and max |y| folded into q8_dev[2] */
int avs_layernorm_fwd_q8(const float* x, const float* g0, const float* b0, const float* g1, const float* b1,
                         const uint8_t* row_mod, const int* out_map, void* y, int y_f32, float* mean, float* rstd, int rows,
                         int D, float eps, uint8_t* y8, float q8, float* q8_dev, avs_stream_t stream);
/* dx = dres + LN'(dy) (dres may be NULL; dx may alias dres); dy is bf16, or fp32 when dy_f32; dx_bf16 (may be NULL) gets a
 * bf16 copy of dx; dg/db are accumulated (+=); dcol (may be NULL) accumulates the column sum of dx, i.e. the bias gradient
 * of the Linear that produced this residual branch; ws: avs_layernorm_ws_floats.
 * dres_bf16 != 0: dres is bf16 (the residual-gradient stream kept in bf16 between blocks - the previous call's dx_bf16);
 * dx may then be NULL (only dx_bf16 is written); dx_bf16 must not alias a bf16 dres.
 * dx8 / q8 (fp8 backward; both or neither): also dx8 = e5m2(clamp(dx * q8[0])) - the gradient operand of an fp8 input-gradient GEMM -
 * with max |dx| folded into the device record q8 (avs_fp8_scale_update) */
int avs_layernorm_bwd(const void* dy, int dy_f32, const float* x, const float* mean, const float* rstd, const float* g0,
                      const float* g1, const uint8_t* row_mod, const int* out_map, const void* dres, int dres_bf16, float* dx,
                      avs_bf16* dx_bf16, float* dg0, float* db0, float* dg1, float* db1, float* dcol, float* ws, int rows,
                      int D, uint8_t* dx8, float* q8, avs_stream_t stream);
/* Deferred parameter-gradient reduce (a stack's LayerNorm backwards: ONE reduce launch at the end of the stack's backward instead of one per
 * LayerNorm inside it).  avs_layernorm_bwd called with dg0 = db0 = dg1 = db1 = dcol = NULL leaves its per-block partial sums in ws
 * (avs_layernorm_bwd_slabs(rows) slabs of 5 * D floats; a ws of its own per call); avs_layernorm_bwd_reduce_batched adds n such slab sets to their
 * targets: desc (device, 7 * n int64) = {ws, slabs, dg0, db0, dg1, db1, dcol} per set, targets may be 0.  Not in the deterministic mode. */
int avs_layernorm_bwd_slabs(int rows);
int avs_layernorm_bwd_reduce_batched(const long long* desc, int n, int D, avs_stream_t stream);

/* ---- bf16 MFMA GEMMs (nn.Linear / PatchEmbed.proj and their backward: cav_mae_base.py:51,55,60,77,96-99,138-143,
 * 600,634-635).  nt: x = alpha*(A[M,K].B[N,K]^T + bias [*aux] + res[res_idx? res_idx[m] : m]); act 0: out = x;
 * act 1 (timm Mlp fc1 + GELU forward): out = gelu'(x) bf16 - what the backward needs of the pre-activation - and
 * out2 = gelu(x) bf16; act 2 (fc2 input gradient): aux = the saved gelu'(x), out = x.  N%128==0, K%64==0.
 * Columns [0, scale_cols) (a multiple of 64, 0 = none) are multiplied by col_scale in addition: the qkv projection
 * uses it to hand the attention kernels q already multiplied by hd^-0.5 * log2(e), rounded to bf16 once.
 * colsum (bf16 output only, may be NULL): colsum[n] += sum over rows of the output - the bias gradient of the layer
 * whose output gradient this GEMM produces (fc1.bias from the fc2 dgrad), without re-reading the matrix.
 * fp32 output (out_f32 == 1) requires act 0.
 * gelu'(x) as 8-bit fixed-point codes (ABI 2; round((g' + 0.1296875) * 202), g' in [-0.129, 1.129]: a step of 0.005): out_f32 == 2 with act 1 -
 * `out` receives one byte per value, ldo in BYTES; act 3 = act 2 whose `aux` holds those codes (ldaux in bytes) - half the bytes of the one
 * output / operand nothing else reads (as avs_gemm_nt_fp8's out_f32 == 2 / a_e5m2 == 2).  Opt-in (EngineOptions.gelu8). */
int avs_gemm_nt_bf16(const avs_bf16* A, long long lda, const avs_bf16* B, long long ldb, int M, int N, int K,
                     const float* bias, const float* res, long long ldr, const int* res_idx, const avs_bf16* aux,
                     long long ldaux, void* out, long long ldo, int out_f32, avs_bf16* out2, long long ldo2, float alpha,
                     int act, int scale_cols, float col_scale, float* colsum, avs_stream_t stream);
/* ---- fp8 (OCP e4m3) variant of the forward / input-gradient GEMM (BASELINE configs[4]'s "fp8 MFMA path"; not used by the default
 * bf16 path): x = alpha * (A8[M,K] . B8[N,K]^T) + bias (+ res, fp32 output only); act 0: out = x (columns [0, scale_cols) times col_scale);
 * act 1: out = gelu'(x), out2 = gelu(x) (bf16), as in avs_gemm_nt_bf16, and out8 (may be NULL) = e4m3(gelu(x) * out8_scale) for the next fp8 GEMM.
 * fp32 accumulation on v_mfma_f32_16x16x128_f8f6f4,
 * the 8-phase 256x256 kernel of the bf16 GEMM with 128-value K-tiles.  alpha carries 1 / (scale_A * scale_B).  N%256==0, K%128==0,
 * K>=256, leading dimensions multiples of 16.
 * Delayed scaling (no host synchronisation anywhere): a tensor's quantisation state is a DEVICE record of 1024 floats:
 * q[0..3] = {scale, 1 / scale, running max |x| since the last update, saturation events} and 15 shards of the running max at q[64 (1 + k)]
 * (producers add to one shard each, every shard in a 256-byte line of its own: a single address would serialise their atomics).  Where an entry point takes such a record
 * (qa / qw / qw2: operands of the GEMM, de-quantisation factor qa[1] * qw[1]; q8: the e4m3 copy a kernel writes - scale q8[0], amax into
 * q8[2]) it overrides the host float beside it.  avs_fp8_scale_update(q [n][1024], hist [nhist][n], n, nhist, pos, margin, first, count, fmax): per
 * record in [first, first + count), hist[pos] = max(q[2], shards); scale = fmax / (margin * max over hist); q[2] *= 0.9 (the floor the producers filter their atomics against);
 * q[3] += (q[2] * old scale > fmax);
 * fmax = 448 (e4m3 tensors) or 57344 (e5m2: the gradient operands of the input-gradient form).
 * Input-gradient form (a_e5m2 != 0; BASELINE configs[4]'s fp8 path in the backward): A holds e5m2 gradients, B the e4m3 transposed weight;
 * act 0, or act 2 with aux / colsum as in avs_gemm_nt_bf16 (fc2 input gradient); out8 (may be NULL) = e5m2(out * q8[0]) for the next one.
 * gelu'(x) as 8-bit codes (fp8 backward: half the bytes between the two epilogues that write and read it): out_f32 == 2 with act 1 - `out` receives
 * round((gelu'(x) + 0.1296875) * 202) in [0, 255], one byte per element, ldo in bytes; a_e5m2 == 2 with act 2 - `aux` holds such codes, ldaux in bytes.
 * m_split / B2 / bias2 / qw2: a second weight set for rows from m_split (m_split % 256 == 0; needs the records), else m_split = 0.
 * avs_absmax: out = max(out, max |x|) (caller zeroes out; x fp32 or bf16);
 * avs_quantize_fp8: y = e4m3(clamp(x * scale, +-448)) (e5m2 != 0: e5m2, +-57344), n%4==0; with q: scale = q[0], max |x| into q[2]. */
int avs_gemm_nt_fp8(const uint8_t* A, long long lda, const uint8_t* B, long long ldb, int M, int N, int K, const float* bias,
                    const float* res, long long ldr, void* out, long long ldo, int out_f32, avs_bf16* out2, long long ldo2, float alpha,
                    int act, int scale_cols, float col_scale, uint8_t* out8, long long ldo8, float out8_scale, const float* qa,
                    const float* qw, float* q8, int m_split, const uint8_t* B2, const float* bias2, const float* qw2, int a_e5m2,
                    const avs_bf16* aux, long long ldaux, float* colsum, float* colsum2, avs_stream_t stream);
int avs_absmax(const void* x, int is_f32, long long n, float* out, avs_stream_t stream);
int avs_quantize_fp8(const void* x, int is_f32, uint8_t* y, long long n, float scale, float* q, int e5m2, avs_stream_t stream);
/* many bf16 tensors in one launch (all weights of a stack, once per forward): desc [n][4] = {src, dst, numel / 4, record index}, cmap
 * [nchunks][2] = {descriptor, first 4-element group} per chunk of 8192 elements; scales from / amax into the records q [.][1024] */
int avs_quantize_fp8_batched(const long long* desc, const int* cmap, int nchunks, float* q, int e5m2, avs_stream_t stream);
/* many small regions zeroed by one launch: desc [n][2] = {address (16-byte aligned), bytes / 16}, cmap [nchunks][2] = {descriptor, first
 * 16-byte group} per chunk of 64 KB.  No reference counterpart (torch allocates every activation afresh); used for the pad rows of a stack's
 * buffers when the two passes of a step share one activation pool (engine.BufferPool) */
int avs_zero_batched(const long long* desc, const int* cmap, int nchunks, avs_stream_t stream);
int avs_fp8_scale_update(float* q, float* hist, int n, int nhist, int pos, float margin, int first, int count, float fmax, avs_stream_t stream);
/* the same GEMM over TWO weight sets in one launch: rows [0, m_split) of A meet B / bias / colsum, rows [m_split, M) meet
 * B2 / bias2 / colsum2 (same shapes and leading dimension; m_split a multiple of 256).  Replaces the two nn.Linear calls the
 * reference makes per layer for its separate audio and visual towers (cav_mae_base.py:487,489: `blk(v, 'v')` on
 * vit_base.blocks and `blk(a)` on ast_base.blocks), whose row counts alone (8 192 / 31 360 at batch 64) leave the chip
 * partly idle. */
int avs_gemm_nt_bf16_dual(const avs_bf16* A, long long lda, const avs_bf16* B, long long ldb, int M, int N, int K,
                          const float* bias, const float* res, long long ldr, const int* res_idx, const avs_bf16* aux,
                          long long ldaux, void* out, long long ldo, int out_f32, avs_bf16* out2, long long ldo2, float alpha,
                          int act, int scale_cols, float col_scale, float* colsum, int m_split, const avs_bf16* B2,
                          const float* bias2, float* colsum2, avs_stream_t stream);
/* number of kernel dispatches avs_gemm_nt_bf16 has issued so far (a call is one dispatch, or two when the rows left over
 * after the whole rounds of 256x256 tiles go to the half-height-tile kernel): lets bench.py quote a per-DISPATCH average
 * that is directly comparable with rocprofv3's per-kernel statistics */
long long avs_gemm_nt_dispatches(void);
/* tile selection of the nt kernel: 0 = automatic (256x256 when that alone fills the chip, else 128x128), 128, 256 */
int avs_gemm_set_tile(int tile);
/* 1 (default): 256x256 nt tiles run as persistent workgroups (one per CU); 0: one workgroup per tile (A/B measurements) */
int avs_gemm_set_persistent(int on);
/* 1 (default): 256x256 tiles of BOTH GEMMs (nt and tn) run the 8-phase kernels (two wave groups offset by a barrier,
 * 16-KiB staging granules, counted vmcnt); 0: the two-buffer kernels (A/B measurements, the bitwise cross-check of
 * tests/test_kernels_gpu.py).  Environment: AVSIAM_GEMM_NT8=0|1. */
int avs_gemm_set_nt8(int on);
/* tile HEIGHTS of the 8-phase nt kernel.  A persistent workgroup per CU pays whole rounds of tiles, so 0 (default) lets the host mix
 * 256-row and 224-row tiles per GEMM such that the rounds the 256-row tiling needs are filled exactly with cheaper tiles (1122 tiles of
 * 256 rows = 4.4 rounds cost 5 tile-times per CU; 24 + 1254 tiles in two heights cost 4.6); 256 / 224: one height for every tile, 240: half
 * the row tiles of each (A/B measurements, the bitwise cross-check of tests/test_kernels_gpu.py).  With two weight sets the 256-row class
 * covers the first set (the row split is a multiple of 256); the forced one-height modes 224 / 240 apply to one weight set only.
 * Environment: AVSIAM_NT_TILE_H. */
int avs_gemm_set_tile_height(int h);
/* tn (weight gradient): C[N1,N2] += A[M,N1]^T . B[M,N2], fp32 atomics; A and B must be allocated and ZERO up to the
 * next multiple of 64 rows; N1%128==0, N2%128==0; splits<=0 picks a split of the contraction that fills the chip. */
int avs_gemm_tn_bf16(const avs_bf16* A, long long lda, const avs_bf16* B, long long ldb, float* C, long long ldc, int M,
                     int N1, int N2, int splits, avs_stream_t stream);
/* The plan avs_gemm_tn_bf16 takes for (M, N1, N2) with splits <= 0 under the current knobs: output tile size (128 | 256) and the number of
 * splits of the token rows.  Host arithmetic only (no device work, callable without a GPU): pins the split heuristic in the CPU test suite. */
int avs_gemm_tn_plan(int M, int N1, int N2, int* tile, int* splits);
/* The kernel family avs_gemm_nt_bf16 dispatches (M, N, K) to under the current knobs, and the workgroups it launches: family 0 two-buffer
 * 128 x 128 tiles | 1 LDS-DMA ring, 128 x 128 | 2 ring, 64 x 128 half-height tiles | 3 two-buffer, 64 x 128 | 4 persistent 256 x 256 (8-phase).
 * Host arithmetic only (callable without a GPU). */
int avs_gemm_nt_plan(int M, int N, int K, int* family, int* workgroups);
/* up to three weight gradients over the SAME M token rows in one launch: Ci[N1_i, N2_i] (contiguous) += Ai^T . Bi; problem i is absent
 * when Ai is NULL (problem 0 must exist).  A block's fc2 / fc1 / proj gradients (timm Mlp + Attention.proj, cav_mae_base.py:77,138-143)
 * exist at the same time; together they fill the chip with 3 splits of the token rows instead of 7 + 7 + 14, i.e. with 40 % of the fp32
 * atomic traffic.  When every N is a multiple of 256 the 8-phase 256 x 256 kernel takes all tiles; otherwise one launch per problem. */
int avs_gemm_tn_bf16_group3(const avs_bf16* A0, long long lda0, const avs_bf16* B0, long long ldb0, float* C0, int N1_0, int N2_0,
                            const avs_bf16* A1, long long lda1, const avs_bf16* B1, long long ldb1, float* C1, int N1_1, int N2_1,
                            const avs_bf16* A2, long long lda2, const avs_bf16* B2, long long ldb2, float* C2, int N1_2, int N2_2,
                            int M, avs_stream_t stream);

/* the same three-problem launch on fp8 operands (fp8 mode 3): Ai = the e5m2 copy of the output gradient [M, N1_i], Bi = the e4m3 copy of
 * the layer input [M, N2_i] (both ZERO up to the next multiple of 64 rows; leading dimensions in bytes, multiples of 16), qai / qbi the two
 * operands' device records (the product is de-quantised with qa[1] * qb[1]); every N a multiple of 256.  v_mfma_f32_32x32x64_f8f6f4 on
 * fragments read with ds_read_b64_tr_b8; fp32 atomics into the gradient arena like the bf16 form. */
int avs_gemm_tn_fp8_group3(const uint8_t* A0, long long lda0, const uint8_t* B0, long long ldb0, float* C0, int N1_0, int N2_0, const float* qa0, const float* qb0,
                           const uint8_t* A1, long long lda1, const uint8_t* B1, long long ldb1, float* C1, int N1_1, int N2_1, const float* qa1, const float* qb1,
                           const uint8_t* A2, long long lda2, const uint8_t* B2, long long ldb2, float* C2, int N1_2, int N2_2, const float* qa2, const float* qb2,
                           int M, avs_stream_t stream);

/* ---- varlen attention (F.scaled_dot_product_attention in Attention.forward, cav_mae_base.py:60-68) on the packed
 * qkv matrix [rows, 3*D] (q|k|v, head h at columns h*hd); one (tile_start, tile_len, tile_q0) triple per tile of
 * tile_rows (128: 4-wave workgroups; 64: 2-wave workgroups, twice as many resident per CU - for short sequences) rows.
 * Head dims hd = D / H: 64 (ViT-B/L encoder), 32 (decoder), 80 (ViT-H encoder; tile_rows 128 only).  lse/delta: [H][rows_total] fp32.
 * Rows of qkv, out, dout and dqkv must be 16-byte aligned: ld and ldo multiples of 8 elements (the kernels load fragments and store
 * accumulator blocks 16 bytes per lane), ldo8 / ld8 of the 8-bit copies multiples of 16; other strides are refused (-2). */
int avs_attn_fwd(const avs_bf16* qkv, long long ld, int D, int H, const int* tile_start, const int* tile_len,
                 const int* tile_q0, int ntiles, int tile_rows, avs_bf16* out, long long ldo, float* lse, int rows_total,
                 avs_stream_t stream);
/* the same, also writing out8 = e4m3(clamp(out * q8[0], +-448)) [rows, D] / ldo8 - the fp8 operand of the proj GEMM in the fp8 mode -
 * and folding max |out| into q8[2] (device quantisation record, see avs_fp8_scale_update); out8 and q8 both NULL = avs_attn_fwd */
int avs_attn_fwd_q8(const avs_bf16* qkv, long long ld, int D, int H, const int* tile_start, const int* tile_len,
                    const int* tile_q0, int ntiles, int tile_rows, avs_bf16* out, long long ldo, float* lse, int rows_total,
                    uint8_t* out8, long long ldo8, float* q8, avs_stream_t stream);
int avs_attn_bwd(const avs_bf16* qkv, long long ld, int D, int H, const int* tile_start, const int* tile_len,
                 const int* tile_q0, int ntiles, int tile_rows, const avs_bf16* out, const avs_bf16* dout, long long ldo,
                 const float* lse, float* delta, int rows_total, avs_bf16* dqkv, avs_stream_t stream);
/* the same, also writing dqkv8 = e5m2(clamp(dqkv * qd8[0], +-57344)) [rows, 3*D] / ld8 - the gradient operand of the fp8 qkv input-gradient
 * GEMM (fp8 mode 2: every input-gradient GEMM of a block on e5m2 x e4m3) - and folding max |dqkv| into the device record qd8; both NULL =
 * avs_attn_bwd.  kv_bf16 = 0 (with dqkv8 only; fp8 mode 3): the key / value thirds of the bf16 dqkv are not written - the input- and
 * weight-gradient GEMMs read the e5m2 copy; the query third (column sum = bias gradient) always is */
int avs_attn_bwd_q8(const avs_bf16* qkv, long long ld, int D, int H, const int* tile_start, const int* tile_len,
                    const int* tile_q0, int ntiles, int tile_rows, const avs_bf16* out, const avs_bf16* dout, long long ldo,
                    const float* lse, float* delta, int rows_total, avs_bf16* dqkv, uint8_t* dqkv8, long long ld8, float* qd8,
                    int kv_bf16, avs_stream_t stream);
/* Pruned form for the LAST decoder block (forward_decoder + forward_mae_loss, cav_mae_base.py:629-635,679-682: rows with mask 0 contribute
 * neither loss nor gradient, so in the last block they are needed as keys / values only).  Every sequence of the launch has the same length
 * (tile_len); only its FIRST lq rows are queries, all rows are keys / values; `out` / `dout` are COMPACT [nseq * lq, D]: sequence s
 * (= tile_start / tile_len) owns rows s * lq ..; lse / delta / dqkv keep the packed row numbering.  The backward writes dq for the first lq
 * rows of every sequence only (the caller zeroes the query third of the other rows, avs_expand_rows_bf16) and dk / dv for all rows. */
int avs_attn_fwd_cq(const avs_bf16* qkv, long long ld, int D, int H, const int* tile_start, const int* tile_len,
                    const int* tile_q0, int ntiles, int tile_rows, avs_bf16* out, long long ldo, float* lse, int rows_total,
                    int lq, avs_stream_t stream);
int avs_attn_bwd_cq(const avs_bf16* qkv, long long ld, int D, int H, const int* tile_start, const int* tile_len,
                    const int* tile_q0, int ntiles, int tile_rows, const avs_bf16* out, const avs_bf16* dout, long long ldo,
                    const float* lse, float* delta, int rows_total, avs_bf16* dqkv, int lq, avs_stream_t stream);
/* the same backward for sequences of at most rows_per_wg (64 | 128) tokens, in ONE kernel: a workgroup per (sequence, head) reads q, k,
 * v, dO, o once, evaluates S and its exponentials once and writes dq, dk and dv (hd 32 | 64).  seq_start / seq_len: [nseq] */
int avs_attn_bwd_fused(const avs_bf16* qkv, long long ld, int D, int H, const int* seq_start, const int* seq_len, int nseq,
                       int rows_per_wg, const avs_bf16* out, const avs_bf16* dout, long long ldo, const float* lse, int rows_total,
                       avs_bf16* dqkv, avs_stream_t stream);
int avs_attn_bwd_fused_q8(const avs_bf16* qkv, long long ld, int D, int H, const int* seq_start, const int* seq_len, int nseq,
                          int rows_per_wg, const avs_bf16* out, const avs_bf16* dout, long long ldo, const float* lse, int rows_total,
                          avs_bf16* dqkv, uint8_t* dqkv8, long long ld8, float* qd8, int kv_bf16, avs_stream_t stream);

/* ---- input normalisation on the device (what the reference dataloader does per sample on the host: dataloader.py:505-513
 * fbank = (fbank - norm_mean) / norm_std [+ rand * amp, roll(shift) when `noise`]; :461-462,152-155 frame / 255 then
 * (x - mean_c) / std_c).  audio: in/out [B, T, F] fp32, out of place, F%4==0; shift/amp per sample or NULL; the noise is a
 * Philox stream keyed by seed.  frames: in [n_images, 3, plane] uint8 -> out fp32, plane = H*W, plane%4==0; mean3/std3
 * are HOST pointers to 3 floats. */
int avs_normalize_audio(const float* in, float* out, int B, int T, int F, float mean, float std, const int* shift,
                        const float* amp, unsigned long long seed, avs_stream_t stream);
int avs_normalize_frames_u8(const uint8_t* in, float* out, int n_images, int plane, const float* mean3, const float* std3,
                            avs_stream_t stream);

/* ---- raw inputs, fused.  The same dataset arithmetic as avs_normalize_* above, applied where the input
 * is READ - the patch gather of the embedding and the target gather of the reconstruction loss - so un-normalised fbank
 * and uint8 frames go straight from the loader's buffers into the model: one read of the raw tensor per consumer instead of
 * read + write + read.  kind 0 / NULL: the tensor is the normalised fp32 tensor of the reference's forward() contract.
 * kind 1 (audio [B, T, F] fp32): value(b,t,f) = (in[b, (t - shift_b) mod T, f] - mean[0]) / std[0] + amp_b * U(b, ts, f); shift / amp
 * are DEVICE arrays per sample or NULL (dataloader.py:510-513: the `noise` augmentation), U is the Philox stream of
 * avs_normalize_audio for the same seed, so both paths agree bit for bit.  kind 2 (frames [n, 3, H, W] uint8):
 * value = (in / 255 - mean[c]) / std[c] (:461-462, 152-155).  The struct itself is read on the HOST at call time. */
typedef struct avs_input_xf {
    int kind;
    float mean[3], std[3];
    const int* shift;
    const float* amp;
    unsigned long long seed;
} avs_input_xf;

/* ---- patch gather of the kept tokens (PatchEmbed input side + random_masking gather: cav_mae_base.py:96-99,
 * 382,431,444-455); the _xf forms take a raw input and its transform */
int avs_im2col_audio(const float* a, const int* row_b, const int* row_tok, avs_bf16* out, int rows, int tlen, int mel,
                     int t_patches, avs_stream_t stream);
int avs_im2col_video(const float* v, const int* row_img, const int* row_tok, avs_bf16* out, int rows, int C, int H, int W,
                     avs_stream_t stream);
int avs_im2col_audio_xf(const float* a, const int* row_b, const int* row_tok, avs_bf16* out, int rows, int tlen, int mel,
                        int t_patches, const avs_input_xf* xf, avs_stream_t stream);
int avs_im2col_video_xf(const void* v, const int* row_img, const int* row_tok, avs_bf16* out, int rows, int C, int H, int W,
                        const avs_input_xf* xf, avs_stream_t stream);
/* random masking on the device (random_masking_unstructured / _structured + the gather index build,
 * cav_mae_base.py:365-439): one workgroup per sequence; seqs = nseq x 16 int32 {L, keep, row_off, src_id, dec_off, enc_base,
 * t_patches, ids_off, mask_off, tmask_x, 0, 0, dec_m_off, dec_k_off, pred_off, pos_base} (the last four: avs_mask_plan_grouped; dec_m_off
 * must be -1 for the two entry points below); L <= 1024.  tmask_lo/hi, fmask (per sequence bit masks of the time columns /
 * frequency rows forced to be removed) are read only where t_patches > 0.  src_row / mask_out / ids_out may be NULL when no
 * sequence uses them. */
int avs_mask_plan(const int* seqs, int nseq, const unsigned* tmask_lo, const unsigned* tmask_hi, const unsigned* fmask,
                  unsigned long long seed, int* row_src, int* row_tok, int* src_row, float* mask_out, int* ids_out,
                  avs_stream_t stream);
/* the same with the Philox key in device memory, read when the kernel runs (a training step replayed from a captured hipGraph
 * freezes kernel arguments; graph_step.GraphedTrainStep advances the key with a node of the graph) */
int avs_mask_plan_dev(const int* seqs, int nseq, const unsigned* tmask_lo, const unsigned* tmask_hi, const unsigned* fmask,
                      const unsigned long long* seed_dev, int* row_src, int* row_tok, int* src_row, float* mask_out, int* ids_out,
                      avs_stream_t stream);
/* GROUPED decoder layout: for the sequences whose descriptor has dec_m_off >= 0 the decoder rows of a sample are ordered [tokens whose
 * prediction is scored (mask 1) | kept tokens] instead of by position (forward_decoder's un-shuffle is a permutation of tokens that the
 * attention blocks are equivariant to, cav_mae_base.py:604-626): src_row[dec row] = encoder row | -1 with dec row = dec_k_off + j (kept) |
 * dec_m_off + j - keep (masked); pos_row[dec row] = pos_base + token (the positional embedding it takes); row_of_pos[dec_off + token] =
 * dec row; pred_id[pred_off + j - keep] = mask_off + token (the (sample, token) a compact prediction row scores).  seed_dev may be NULL. */
int avs_mask_plan_grouped(const int* seqs, int nseq, const unsigned* tmask_lo, const unsigned* tmask_hi, const unsigned* fmask,
                          unsigned long long seed, const unsigned long long* seed_dev, int* row_src, int* row_tok, int* src_row,
                          float* mask_out, int* ids_out, int* pos_row, int* row_of_pos, int* pred_id, avs_stream_t stream);
int avs_cast_scale_bf16(const float* x, avs_bf16* y, long long n, float alpha, avs_stream_t stream);
/* out[r, 0:cols] = src_row[r] >= 0 ? in[src_row[r], 0:cols] : 0 (bf16, cols % 8 == 0, leading dimensions in elements, % 8 == 0).
 * in == NULL: only the rows with src_row[r] < 0 are written (zeroed).  No reference counterpart: bookkeeping of the pruned last decoder block */
int avs_expand_rows_bf16(const avs_bf16* in, long long ld_in, const int* src_row, avs_bf16* out, long long ld_out, int rows, int cols,
                         avs_stream_t stream);
int avs_scatter_add_rows(const avs_bf16* src, const int* idx, float* dst, int rows, int D, float scale, avs_stream_t stream);
/* out[c] += sum over rows of x[r][c], c < C (C % 64 == 0); ld: leading dimension of x (a column range of a wider matrix is allowed) */
int avs_colsum_bf16(const avs_bf16* x, long long ld, float* out, int rows, int C, avs_stream_t stream);
/* y[n] += alpha * sum_k x[k] * W[k][n]  (x fp32 [K], W bf16 [K, N] / ld, N % 256 == 0, K % 32 == 0): the value third of the qkv bias
 * gradient = (proj bias gradient) . W_proj, because softmax rows sum to one (Attention, cav_mae_base.py:51,60-77); the key third is 0 */
int avs_vecmat_bf16(const float* x, const avs_bf16* W, long long ld, float* y, int K, int N, float alpha, avs_stream_t stream);
/* n such products of one shape in one launch: desc (device, int64 [n][3]) = {x, W, y} pointers (a stack's blocks at the end of its backward) */
int avs_vecmat_bf16_batched(const long long* desc, int n, long long ld, int K, int N, float alpha, avs_stream_t stream);

/* ---- decoder un-shuffle (forward_decoder, cav_mae_base.py:604-626) */
int avs_unshuffle_fwd(const float* x, const int* src_row, const int* pos_row, const uint8_t* row_mod,
                      const float* mask_token, const float* pos_a, const float* pos_v, int La, const float* mod_a,
                      const float* mod_v, float* out, int rows, int D, avs_stream_t stream);
int avs_unshuffle_bwd(const float* dout, const int* src_row, int B, int T, int La, int Lv, float* dx, float* dpos_a,
                      float* dpos_v, float* dmask, float* dmod_a, float* dmod_v, int D, avs_stream_t stream);
/* the same when the decoder rows are not in position order (avs_mask_plan_grouped): row_of_pos[b * (La + T * Lv) + position] = decoder row */
int avs_unshuffle_bwd_map(const float* dout, const int* src_row, int B, int T, int La, int Lv, float* dx, float* dpos_a,
                          float* dpos_v, float* dmask, float* dmod_a, float* dmod_v, int D, const int* row_of_pos, avs_stream_t stream);

/* ---- token mean per packed sequence (.mean(dim=1), cav_mae_base.py:563,566).  row_map (may be NULL): segment s is row row_map[s]
 * of reps / dreps - the mixed encoder's inverse permutation (:584-590) folded into the reduction */
int avs_segment_mean_fwd(const float* y, const int* seg_start, float* reps, int nseg, int D, const int* row_map, avs_stream_t stream);
int avs_segment_mean_bwd(const float* dreps, const int* seg_start, float* dy, int nseg, int D, float scale, const int* row_map,
                         avs_stream_t stream);

/* ---- masked-MSE with patchify on the fly (patchify + forward_mae_loss, cav_mae_base.py:343-351,663-683) */
/* loss[0] = masked mean; total (may be NULL): total[0] = (total_init ? 0 : total[0]) + loss[0]  (loss_mae = a + v, :707) */
int avs_mae_loss_fwd(const float* pred, const float* inp, const float* mask, float* row_loss, float* loss, float* total,
                     int total_init, int rows, int audio, int L, int C, int H, int W, float nmask, avs_stream_t stream);
int avs_mae_loss_bwd(const float* pred, const float* inp, const float* mask, const float* gout, avs_bf16* dpred, int rows,
                     int audio, int L, int C, int H, int W, float nmask, avs_stream_t stream);
/* the same with the target read from a raw input (see avs_input_xf) */
int avs_mae_loss_fwd_xf(const float* pred, const void* inp, const float* mask, float* row_loss, float* loss, float* total,
                        int total_init, int rows, int audio, int L, int C, int H, int W, float nmask, const avs_input_xf* xf,
                        avs_stream_t stream);
int avs_mae_loss_bwd_xf(const float* pred, const void* inp, const float* mask, const float* gout, avs_bf16* dpred, int rows,
                        int audio, int L, int C, int H, int W, float nmask, const avs_input_xf* xf, avs_stream_t stream);
/* the same four with a patch STRIDE (1..16) on 16 x 16 patch storage: token (t, f) / (gy, gx) starts at pixel t*stride / gy*stride, only
 * the stride x stride corner of the 256 positions per channel is gathered (the rest of the im2col row is zero) / scored (mean over the
 * scored elements; dpred is zero elsewhere).  stride 16 = the entry points above.  14: the patch grid of ViT-H/14 (config.stride). */
int avs_im2col_audio_s(const float* a, const int* row_b, const int* row_tok, avs_bf16* out, int rows, int tlen, int mel, int t_patches,
                       int stride, const avs_input_xf* xf, avs_stream_t stream);
int avs_im2col_video_s(const void* v, const int* row_img, const int* row_tok, avs_bf16* out, int rows, int C, int H, int W, int stride,
                       const avs_input_xf* xf, avs_stream_t stream);
int avs_mae_loss_fwd_s(const float* pred, const void* inp, const float* mask, float* row_loss, float* loss, float* total,
                       int total_init, int rows, int audio, int L, int C, int H, int W, float nmask, int stride, const avs_input_xf* xf,
                       avs_stream_t stream);
int avs_mae_loss_bwd_s(const float* pred, const void* inp, const float* mask, const float* gout, avs_bf16* dpred, int rows, int audio,
                       int L, int C, int H, int W, float nmask, int stride, const avs_input_xf* xf, avs_stream_t stream);
/* the same two on COMPACT predictions: only the rows whose mask is 1 exist; prediction row r scores (sample, token) row_id[r] - id_base of
 * the [N * L] numbering mask and targets use (row_id: avs_mask_plan_grouped's pred_id; NULL = the two entry points above) */
int avs_mae_loss_fwd_id(const float* pred, const void* inp, const float* mask, float* row_loss, float* loss, float* total,
                        int total_init, int rows, int audio, int L, int C, int H, int W, float nmask, int stride, const avs_input_xf* xf,
                        const int* row_id, int id_base, avs_stream_t stream);
int avs_mae_loss_bwd_id(const float* pred, const void* inp, const float* mask, const float* gout, avs_bf16* dpred, int rows, int audio,
                        int L, int C, int H, int W, float nmask, int stride, const avs_input_xf* xf, const int* row_id, int id_base,
                        avs_stream_t stream);


/* ---- bidirectional InfoNCE (forward_contrastive, cav_mae_base.py:641-661) */
int avs_l2norm_fwd(const float* x, float* xn, float* norm, int rows, int D, avs_stream_t stream);
int avs_l2norm_bwd(const float* dxn, const float* xn, const float* norm, float* dx, int rows, int D, float scale,
                   avs_stream_t stream);
int avs_gemm_f32_small(const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn, float* C,
                       long long scm, int M, int N, int K, float alpha, avs_stream_t stream);
/* out = {nce, c_acc, weight * nce} */
int avs_infonce_fwd(const float* total, float* stats, float* out, int N, float weight, avs_stream_t stream);
int avs_infonce_dlogits(const float* total, const float* stats, const float* gout, float weight, float* dtotal, int N,
                        avs_stream_t stream);

/* ---- weights: bf16 shadow copies and the fused Adam step (torch.optim.Adam as built at
 * src/traintest_cavmae_base.py:64-66) */
int avs_transpose_bf16(const avs_bf16* in, avs_bf16* out, int R, int C, avs_stream_t stream);
/* one launch for many matrices: desc = nmat x {src, dst, R, C, first_tile, tiles_per_row} (int64), tile_map[b] = matrix of
 * workgroup b, ntiles = sum of ceil(R/64)*ceil(C/64) */
int avs_transpose_batched(const long long* desc, const int* tile_map, int ntiles, avs_stream_t stream);
int avs_cast_bf16(const float* x, avs_bf16* y, long long n, avs_stream_t stream);
int avs_adam(float* p, const float* g, float* m, float* v, avs_bf16* p_bf16, long long n, float lr, float beta1,
             float beta2, float eps, float weight_decay, int step, float grad_scale, avs_stream_t stream);
/* the same update with the step count (>= 1, the count of THIS update) read from device memory when the kernel runs - for a step
 * replayed from a captured hipGraph; bias corrections in double as above */
int avs_adam_dev(float* p, const float* g, float* m, float* v, avs_bf16* p_bf16, long long n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, const int* step_dev, float grad_scale, avs_stream_t stream);

/* ---- Collectives of the data-parallel path: thin wrappers over RCCL on a stream of the communicator's own, with event hand-off
 * (SURVEY.md 8(b)).  Replace, on the data path, torch.distributed's all_gather / all_reduce in GatherLayer
 * (src/models/gather_layer.py:21-37) and DistributedDataParallel's bucketed gradient all-reduce
 * (src/traintest_cavmae_base.py:58-59).  RCCL is resolved at run time (the copy PyTorch loaded, else the system's).
 *  avs_comm_unique_id  rank 0: the 128-byte rendezvous id, to be handed to the other ranks over any host channel
 *  avs_comm_init       collective over all ranks (returns when every rank has called it); uses the current device
 *  avs_allreduce       buf[count] <- SUM over ranks, in place          dtype 0 = fp32, 1 = bf16
 *  avs_allgather       out[world * count] <- every rank's in[count], rank-major
 *  avs_reducescatter   out[count] <- SUM over ranks of their in[rank * count ...]
 *     each is ordered behind the work already queued on `after`, runs on the communicator's stream and returns at once
 *  avs_comm_wait       `stream` waits on the device for every collective issued so far */
int avs_comm_unique_id(void* id128);
int avs_comm_init(const void* id128, int rank, int world, void** comm);
int avs_comm_destroy(void* comm);
int avs_comm_rank(void* comm);
int avs_comm_world(void* comm);
int avs_allreduce(void* comm, void* buf, unsigned long long count, int dtype, avs_stream_t after);
int avs_allgather(void* comm, const void* in, void* out, unsigned long long count, int dtype, avs_stream_t after);
int avs_reducescatter(void* comm, const void* in, void* out, unsigned long long count, int dtype, avs_stream_t after);
int avs_comm_wait(void* comm, avs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
