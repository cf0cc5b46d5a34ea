/* vc2.h -- C ABI of the MI355X-native VidCom2 token-compression hot path (libvc2hip.so).
 *
 * The reference (xuyang-liu16/VidCom2) has no FFI: its boundary is the Python module
 * token_compressor/vidcom2/vidcom2.py.  Each entry point below names the reference function
 * (file:line) whose tensor work it replaces; vidcom2_amd/vidcom2.py binds them with ctypes
 * and re-exposes the reference's Python names and signatures.
 *
 * Conventions
 *   - plain pointers + sizes only; no torch / C++ types cross this boundary
 *   - every pointer is a DEVICE pointer unless the name says host; `stream` is a hipStream_t
 *     passed as void* (NULL = the legacy default stream)
 *   - functions only ENQUEUE work on `stream`; none of them synchronises, allocates or frees
 *     device memory (the caller passes a workspace sized by vc2_workspace_bytes)
 *   - return 0 on success, a negative VC2_ERR_* otherwise (nothing was enqueued on error);
 *     vc2_last_error() gives a thread-local message
 *   - dtype codes: 0 = fp32, 1 = bf16, 2 = fp16.  All arithmetic follows the reference's
 *     "every op in the input dtype" behaviour: each torch op is evaluated exactly and rounded
 *     fp32 -> T (DESIGN.md "Numerics contract"); index outputs are int64 like torch's
 *   - row-major contiguous tensors; 16-byte aligned base pointers
 */
#ifndef VC2_H_
#define VC2_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VC2_F32 0
#define VC2_BF16 1
#define VC2_F16 2

#define VC2_OK 0
#define VC2_ERR_ARG (-1)          /* bad argument (null pointer, non-positive size, bad dtype) */
#define VC2_ERR_SHAPE (-2)        /* rows not divisible by tokens-per-frame (torch .view RuntimeError) */
#define VC2_ERR_UNSUPPORTED (-3)  /* shape outside what the kernels cover (see vc2_limits) */
#define VC2_ERR_LAUNCH (-4)       /* hipLaunch failure; message in vc2_last_error() */
#define VC2_ERR_WORKSPACE (-5)    /* workspace too small */

#define VC2_MAP_LINEAR 0          /* vidcom2.py:99-103  _map_linear_offset */
#define VC2_MAP_GRID_VID 1        /* vidcom2.py:105-115 _map_grid_vid      */
#define VC2_MAP_LOCAL 2           /* per-frame local indices (select_outlier_indices output) */

const char* vc2_last_error(void);
const char* vc2_version(void);

/* Accumulation semantics of the reference's fp32-accumulated reductions in half precision (token L2 norm,
 * squared-distance row sums, centre means):
 *   4 (default)  "torch order": wherever the exactly computed value lies within a few fp32-ulps (128 / 48) of a T rounding
 *                boundary, torch's own CPU accumulation order (8 interleaved fp32 chains over the variance-sorted
 *                channels / cascade sum) is replayed for that token -> bit-exact to the CPU reference.  FRAME-centre means
 *                (round 6): the model shapes -- 16-bit inputs, D = 1024 / 3584 / 4096, frames of <= 512 tokens, every frame
 *                workgroup resident -- add every frame's x^ in torch's OWN order inside sweep 2 (16-row blocks, fp32, the
 *                outer-sum cascade of SumKernel.cpp): the frame means are the reference's bits by construction, no margin,
 *                no replay (environment VC2_S2_ORD=0: the form below).  Other shapes, and the VIDEO-centre mean everywhere:
 *                exact (fp64) sums, and a mean within 16 fp32-ulps of a T boundary is replayed in torch's order; for frame
 *                means the margin has, besides the 16 ulps, a term relative to sum |x^| over the frame (4 u A / n, A bounded
 *                from sweep 1's statistics and the frame's smallest denominator): it grows under cancellation, where torch's
 *                cascade errs by far more than ulps OF THE MEAN.  Reproduces the reference on every fixture (incl. the
 *                adversarial `cancel` ones) and every soak case.
 *   0            "exact": every reduction correctly rounded (DESIGN.md "Numerics contract").
 *   3            "proven": a PROVEN bound decides which centre means are replayed (forward error bound of torch's
 *                cascade relative to sum |x^|, bounded from sweep 1's statistics): flags 50x more means, costs ~45 %
 *                more time (256 against 178 us per pass at the target shape, round 5; with VC2_S2_ORD=0); the test-suite runs every fixture
 *                in it as well and asserts the same results.
 *   1            "fast" (the default of rounds 1-3, now opt-in): the 16-ulp margin alone for all centre means.  ~1 %
 *                faster than mode 4 (176 against 178 us, round 5); NOT bit-exact under cancellation (0.2 % of random `cancel` inputs differ in last-bit
 *                f scores, rarely a kept index) -- no claim of reference parity is made for it.
 *   (2: debug -- every value is replayed.)
 * fp32 inputs are unaffected.  vc2_set_mode is PROCESS-WIDE (default 4) and also drops the calling thread's own
 * override; vc2_set_thread_mode(mode) overrides it for the calling thread only (-1: follow the process-wide setting
 * again), so that a worker thread follows the application's choice unless it asks otherwise; vc2_get_mode returns
 * what a pass issued by the calling thread would use.  (All ranks / threads of a frame-sharded pass must use the
 * same mode: it decides which exchanges take place.) */
int vc2_set_mode(int mode);
int vc2_set_thread_mode(int mode);
int vc2_get_mode(void);
/* Workspace (bytes) needed by any entry point below for an [F*N, D] input. */
int vc2_workspace_bytes(int64_t F, int64_t N, int64_t D, int dtype, size_t* out_bytes);

/* Upper bound on the number of kept tokens sum(ks) for a given base_scale: lets the caller
 * allocate the output of vc2_compress before the budgets are known (the reference learns K
 * from a host sync, vidcom2.py:72). */
int64_t vc2_kept_capacity(int64_t F, int64_t N, double base_scale);

/* ---- stage entry points ------------------------------------------------------------ */

/* vidcom2.py:40  x.var(dim=0, unbiased=False) -> var T[D] (also fp32-widened copy var_f32[D],
 * may be NULL).  Sweep 1 of X. */
int vc2_chan_var(const void* x, int64_t R, int64_t D, int dtype, void* ws, size_t ws_bytes,
                 void* var_T, float* var_f32, void* stream);

/* vidcom2.py:41-42  torch.topk(var, k, largest=False): the SET of the k selected channels, with ties
 * at the k-th value broken exactly like the CPU reference (libstdc++ introselect; SURVEY.md
 * Appendix A).  var_f32 = widened T values.  Outputs (each may be NULL): byte mask mask[D]
 * (1 = selected); the ascending channel list cols[k] the scoring sweeps consume; perm[k] = the selected channels
 * in the order nth_element / partial_sort left them (the input of the ORDER replay below).
 * With order / opos / spos (need perm; opos / spos also need cols) a second kernel replays torch.topk(sorted=True)'s
 * OWN order: order[k] = ascending variance, libstdc++ sort tie order -- the column order of `x[:, topk_idx]`,
 * vidcom2.py:43; opos[k] = position of order[p] inside cols; spos[k] = its inverse (position of cols[i] in that
 * order: what the "torch order" replays of the scoring sweeps need).  The fused pass and vc2_scores_phase1 do not
 * call this second kernel: they attach the same job to sweep 2 as a rider workgroup. */
int vc2_chan_select(const float* var_f32, int64_t D, int64_t k, uint8_t* mask, int32_t* cols, int32_t* perm,
                    int32_t* order, int32_t* opos, int32_t* spos, void* stream);

/* vidcom2.py:43  x[:, idx] column gather -> out T[R, C]; idx int64[C] on device. */
int vc2_gather_cols(const void* x, int64_t R, int64_t D, int dtype, const int64_t* idx, int64_t C,
                    void* out, void* stream);

/* vidcom2.py:45-62  compute_gaussian_scores over the C channels listed (ascending) in cols[C]
 * (cols NULL = all channels, C == D, i.e. x already holds the selected features, already in torch.topk
 * order).  spos = vc2_chan_select's output (needed for mode 1 when cols != NULL).  Sweeps 2 and 3 of X.
 * Outputs: v_T, f_T  T[F,N] (may be NULL), total_f32 fp32-widened RN_T(v+f) [F,N] (vidcom2.py:33),
 * s_f32[F] = -mean(v, -1) widened (vidcom2.py:32).
 * Cost note: in modes 3 and 4 (4 is the default) the frame-mean replay margins are bounded from the channel statistics
 * of x, so this stage call runs the statistics sweep (sweep 1) over x itself before sweeps 2 and 3 -- three sweeps, where
 * modes 0 / 1 take two.  The one-launch pass (vc2_compress*) and vc2_scores_phase1 behind vc2_chan_stats reuse sweep 1's
 * partials instead. */
int vc2_scores(const void* x, int64_t F, int64_t N, int64_t D, int dtype, const int32_t* cols,
               int64_t C, const int32_t* spos, void* ws, size_t ws_bytes, void* v_T, void* f_T,
               float* total_f32, float* s_f32, void* stream);

/* vidcom2.py:64-68  compute_scales(scores, base, temp) on T[F] -> scales T[F]. */
int vc2_compute_scales(const void* s_T, int64_t F, double base, double temp, int dtype, void* ws,
                       size_t ws_bytes, void* scales_T, void* stream);

/* vidcom2.py:72-77  ks = (scales*tpf).round().long().clamp(min=1); per frame the k_f smallest
 * total scores (CPU-reference tie-breaking), ascending.  map_mode selects the index mapping
 * fused into the output (VC2_MAP_*; grid_h only for GRID_VID).  Outputs: ks int64[F],
 * (tpf = the reference's multiplier, normally N; a k_f > N is reported raw in ks while min(k_f, N) tokens are
 * selected -- the caller raises torch.topk's "k out of range" error),
 * offs int64[F+1] (exclusive prefix of min(ks, N)), idx_out int64[cap] and K_out[0] = number of indices
 * written.  K_out[1] = status bits, all zero on a sound pass: 1 = the count would have exceeded `cap` (nothing past
 * cap is written), 2 = a bounded wait between workgroups of one launch of this pass expired, 4 = a loop bound of the
 * selection replay expired IN THIS PASS (any of its kernels: channel selection, ORDER riders, per-frame selection; the
 * one-launch pass keeps the bit in its own workspace word, so two passes in flight on two streams cannot report each
 * other's hits; vc2_selftest_counters has the process-wide details); the Python mirror raises on any of them.  Stage
 * calls (this function) report their own launch only: K_out[1] is zeroed before the launch and every workgroup that
 * sees a hit ORs it in (a hit of ANY frame's selection is reported, whichever workgroup wrote the count). */
int vc2_select(const void* scores_T, const void* scales_T, int64_t F, int64_t N, int64_t tpf, int dtype,
               int map_mode, int64_t grid_h, void* ws, size_t ws_bytes, int64_t* ks, int64_t* offs,
               int64_t* idx_out, int64_t cap, int64_t* K_out, void* stream);

/* vidcom2.py:99-103 / :105-115 as standalone mappers: local per-frame indices (concatenated,
 * with ks/offs) -> global indices. */
int vc2_map_indices(const int64_t* local_idx, const int64_t* ks, const int64_t* offs, int64_t F,
                    int map_mode, int64_t stride_or_h, int64_t* out, void* stream);

/* vidcom2.py:91 / :96  flat[global_idx]: dst[j,:] = src[idx[j],:] for j < K_dev[0] (K read on
 * device; at most `cap` rows). */
int vc2_gather_rows(const void* src, int64_t src_rows, int64_t D, int dtype, const int64_t* idx,
                    const int64_t* K_dev, int64_t cap, void* dst, void* stream);

/* ---- one-shot hot path -------------------------------------------------------------- */

/* vidcom2.py:15-36  vidcom2_compression for the "linear" mapper (llava_ov / qwen*), and for
 * "grid_vid" when gather_src/map_mode say so: everything from the feature tensor resident in HBM
 * to kept rows + indices + budgets, enqueued back-to-back with no host round trip.
 *   x           T[F*N, D]
 *   gather_src  rows to gather from (x itself for linear; img_feat for grid_vid), gather_rows rows
 *   out_rows    T[cap, D]; idx_out int64[cap]; ks int64[F]; K_out int64[2] (see vc2_select)
 *   v_T/f_T     optional T[F,N] score outputs (NULL to skip)
 */
int vc2_compress(const void* x, int64_t F, int64_t N, int64_t D, int dtype, double base_scale,
                 int map_mode, int64_t grid_h, const void* gather_src, int64_t gather_rows,
                 void* ws, size_t ws_bytes, void* out_rows, int64_t* idx_out, int64_t cap,
                 int64_t* ks, int64_t* K_out, void* v_T, void* f_T, void* stream);

/* The same pass with
 *  - `tail_rows` extra rows (T[tail_rows, D], e.g. LLaVA's image_newline embedding, reference
 *    models/llava.py:160-168) written right behind the K kept rows by the gather launch itself: out_rows is
 *    T[cap + tail_rows, D] and rows [0, K + tail_rows) are valid afterwards;
 *  - flags & VC2_FLAG_HAVE_STATS: sweep 1 is skipped -- ws already holds the channel-statistics partials of x,
 *    left there by vc2_pool_stats on the same stream (same F, N, D, dtype). */
#define VC2_FLAG_HAVE_STATS 1
int vc2_compress_ex(const void* x, int64_t F, int64_t N, int64_t D, int dtype, double base_scale,
                      int map_mode, int64_t grid_h, const void* gather_src, int64_t gather_rows,
                      void* ws, size_t ws_bytes, void* out_rows, int64_t* idx_out, int64_t cap,
                      int64_t* ks, int64_t* K_out, void* v_T, void* f_T, const void* tail, int64_t tail_rows,
                      int flags, void* stream);

/* vc2_compress_ex with a HOST MIRROR of the count: K_host (NULL: none) points at FOUR int64 words of pinned, device-mapped
 * host memory (hipHostMalloc / torch's pin_memory).  The selection launch writes K_host[1] = the status known so far and
 * then K_host[0] = K_out[0] there, BEFORE the gather launch runs: a caller that sets K_host[0] = -1 beforehand and spins
 * in vc2_wait_host_count has the one number it needs on the host (the reference's `.tolist()`, vidcom2.py:72) ~15 us
 * before the pass ends and without a device-to-host copy behind it; everything it does with the outputs afterwards is
 * ordered by the stream.  K_host[1] is the EARLY status (capacity, expired waits, guard hits of the earlier kernels); the
 * last selection workgroup to finish writes K_host[2] = the launch's final status | 2^62 (set K_host[2] = 0 beforehand):
 * a caller that went on with the early words must look at K_host[2] before it relies on the kept set being what the
 * reference keeps (bit 4, a loop bound of the selection replay expired: "cannot happen", never swallowed).
 * vc2_wait_host_count returns the count, or a negative number after timeout_s seconds. */
int vc2_compress_ex2(const void* x, int64_t F, int64_t N, int64_t D, int dtype, double base_scale,
                     int map_mode, int64_t grid_h, const void* gather_src, int64_t gather_rows,
                     void* ws, size_t ws_bytes, void* out_rows, int64_t* idx_out, int64_t cap, int64_t* ks,
                     int64_t* K_out, void* v_T, void* f_T, const void* tail, int64_t tail_rows, int flags,
                     int64_t* K_host, void* stream);
int64_t vc2_wait_host_count(const int64_t* K_host, double timeout_s);

/* ---- upstream fusion: LLaVA's get_2dPool (llava/model/llava_arch.py:171-190) + sweep 1 -------
 * xin T[F, H*W, D] (the projector output, token-major) -> x_out T[F, h*w, D] = its 2x2 pool, and the sweep-1
 * channel-statistics partials of x_out in ws (workspace of vc2_workspace_bytes(F, h*w, D)): the pooled tensor is
 * written once and vc2_compress_ex(..., VC2_FLAG_HAVE_STATS) reads it only twice more.
 *   VC2_POOL_AVG       avg_pool2d(kernel 2, stride 2): h = H/2, w = W/2; bit-exact to torch (fp32 sum in window order, / 4)
 *   VC2_POOL_MAX       max_pool2d(2): bit-exact (NaN propagates like torch)
 *   VC2_POOL_BILINEAR  interpolate(size=ceil(H/2) x ceil(W/2), mode="bilinear", align_corners=False): bit-exact to
 *                      torch 2.10's x86 CPU kernel (ATen's source index / lambda arithmetic and the fma order of its
 *                      vectorised channel loop); needs D % 8 == 0 (fp32) / D % 16 == 0 (16-bit), else
 *                      VC2_ERR_UNSUPPORTED (the loop's scalar tail contracts differently). */
#define VC2_POOL_AVG 1
#define VC2_POOL_MAX 2
#define VC2_POOL_BILINEAR 3
int vc2_pool_out_tokens(int64_t H, int64_t W, int mode, int64_t* h_out, int64_t* w_out);
int vc2_pool_stats(const void* xin, int64_t F, int64_t H, int64_t W, int64_t D, int dtype, int mode, void* ws,
                   size_t ws_bytes, void* x_out, void* stream);

/* ---- hook-side fusion: kept rows written once, at their final positions ------------
 * vc2_gather_scatter: up to 8 tensors T[src_rows[t], D] share ONE index list:
 *     dsts[t][dst_pos ? dst_pos[j] : dst_row0 + j] = srcs[t][idx ? idx[j] : j]      for j < n
 * n = min(n_dev[0], n_max) when n_dev (device) is given, else n_max; `tail_rows` rows of `tail` are appended to dsts[0]
 * behind the gathered ones (dst_row0 + n ...).  srcs / dsts / src_rows / dst_rows are HOST arrays of n_src entries.
 * Replaces flat[global_idx] + torch.cat(newline) (vidcom2.py:91, models/llava.py:160-168), inputs_embeds[:, keep]
 * (models/qwen2_5_vl.py:162-182) and the N+1 gathers of Qwen3-VL's deep-stack (models/qwen3_vl.py:140-165).
 * status (optional device word): |= 2 if a row index fell outside its tensor (such rows are skipped).
 * vc2_keep_positions (models/qwen2_5_vl.py:153-160): keep_out = ascending positions s in [0, S) with
 * !video_mask[s] or (ordinal of s among the video positions) in kept[0..K) (ascending); K = min(K_dev[0], K_max) or
 * K_max; vis_rows_out (optional, needs visual_mask) = ordinals, among the positions flagged in visual_mask, of the
 * kept ones.  keep_cap / vis_cap = entries the two outputs hold: nothing is written past them.  counts_out (optional,
 * device int64[3]) = {positions found, visual rows found, error bits}: 1 = more positions than keep_cap, 2 = more
 * visual rows than vis_cap, 4 = kept[] not strictly ascending inside [0, video positions), 8 = fewer positions than
 * keep_cap (the rest of keep_out is then filled with -1, which vc2_gather_scatter reports instead of reading). */
int vc2_gather_scatter(const void* const* srcs, const int64_t* src_rows, void* const* dsts, const int64_t* dst_rows,
                       int n_src, int64_t D, int dtype, const int64_t* idx, const int64_t* n_dev, int64_t n_max,
                       const int64_t* dst_pos, int64_t dst_row0, const void* tail, int64_t tail_rows,
                       int32_t* status, void* stream);
int vc2_keep_positions(const uint8_t* video_mask, int64_t S, const int64_t* kept, const int64_t* K_dev, int64_t K_max,
                       const uint8_t* visual_mask, int64_t* keep_out, int64_t keep_cap, int64_t* vis_rows_out,
                       int64_t vis_cap, int64_t* counts_out, void* stream);

/* ---- frame-sharded multi-GPU building blocks -------------------------
 * A rank holds frames [f0, f0+F_local) of a video with F_total frames.  Between the local sweeps the Python
 * layer all-gathers three small fp64 / fp32 arrays over RCCL.  Exchanges 1 and 2 carry CANONICAL partials -- per
 * stat block of 8 frames (mean, M2), per group of 16 frames the sums of the normalised tokens -- that every rank then
 * folds in the same fixed order, so the reduced bits do not depend on the world size (ranks holding a multiple of 16
 * frames; otherwise equal up to fp64 rounding):
 *   1. per stat block (mean, M2)   bstats[nb][2][D], nb = F_local / block_frames     -> vc2_chan_var_from_stats
 *   2. per 16-frame group: sums of x^, then bounds of sum |x^|   csum_parts[2 * ceil(F_local/16)][C]   -> vc2_scores_phase2
 *      (the bounds size the video centre's replay margin soundly, see mean_delta in the kernels)
 *   3. per-frame uniqueness   s[F_local]                                               -> vc2_select_sharded
 * F_total (the WHOLE video's frame count) fixes how frames are cut into row groups / splits on every rank. */
int vc2_stat_block_frames(void);     /* 8: the unsharded pass's stat block */
/* block_frames: frames per stat block, 1..8; use 8 when F_local is a multiple of 8 (bit-identical to the unsharded
 * pass), else the largest power of two dividing F_local -- the same value on every rank. */
int vc2_chan_stats(const void* x, int64_t F, int64_t N, int64_t D, int dtype, int64_t F_total, int block_frames,
                   void* ws, size_t ws_bytes, double* bstats /*[F / block_frames][2][D]*/, void* stream);
/* bstats of ALL ranks concatenated in rank order; rows_per_block = block_frames * N (the last block holds the
 * remainder). */
int vc2_chan_var_from_stats(const double* bstats /*[NB][2][D]*/, int64_t NB, int64_t rows_per_block,
                            int64_t R_total, int64_t D, int dtype, void* var_T, float* var_f32, void* stream);
/* perm != NULL (with var_f32 = the widened variances vc2_chan_select consumed): in mode 1 spos[C] is PRODUCED by a
 * rider workgroup of sweep 2 (the ORDER replay of vc2_chan_select, off the critical path) and then used by the
 * fix-up kernels; perm == NULL: spos must already be valid (or NULL: mode 0 / cols == NULL). */
int vc2_scores_phase1(const void* x, int64_t F, int64_t N, int64_t D, int dtype, const int32_t* cols,
                      int64_t C, int32_t* spos, const int32_t* perm, const float* var_f32, int64_t F_total,
                      void* ws, size_t ws_bytes, double* csum_parts /*[2 * ceil(F/16)][C]*/, void* stream);
/* csum_all: the ranks' csum_parts in rank order, P rows in all; rows_per_rank = rows every rank contributed
 * (2 * ceil(F_local/16): sums, then bounds), or 0 for sums only (then |x^| <= 1 bounds the margin: sound, but many
 * more columns are treated as boundary-near). */
int vc2_scores_phase2(const void* x, int64_t F, int64_t N, int64_t D, int dtype, const int32_t* cols,
                      int64_t C, const int32_t* spos, const double* csum_all /*[P][csum_stride]*/, int64_t P,
                      int64_t csum_stride, int64_t rows_per_rank, int64_t R_total, void* ws, size_t ws_bytes, void* v_T,
                      void* f_T, float* total_f32, float* s_f32, void* stream);
/* Video-centre replay of the frame-sharded pass ("torch order" mode, 16-bit inputs, every rank the same number of
 * rows and at least B of them, B = 16 rows up to 2^19 tokens per video, 32 up to 2^23, 64 up to 2^25 -- torch's
 * level_power 4 / 5 / 6).  Between exchange 2 and phase 2:
 *   vc2_video_centre_blocks  flags the boundary-near video-centre columns (identically on every rank) and writes, for
 *                            the first `cap` of them, a record of 129 + F*N/16 floats to blocks_out[cap][129 + F*N/16]:
 *                            [0,64) the x^ values of this rank's first rows that end a B-row block begun by the previous
 *                            rank (row0 = the video row index of this rank's first row tells how many), [64,128) those
 *                            of its last rows that begin a block the next rank ends, [128,..) the sums of its complete
 *                            blocks (rows added in order: torch's SumKernel cascade, level 0)
 *   (all-gather -> blocks_all[world][cap][129 + F*N/16], rank order)
 *   vc2_scores_phase2_blocks = vc2_scores_phase2, which then finishes the cascade over the whole video for those
 *                            columns, so their means round like the unsharded pass / the reference.
 * When the conditions do not hold both calls fall back to vc2_scores_phase2's behaviour (exact means, counted). */
int vc2_video_centre_blocks(const void* x, int64_t F, int64_t N, int64_t D, int dtype, const int32_t* cols,
                            int64_t C, const int32_t* spos, const double* csum_all, int64_t P,
                            int64_t csum_stride, int64_t rows_per_rank, int64_t R_total, int64_t row0, void* ws,
                            size_t ws_bytes, float* blocks_out, int cap, void* stream);
int vc2_scores_phase2_blocks(const void* x, int64_t F, int64_t N, int64_t D, int dtype, const int32_t* cols,
                             int64_t C, const int32_t* spos, const double* csum_all, int64_t P,
                             int64_t csum_stride, int64_t rows_per_rank, int64_t R_total, void* ws, size_t ws_bytes,
                             void* v_T, void* f_T, float* total_f32, float* s_f32, const float* blocks_all, int world,
                             int cap, void* stream);

/* More flagged columns than one exchange carries (only the proven-margin and debug modes flag that many): further
 * ROUNDS of exchange 2b.  After vc2_video_centre_blocks (round 0: flags, count, records of the first `cap` columns) the
 * caller reads the count (vc2_video_centre_flagged; every rank reads the same number), and for every round r:
 *   vc2_video_centre_blocks_round(col_offset = r * cap)   records of flagged columns [r * cap, (r + 1) * cap)   (r >= 1)
 *   (all-gather)
 *   vc2_video_centre_finish_round(col_offset = r * cap)   torch's cascade over the whole video for those columns; the
 *                                                         video centre in the workspace is corrected in place  (r >= 0)
 * and finally vc2_scores_phase2_blocks(blocks_all = NULL, world = -1): "the centre is final", sweep 3 only.
 * A level-0 block may meet any number of ranks (ranks with fewer rows than a block included); the ranks must hold
 * equal row counts. */
int vc2_video_centre_blocks_round(const void* x, int64_t F, int64_t N, int64_t D, int dtype, const int32_t* cols,
                                  int64_t C, const int32_t* spos, int64_t R_total, int64_t row0, void* ws,
                                  size_t ws_bytes, float* blocks_out, int cap, int col_offset, void* stream);
int vc2_video_centre_flagged(int64_t F, int64_t N, int64_t D, int dtype, int64_t R_total, const void* ws, size_t ws_bytes,
                             int32_t* count_host, void* stream);   /* synchronises `stream`; host pointer */
int vc2_video_centre_finish_round(int64_t F, int64_t N, int64_t D, int dtype, int64_t C, const int32_t* spos,
                                  int64_t R_total, void* ws, size_t ws_bytes, const float* blocks_all, int world,
                                  int cap, int col_offset, void* stream);

/* Step 3 of the sharded path: s_all_f32[F_total] = the all-gathered per-frame uniqueness scores
 * (fp32-widened T values); budgets are computed over all F_total frames, selection + gather only for
 * this rank's frames [f0, f0+F_local).  idx_out holds LOCAL linear indices (f_local*N + n).  K_out is int64[3] here:
 * K_out[2] = the number of video-centre values within the replay margin of a T rounding boundary, which this path
 * leaves at the exactly rounded mean (mode 1; the unsharded pass replays torch's summation order for them). */
int vc2_select_sharded(const float* total_f32, const float* s_all_f32, int64_t F_total, int64_t f0,
                       int64_t F_local, int64_t N, int64_t D, double base_scale, int dtype, void* ws,
                       size_t ws_bytes, int64_t* ks, int64_t* idx_out, int64_t cap, int64_t* K_out,
                       const void* gather_src, void* out_rows, void* stream);

/* _multi_scale_gaussian(x, center, alphas) as a standalone call -- vidcom2.py:59-62 (the fused pass
 * never materialises x; this entry serves callers of the helper itself).  x: T[F*N, C] as given,
 * centre: T[n_centres, C] with n_centres == 1 (video centre) or F (one per frame), alphas: n_alphas
 * host doubles (<= 16).  out: T[F*N] = sum_a exp(-||x - c||^2 / (2 a)), every op rounded to T. */
int vc2_multi_scale_gaussian(const void* x, int64_t F, int64_t N, int64_t C, int dtype, const void* centre,
                             int64_t n_centres, const double* alphas, int n_alphas, void* out_T, void* stream);

/* ---- small device utilities used by tests / bench ----------------------------------- */
/* exp over every T bit pattern as the path computes it: out[i] = RN_T(exp(in[i])) (KAT). */
int vc2_kat_exp(const void* in_T, int64_t n, int dtype, void* out_T, void* stream);
/* RN_T round trip of fp32 values (KAT for the conversion instructions). */
int vc2_kat_round(const float* in, int64_t n, int dtype, void* out_T, void* stream);

/* ---- per-kernel timing (bench.py roofline leg) ----------------------------------------
 * When enabled every kernel launch is bracketed by hipEvents on its own stream.
 * vc2_profile_collect synchronises on them and returns accumulated milliseconds / launch counts
 * per kernel (returns the number of kernels reported). */
int vc2_profile_enable(int on);
int vc2_profile_collect(int max_kernels, const char** names, double* total_ms, int64_t* launches);

/* Diagnostic (SYNCHRONISES the device): the "torch order" replay counters of the last pass that used `ws`
 * for this shape, out8 = host int32[8]. */
int vc2_pass_counters(int64_t F, int64_t N, int64_t D, int dtype, const void* ws, int32_t* out8_host);
/* Diagnostic (SYNCHRONISES): every loop of the selection replay is bounded; out8 = how often each bound actually
 * expired since the last reset (all zero on a healthy run; the test-suite asserts it). */
int vc2_selftest_counters(int32_t* out8_host, int reset);
/* Test hook: the selection launches that follow report a loop-bound hit (status bit 4) from the workgroup of local frame
 * `frame` (-1: off) -- how the test-suite checks that a hit in ANY workgroup of the launch reaches K_out[1] and the host
 * mirror's final word.  Process-wide; nothing is computed differently. */
int vc2_selftest_force_guard(int frame);
/* Host arithmetic only (no device call): the geometry of sweep 2's ORD form for this shape in the current mode -- the
 * workgroups that add a frame's x^ in torch's own order (vidcom2.py:51-52; k_norm_colsum2<.., ORD>, OrdGeo).  out5 = int32[cap][5]
 * rows {frame, piece, pieces of that frame, first 16-row block, blocks}, in launch order.  Returns the number of streaming
 * workgroups (0: the shape has no ORD geometry -- the row-interleaved sweep with margins and replays runs), or an error (< 0).
 * The test-suite checks that the pieces tile every frame exactly once and fit the resident workgroup slots. */
int vc2_selftest_ord_pieces(int64_t F, int64_t N, int64_t D, int dtype, int32_t* out5_host, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* VC2_H_ */
