/*
 * gsraster.h -- C ABI of libgsraster_hip.so, the MI355X (gfx950) tile rasterizer that
 * replaces GS-LIVM's CudaRasterizer::Rasterizer static API.
 *
 * Every entry point cites the reference interface it replaces (paths relative to the
 * GS-LIVM tree).  All pointers are DEVICE pointers to f32 data unless stated; matrices
 * are 16 floats, column-major (the layout include/gs/cuda_rasterizer/auxiliary.h:48-64
 * indexes).  Nullable pointers follow the reference's convention: NULL == "not
 * provided" (a size-0 tensor in the Torch binding, src/gs/rasterizer.cu:178-193).
 *
 * No Torch, GLM or STL types cross this boundary.  The library never allocates or frees
 * device memory: scratch comes from the three allocator callbacks (forward) or from the
 * blobs those callbacks returned (backward).  All work is enqueued on `stream`
 * (a hipStream_t passed as void*; NULL = the null stream).
 *
 * Errors: functions return a negative gsr_status; gsr_last_error() returns a
 * thread-local message.  (The reference throws std::runtime_error / exits; a C ABI
 * cannot, the binding in gs-livm_amd/csrc/torch_binding.cpp converts.)
 */
#ifndef GSRASTER_H_INCLUDED
#define GSRASTER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_ABI_VERSION 2

typedef enum gsr_status {
  GSR_OK = 0,
  GSR_ERR_INVALID_ARGUMENT = -1, /* bad shape / null required pointer                 */
  GSR_ERR_ALLOC = -2,            /* an allocator callback returned NULL               */
  GSR_ERR_HIP = -3,              /* a HIP runtime call or kernel launch failed        */
  GSR_ERR_UNSUPPORTED = -4       /* e.g. SH degree > 3, M > 16                        */
} gsr_status;

/* Replaces std::function<char*(size_t)> (include/gs/cuda_rasterizer/rasterizer.h:24-26;
 * Torch side: resizeFunctional, src/gs/rasterize_points.cu:36-44).  Called at most once
 * per gsr_forward per blob; must return device memory (>= `bytes`, 256-B aligned) that
 * stays valid until the matching gsr_backward has completed on the stream. */
typedef char* (*gsr_alloc_fn)(void* ctx, size_t bytes);

/* Replaces CudaRasterizer::Rasterizer::forward (rasterizer.h:23-51,
 * src/cuda_rasterizer/rasterizer_impl.cu:181-342).
 *   P Gaussians, D = SH degree (0..3), M = SH coefficients per Gaussian (shs is [P][M][3]).
 *   out_color [3][H][W] planar, out_depth [H][W], out_acc [H][W] are fully written.
 *   radii [P] int32 (nullable: an internal array is used, as rasterizer_impl.cu:217-219).
 *   prefiltered / debug: accepted for signature parity; prefiltered has no effect in the
 *   reference forward either; debug != 0 synchronises the
 *   stream after every stage and reports kernel errors (auxiliary.h:146-154).
 * Returns, on success, the BINNING KEY (>= 0): the number to pass to gsr_backward as `R` (and to
 * gsr_binning_view_of) together with the three blobs -- the instance capacity the binning blob was carved for;
 * a negative gsr_status on failure.
 *   - Synchronous forward (a host thread's first forward, every debug != 0 forward, GSR_SYNC_FORWARD=1): the
 *     host waits for num_rendered where the reference has its blocking cudaMemcpy
 *     (rasterizer_impl.cu:277), sizes the binning blob exactly, and the key IS num_rendered.
 *   - Speculative forward (the steady state): the binning blob is allocated for a capacity predicted
 *     from the calling thread's recent forwards (5/4 of the largest of the last four + 64 Ki), every
 *     kernel is enqueued at once and reads num_rendered from device memory, and the host looks at
 *     the count only after the last launch, when it has long been written -- no host wait, no GPU
 *     idle.  The key is the capacity (>= num_rendered).  If num_rendered exceeded the capacity
 *     the binning allocator is called a SECOND time (exact size; the first allocation may be
 *     released) and the binning chain is enqueued again; results are identical either way.
 *   gsr_last_num_rendered() returns the calling thread's last forward's exact num_rendered (the
 *   reference's return value: the instances emitted, sorted and ranged) without touching the device.
 * The host-thread state behind this (mailbox word, counters, prediction) is per thread and device;
 * one thread's forwards must be ordered on the device (ONE stream at a time per host thread -- the
 * reference uses the default stream only); a thread that changes streams is detected and the
 * previous stream is drained first.  Calls from different host threads are independent. */
int gsr_forward(gsr_alloc_fn geometry_alloc, void* geometry_ctx, gsr_alloc_fn binning_alloc, void* binning_ctx,
                gsr_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background, int width,
                int height, const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_depth, float* out_acc,
                int* radii, int debug, void* stream);

int gsr_last_num_rendered(void);
/* Test / tuning hook for the speculative forward: capacity >= 1 = the calling thread's NEXT forward
 * allocates its binning blob for exactly that many instances (smaller than num_rendered forces the
 * overflow path); 0 = forget the thread's history (its next forward is synchronous); negative = no
 * override.  Returns the previous override (-1 = none). */
long long gsr_set_binning_capacity_hint(long long capacity);
unsigned long long gsr_speculative_forwards(void);  /* process-wide counters */
unsigned long long gsr_speculation_overflows(void);

/* Near/far frames.  A tile's list is depth-ordered and a pixel stops reading it once its transmittance
 * is below 1e-4 (forward.cu:380-383); in dense scenes every tile is finished after a few per cent of
 * its list, and emitting, sorting and ranging the rest is most of the forward.  A speculative forward in
 * the default binning mode whose predicted instance count is at least three times the near budget
 * (GSR_NEAR_ENTRIES list entries per tile, default 320) therefore bins the Gaussians in two chains
 * in depth order: the NEAR Gaussians (until they fill the budget) are binned and blended; then only
 * the FAR Gaussians whose tile rectangle still contains an unfinished tile are binned (whole
 * rectangles) and blended on top.  What is left out lies, in every tile it would have gone to,
 * behind the point where every pixel has stopped: images, n_contrib, final_T and every gradient are
 * bit-identical to the one-chain frame; only the lists (num_rendered, point_list, ranges -- each tile's
 * list is its near segment followed by its far segment, gsr_image_view.ranges / .ranges_far) are
 * shorter.  Frames with gsr_set_reference_rects(1), debug frames and synchronous forwards are never
 * split.  gsr_set_near_far(0) / GSR_NEAR_FAR=0 switches the feature off process-wide; returns the previous value.
 * gsr_set_near_far_thread(mode): the same for the CALLING THREAD's forwards only (1 on, 0 off, negative = follow the
 * process-wide value, the default) -- the reference renders from several threads, and a thread that needs a mode of its
 * own must not disturb the others; returns the thread's previous setting (-1 = none).
 * gsr_last_near_far: 1 if the calling thread's last forward was split, with its two instance counts.
 * gsr_set_near_far_hints (test / tuning hook, calling thread): near list entries per tile (< 0 =
 * default) and the far capacity of the NEXT split forward (< 0 = from history; too small forces the
 * redo path). */
int gsr_set_near_far(int on);
int gsr_set_near_far_thread(int mode);
int gsr_near_far(void);  /* what the calling thread's next forward will use */
int gsr_last_near_far(unsigned* near_instances, unsigned* far_instances);
void gsr_set_near_far_hints(long long near_entries_per_tile, long long far_capacity);
unsigned long long gsr_near_far_forwards(void);
/* Far-chain speculation.  When a thread's last two split forwards left no quad unfinished after the near
 * chain (a dense scene: the far chain's launches found nothing to do), its next split forward is
 *   - asynchronous, where the device supports stream-side waits (hipStreamWaitValue32; GSR_ASYNC_FAR=0
 *     switches this off) and while ONE host thread of the process is rendering (one thread at a time owns the
 *     mechanism, and a process whose rendering hands over between threads twice within 100 ms takes the
 *     host-decided variant below): the far chain is enqueued on a stream of the library's own behind a wait for
 *     the near blend's decision, each of its kernels gated on "needed", and the caller's stream waits
 *     for the frame's go word -- stored by the near blend itself when nothing is left to do, by the far
 *     chain's last kernel otherwise.  gsr_forward returns without waiting for the decision; the far
 *     segment of the binning blob is sized for every instance behind the near budget;
 *   - otherwise decided by the host: the count of unfinished quads arrives in the mailbox where
 *     num_rendered does, and the far chain is enqueued only if it is non-zero (one host round trip).
 * Such a frame also sorts only its NEAR candidates by depth up front (partial depth sort: the Gaussians whose depth key
 * lies in the top-byte groups the near budget can reach); the sort of all P Gaussians moves into the far chain
 * (GSR_FULL_DEPTH_SORT=1: always up front).  gsr_geometry_view.depth_order is the full order only in frames whose far
 * chain ran or that sorted everything up front.
 * Results are identical in every variant.  While an asynchronous frame is in flight,
 * gsr_last_num_rendered / gsr_last_near_far report the near chain's count and add the far chain's once
 * the frame has got there (they never wait).
 * gsr_set_far_speculation (test / tuning hook, calling thread): 1 = the NEXT split forward speculates,
 * 0 = none does, negative = automatic (and the streak restarts); returns the previous setting.
 * gsr_last_far_skipped: 1 if the calling thread's last forward completed without running a far chain. */
int gsr_set_far_speculation(int mode);
int gsr_last_far_skipped(void);
unsigned long long gsr_far_skips(void);       /* process-wide counters */
unsigned long long gsr_far_skip_misses(void);
unsigned long long gsr_async_far_frames(void);
/* Asynchronous frames report their outcome (quads left unfinished, far-chain instances) through a mailbox slot of
 * their own; up to eight such frames of a thread may be outstanding.  gsr_async_outcomes_pending: how many of the calling
 * thread's are still unresolved (never waits); gsr_async_outcomes_lost: frames whose slot had to be reused before it
 * was read (a ninth frame in flight) -- each is counted as a miss by the speculation rule.  Process-wide counter. */
int gsr_async_outcomes_pending(void);
unsigned long long gsr_async_outcomes_lost(void);
/* What gsr_backward knows about the forward that filled an image blob (tile order already computed, frame was split)
 * is kept per blob address for as long as the blob may be differentiated; a backward that finds nothing (more than
 * 4096 blobs alive) takes the general path -- same results, one or two launches more -- and is counted here. */
unsigned long long gsr_frame_note_misses(void);
/* Per-view histories.  Everything a speculative forward predicts from -- the binning capacity, the far capacity, the
 * far-chain speculation's streak, the near budget below -- is kept per VIEW of the calling thread: a view is recognised
 * by the device address of its view matrix and the image size (the reference renders several keyframes in turn before
 * one backward, each Camera holding its matrices for its lifetime: lioOptimization.cpp:1691-1737, camera.cu:36-48).
 * Up to 32 views per thread, least recently used replaced; a view seen for the first time starts from the thread's most
 * recently used history, so a caller with one camera, or one that passes a fresh matrix tensor every frame, sees the
 * per-thread behaviour.  A wrong guess costs a redo or a far chain, never a result.
 * Adaptive near budget (per view).  The budget of a split frame is GSR_NEAR_ENTRIES (320) list entries per tile
 * times scale / 256.  A frame whose far chain ran, over less than four times the near chain's instances, raises the
 * scale by 64 (a quarter of the configured budget), up to 768, ON PROBATION: if the view's next frame still leaves more
 * than three quarters of those quads unfinished, the tiles do not finish for lack of budget (sky, the border of the
 * map) -- the raise is taken back and none is tried for 256 frames of the view.  Sixty-four frames in a row without a
 * far chain lower the scale by 16, down to 256.  gsr_near_budget_scale returns the current scale of the view of the
 * calling thread's last forward; gsr_near_budget_feedback (test hook) feeds the outcome of a frame -- unfinished quads
 * after the near chain, near and far instance counts -- to that view's rule as a forward does and returns the scale
 * after it.  The rule rests while gsr_set_near_far_hints sets the budget. */
unsigned gsr_near_budget_scale(void);
unsigned gsr_near_budget_feedback(unsigned unfinished_quads, unsigned near_instances, unsigned far_instances);
/* Eight misses in a row of frames that splitting does not shorten by a quarter (near + far instances against all the
 * frame's; or a far chain of at least four times the near chain: a sparse scene) pause the splitting: the view's next
 * 256 frames are binned in one chain, then it tries again.  gsr_near_far_pause(frames): returns the frames left of the
 * pause of the view of the calling thread's last forward and, if frames >= 0, sets them (0 ends a pause). */
int gsr_near_far_pause(int frames);

/* Replaces CudaRasterizer::Rasterizer::backward (rasterizer.h:53-88,
 * rasterizer_impl.cu:346-457).  geom/binning/image blobs are the ones the forward
 * allocator callbacks returned; R is gsr_forward's return value (the binning key).
 * The nine gradient outputs are FULLY OVERWRITTEN (Gaussians with radii <= 0 get
 * zeros), so the caller need not pre-zero them (the reference requires zeroed buffers,
 * rasterize_points.cu:173-181; zeroed buffers remain valid input).
 *   dL_dmean2D [P][3] (.x,.y written, .z = 0), dL_dconic [P][4] (.x,.y,.w; .z = 0),
 *   dL_dopacity [P], dL_dcolor [P][3], dL_dmean3D [P][3], dL_dcov3D [P][6],
 *   (dL_dcov3D may be NULL when cov3D_precomp is NULL: the gradient w.r.t. a covariance the library computed
 *   itself from scales and rotations is an intermediate; it is then not written),
 *   dL_dsh [P][M][3], dL_dscale [P][3], dL_drot [P][4].
 * The gradient w.r.t. depth is not an input, exactly as in the reference
 * (src/gs/rasterizer.cu:79; backward.cu:451-452). */
int gsr_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                 const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                 char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
                 const float* dL_dacc, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                 float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug,
                 void* stream);

/* Replaces CudaRasterizer::Rasterizer::markVisible (rasterizer.h:21,
 * rasterizer_impl.cu:128-135): present[i] = z_view > 0.2 (1-byte bool). */
int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     unsigned char* present, void* stream);

/* Replace required<GeometryState|ImageState|BinningState>() (rasterizer_impl.h:62-67):
 * bytes the corresponding allocator callback will be asked for. */
size_t gsr_geometry_bytes(int P);
size_t gsr_image_bytes(int width, int height);
size_t gsr_binning_bytes(int R);

/* Views into the opaque blobs, for parity tests and tooling (the reference exposes the
 * same arrays through GeometryState/BinningState/ImageState, rasterizer_impl.h:28-60).
 * Pointers are device pointers into the blob; arrays the implementation does not keep
 * are NULL. */
typedef struct gsr_geometry_view {
  const float* depths;            /* NULL: not kept (the view-space depth is splats[i][9])              */
  const int32_t* radii;           /* [P] internal radii: valid only when gsr_forward got radii == NULL  */
  const float* splats;            /* [P][12]: x, y, conic.x, conic.y, conic.z, opacity, r, g, b, depth, hx, hy */
  const float* cov3D;             /* [P][6] filled only when the forward ran with debug != 0 (the backward
                                     recomputes the covariance from scale and rotation)                 */
  const uint32_t* tiles_touched;  /* NULL: not kept separately (gpack[i][0])                            */
  const uint32_t* point_offsets;  /* [P] inclusive scan in id order (the reference's array; unused by this
                                     pipeline: filled only when the forward ran with debug != 0) */
  const uint8_t* clamped;         /* [P] bit0..2 = r,g,b clamp flags      */
  const uint32_t* depth_order;    /* [P] Gaussian ids by (depth bits, id); culled Gaussians last */
  const uint32_t* num_rendered;   /* device counters of the forward: [0] instances of all Gaussians (= num_rendered of a
                                     one-chain frame); near/far frames ([12] == 1): [6] near instances, [8] far
                                     instances (num_rendered = [6] + [8]), [7] first far Gaussian in depth_order,
                                     [9] tiles unfinished after the near phase, [10] far Gaussians emitted */
  const uint32_t* gpack;          /* [P][2]: tiles touched, packed tile rect x0 | y0 << 10 | width << 20 */
} gsr_geometry_view;
/* The reference's 64-bit sorted key of instance i is ((uint64)tile << 32) | bits(depths[point_list[i]]) with
 * tile = the tile whose range [ranges[tile][0], ranges[tile][1]) contains i: this implementation sorts the
 * Gaussians by depth once and the instances by a 16- or 32-bit tile id only (gs-livm_amd/csrc/radix_sort.hip);
 * the sorted tile ids are scratch (overwritten by the backward), so the view exposes point_list and the keys
 * are implied by ranges + point_list. */
typedef struct gsr_binning_view {
  const uint32_t* point_list;     /* [R] sorted Gaussian ids              */
} gsr_binning_view;
typedef struct gsr_image_view {
  const uint32_t* ranges;         /* [tiles][2] the tile's list segment in point_list (near segment of a near/far frame) */
  const float* final_T;           /* [H][W]                               */
  const uint32_t* n_contrib;      /* [H][W]                               */
  const uint32_t* quad_last;      /* [tiles][4] max n_contrib inside each 8x8 quad of the tile (0,1 = top, 2,3 = bottom);
                                     the tile's maximum = list entries the backward walks */
  const uint32_t* ranges_far;     /* [tiles][2] near/far frames: the far segment that follows `ranges` in the tile's
                                     list (positions count through both); (0, 0) otherwise */
} gsr_image_view;
int gsr_geometry_view_of(char* geom_buffer, int P, gsr_geometry_view* out);
int gsr_binning_view_of(char* binning_buffer, int R, gsr_binning_view* out);
int gsr_image_view_of(char* image_buffer, int width, int height, gsr_image_view* out);

/* Binning mode.  The reference emits one (Gaussian, tile) instance for EVERY tile of the square
 * getRect() puts around the 3-sigma radius (include/gs/cuda_rasterizer/auxiliary.h:39-46,
 * duplicateWithKeys, src/cuda_rasterizer/rasterizer_impl.cu:64-101) and then skips, pixel by pixel, the
 * instances whose alpha stays below 1/255 (forward.cu:374-376).
 *   0 (default): "culled" -- an instance is emitted only where the Gaussian's exact-conservative
 *      alpha >= 1/255 footprint box overlaps the tile.  Images, radii and gradients are unchanged;
 *      tiles_touched, num_rendered, the per-tile lists, ranges and n_contrib describe the shorter lists.
 *   1: "reference" -- getRect's square as it is: tiles_touched, num_rendered, point_list, ranges and
 *      n_contrib equal the reference's bit for bit (about 1.4x the instances at 2 M Gaussians, 1080p).
 * Process-wide default, read at every gsr_forward (a backward follows the mode its forward ran in: the rectangles
 * are stored in the blobs); the initial value comes from the environment variable GSR_REFERENCE_RECTS
 * (unset / "0" = culled).  gsr_set_reference_rects returns the previous value.
 * gsr_set_reference_rects_thread(mode): the mode of the CALLING THREAD's forwards only (1 / 0; negative = follow the
 * process-wide value, the default); returns the thread's previous setting (-1 = none).  gsr_reference_rects: what the
 * calling thread's next forward will use. */
int gsr_set_reference_rects(int on);
int gsr_set_reference_rects_thread(int mode);
int gsr_reference_rects(void);

/* getHigherMsb (rasterizer_impl.cu:35-48): number of tile-id bits the sort covers. */
uint32_t gsr_higher_msb(uint32_t n);

/* ---- "next" row: the caller's per-Gaussian elementwise work, fused ----
 * gsr_activate replaces the five Torch ops of GaussianModel's getters (include/gs/gs/gaussian.cuh:40-54):
 *   scales = exp(_scaling) [P][3], rotations = normalize(_rotation) [P][4] (x / max(|x|, 1e-12)),
 *   opacities = sigmoid(_opacity) [P], shs = cat(_features_dc [P][1][3], _features_rest [P][M-1][3], 1).
 * gsr_activate_backward is the matching chain rule (needs the raw rotation and the activated scales /
 * opacities); every output element is written.
 * gsr_adam_step is torch::optim::Adam::step (as configured in src/gs/gaussian.cu:396-428: per-group lr,
 * no weight decay, no amsgrad) for up to 8 tensors in ONE launch; `step` counts from 1; zero_grads != 0
 * clears the gradients it consumed (the reference's zero_grad, src/liw/lioOptimization.cpp:1831-1832).
 * beta1 / beta2 / eps are doubles as in torch::optim::AdamOptions: 1 - beta is formed in double and then narrowed.
 * Pointer arrays are HOST arrays of device pointers. */
int gsr_activate(int P, int M, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                 const float* features_dc, const float* features_rest, float* scales, float* rotations,
                 float* opacities, float* shs, void* stream);
int gsr_activate_backward(int P, int M, const float* rotation_raw, const float* scales, const float* opacities,
                          const float* dL_dscales, const float* dL_drotations, const float* dL_dopacities,
                          const float* dL_dshs, float* dL_dscaling_raw, float* dL_drotation_raw,
                          float* dL_dopacity_raw, float* dL_dfeatures_dc, float* dL_dfeatures_rest, void* stream);
/* gsr_model_step: the whole optimiser tail in one pass over the model -- gsr_activate_backward's chain rule, Adam on
 * the six groups (order xyz, features_dc, features_rest, scaling, rotation, opacity: the reference's group order,
 * src/gs/gaussian.cu:396-428) and gsr_activate of the UPDATED parameters for the next forward (outputs nullable).
 * Gradients are w.r.t. xyz and the ACTIVATED scales / rotations / opacities / shs, i.e. exactly what the
 * rasterizer's backward returns; the raw-space gradients never reach memory.  Same arithmetic as the three
 * separate entry points. */
int gsr_model_step(int P, int M, float* const* params6, float* const* exp_avg6, float* const* exp_avg_sq6,
                   const float* dL_dxyz, const float* dL_dscales, const float* dL_drotations, const float* dL_dopacities,
                   const float* dL_dshs, float* scales_out, float* rotations_out, float* opacities_out, float* shs_out,
                   const float* lr6, double beta1, double beta2, double eps, int step, void* stream);
int gsr_adam_step(int n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const size_t* numel, const float* lr, double beta1, double beta2, double eps,
                  int step, int zero_grads, void* stream);

/* ---- "next" row: the photometric loss that follows every render, fused ----
 *   L = (1 - lambda) * mean|img - gt| + lambda * (1 - mean(SSIM(img, gt)))
 * replaces gaussian_splatting::l1_loss + ::ssim (include/gs/gs/loss_utils.cuh:11-13,43-70; combined at
 * src/liw/lioOptimization.cpp:1705-1710) and their autograd: five grouped 11x11 conv2d per view there, three
 * launches here.  img/gt: [channels][H][W] device f32; window11_host: the 11 taps of the separable window on the
 * HOST (the reference's 2-D window is their outer product, loss_utils.cuh:33-37; it need not be symmetric);
 * zero padding 5 like conv2d(padding = window_size/2).  loss_out3 (device) = {L, mean L1, mean SSIM};
 * dL_dimg (nullable) = dL/dimg for upstream gradient 1.  Deterministic (fixed-order reductions). */
size_t gsr_photometric_loss_workspace(int channels, int height, int width);
int gsr_photometric_loss(int channels, int height, int width, const float* img, const float* gt,
                         const float* window11_host, float lambda_dssim, float* loss_out3, float* dL_dimg,
                         char* workspace, size_t workspace_bytes, void* stream);

/* ---- "next" row: the data formats either side of the path ----
 * gsr_init_gaussians: new map points -> leaf parameter rows, the arithmetic of GaussianModel::addNewPointcloud
 * (src/gs/gaussian.cu:241-313): _xyz = xyz; _scaling = log(sqrt(diag(cov) * scale_factor)) (decomposeSR keeps the
 * diagonal, :10-11, 276-281); _rotation = (1, 0, 0, 0); _opacity = inverse_sigmoid(0.5) = 0;
 * _features_dc[.][0][c] = (rgb_c / 255 - 0.5) / C0; _features_rest = 0.  covs [n][3][3], rgbs [n][3] in 0..255.
 * The *_out pointers address row `P_old` of the caller's capacity buffers: rows are initialised in place, so
 * growing the map moves O(n) bytes instead of re-concatenating all six tensors and their Adam moments
 * (densification_postfix / cat_tensors_to_optimizer, :451-472, 524-540).
 * gsr_pack_ply_rows: the vertex rows of the reference's PLY export (construct_list_of_attributes :474-492,
 * Save_ply :494-522): per Gaussian gsr_ply_row_floats(M) = 14 + 3M little-endian f32 --
 *   x y z | nx ny nz (zeros) | f_dc_0..2 | f_rest_0..3(M-1)-1 | opacity | scale_0..2 | rot_0..3
 * with f_dc / f_rest in the reference's transpose(1,2).flatten(1) (channel-major) order.  rows: device buffer of
 * P * (14 + 3M) floats; the host copies it once and writes header + rows (gs-livm_amd/ply.py). */
int gsr_init_gaussians(int n, int M, const float* xyz, const float* covs, const float* rgbs, float scale_factor,
                       float* xyz_out, float* features_dc_out, float* features_rest_out, float* scaling_out,
                       float* rotation_out, float* opacity_out, void* stream);
size_t gsr_ply_row_floats(int M);
int gsr_pack_ply_rows(int P, int M, const float* xyz, const float* features_dc, const float* features_rest,
                      const float* opacity, const float* scaling, const float* rotation, float* rows, void* stream);

/* Optional per-kernel device timing (hipEvent pairs recorded on the launch stream around every
 * kernel launch while enabled).  Measurement aid for bench.py's roofline line; the reference has only
 * host wall-clock timers (include/common/timer/timer.h:36-52).  Not thread-safe; off by default.
 *   (the event pool is guarded by a mutex; enabling / reading while other threads launch is safe, the
 *   attribution of events to forwards of concurrent threads is then simply interleaved)
 *   gsr_profile_enable(1) starts a fresh recording, gsr_profile_enable(0) stops it.
 *   gsr_profile_read synchronises the recorded events, writes per-kernel total milliseconds and launch
 *   counts for kernel ids [0, gsr_kernel_count()), clears the recording, returns the number of ids written. */
int gsr_kernel_count(void);
const char* gsr_kernel_name(int kernel_id);
int gsr_profile_enable(int on);
/* like gsr_profile_enable(1), but records only the listed kernel ids: an event pair costs a few microseconds
 * of GPU idle time per launch, so a timed run should carry events for the kernel under study only. */
int gsr_profile_enable_only(const int* kernel_ids, int n);
int gsr_profile_read(int max_ids, double* total_ms, int* launches);
/* diagnostics: number of gsr_forward calls of this process whose instance count reached the host through the
 * fallback stream query instead of the polled mailbox word (0 in a healthy run; each one leaves the GPU idle for up
 * to one query period, 25 us). */
unsigned long long gsr_mailbox_slow_path_hits(void);
/* ... and what the most recent such exit saw: the ticket it waited for, the ticket in the word at the last spin
 * and right after the first successful hipStreamQuery, the time since the enqueue, and whether the word was
 * already there when that HIP call returned (1 = the count was only invisible to the spinning load until a HIP
 * call was made; 0 = the store itself arrived after the stream had drained).  count = 0: never happened. */
typedef struct gsr_mailbox_event {
  unsigned count;
  uint32_t ticket_expected, ticket_seen_before_query;
  double elapsed_us;
  uint32_t ticket_seen_after_query;
  int first_query_result;
  int visible_at_query;
  /* how the wait was spent: stream queries made before the one that found the stream drained, the longest any single
   * query took to RETURN, and the longest the host went without getting to run (gap between two polls of the word):
   * a drained-late stream shows many quick queries, a stalled driver call one long query, a descheduled host thread one
   * long gap */
  unsigned queries;
  double longest_query_us, longest_poll_gap_us;
} gsr_mailbox_event;
int gsr_mailbox_slow_path_last(gsr_mailbox_event* out);

const char* gsr_last_error(void);
int gsr_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GSRASTER_H_INCLUDED */
