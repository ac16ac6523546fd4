// gsr_internal.hpp -- state layouts and stage launchers shared by the HIP translation units
// of libgsraster_hip.so (gfx950 only).  Not part of the public ABI (include/gsraster.h).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stddef.h>
#include <stdint.h>

namespace gsr {

constexpr int TILE = 16;             // tile edge in pixels (reference: config.h:16-17)
constexpr int SPLAT_F4 = 3;          // float4s per splat record (48 B)
constexpr int GRAD_F4 = 3;           // float4s per per-instance gradient record (9 used of 12 floats)
constexpr int PRE_BLOCK = 256;       // threads per block of the per-Gaussian kernels
constexpr int SORT_TILE = 4096;      // pairs per workgroup tile of the depth sort (4 waves x 64 lanes x 16)
constexpr int TSORT_TILE = 4096;     // pairs per workgroup tile of the tile sort (8192 on 8 waves -- twice the run length
                                     // per digit in the write-out -- measured the same: the pass is traffic-bound)
constexpr int TSORT_WAVES = 4;
constexpr int EMIT_CHUNK = 2048;     // instance slots emitted per workgroup
constexpr int SORT_CHUNK = 64;       // workgroup tiles per scan chunk
constexpr size_t ALIGN = 256;

inline size_t align_up(size_t v) { return (v + ALIGN - 1) & ~(ALIGN - 1); }

// Bump carver over an opaque blob; with base == nullptr it only measures (the same trick as
// the reference's required<T>(), rasterizer_impl.h:62-67).
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(char* b) : base(b) {}
  template <typename T>
  T* take(size_t count) {
    off = align_up(off);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

// Per-Gaussian state (replaces GeometryState, rasterizer_impl.cu:137-151).  The splat record packs
// everything the blend kernels need about one Gaussian into 48 contiguous bytes so a tile's list
// is gathered with three 16-B loads per instance:
//   f4[0] = (x, y, conic.x, conic.y)  f4[1] = (conic.z, opacity, r, g)  f4[2] = (b, depth, hx, hy)
// hx/hy: half-extents (pixels) of a box that contains every pixel the splat can contribute to
// (alpha >= 1/255), used for exact-conservative culling of 8x8 pixel quads.
// The Gaussians are also sorted by depth once (order[]), so the per-instance sort only has to order
// by tile id (see BinningState).
struct SortScratch {  // scratch of one radix sort over n (u32 key, u32 value) pairs
  uint32_t* counts;      // [ntiles][256] per workgroup-tile digit counts -> exclusive in-chunk prefixes
  uint32_t* chunk_sums;  // [nchunks][256] -> exclusive chunk bases
  uint32_t* digit_base;  // [256] digit totals
  static void carve(Carver& c, size_t n, SortScratch& s) {
    const size_t ntiles = (n + SORT_TILE - 1) / SORT_TILE;
    const size_t nchunks = (ntiles + SORT_CHUNK - 1) / SORT_CHUNK;
    s.counts = c.take<uint32_t>(ntiles * 256);
    s.chunk_sums = c.take<uint32_t>(nchunks * 256);
    s.digit_base = c.take<uint32_t>(256);
  }
};

// Scratch of the single-launch (decoupled look-back) kernels -- the depth sort (radix_sort.hip: all-digit
// histograms, tile tickets, per-(pass, tile, digit) status words) and the slot-offset scan (k_scan_offsets:
// ticket + one 64-bit status word per 4096 Gaussians) -- in ONE array that k_preprocess clears as a side job.
constexpr int SCAN_ITEMS = 16;                    // consecutive Gaussians per thread in k_scan_offsets
constexpr int SCAN_TILE = PRE_BLOCK * SCAN_ITEMS;  // 4096
// Workgroup tile of the depth sort's scatter passes: maps of up to 128 k Gaussians get 1024-pair tiles so a pass
// still spreads over the chip (measured: 48 -> 35 us for the four passes at 10 k; slower from ~300 k up, where the
// longer look-back chain costs more than the wider spread gains).
constexpr int SORT_TILE_SMALL = 1024;
constexpr int SORT_TILE_BIG = 8192;  // on 8 waves, beyond 1 M Gaussians: half as many links in the look-back chain
                                     // (2 M: 122 -> 113 us for the four passes; slower than 4096 at 500 k)
inline size_t depth_sort_tile(size_t n) {
  return n <= ((size_t)1 << 17) ? SORT_TILE_SMALL : n <= ((size_t)1 << 20) ? SORT_TILE : SORT_TILE_BIG;
}

struct DepthSortScratch {
  uint32_t* words;
  size_t nwords;
  size_t scan_off;   // word offset of the scan's status array
  size_t scanB_off;  // ... of the second-phase scan's (k_scan_offsets_far)
  size_t scanC_off;  // ... of the near-candidate compaction's (k_compact_near), followed by its [4][256] digit counts
  __host__ __device__ uint32_t* ghist() const { return words; }           // [4][256]
  __host__ __device__ uint32_t* tickets() const { return words + 1024; }  // [0..3] sort passes, [4] scan, [5] far scan,
                                                                          // [6] compaction, [8..11] near sort passes
  __host__ __device__ uint32_t* status(int pass, int ntiles) const {      // [4][ntiles][256]
    return words + 1088 + (size_t)pass * ntiles * 256;
  }
  __host__ __device__ unsigned long long* scan_status() const {           // [ceil(n / SCAN_TILE)]
    return reinterpret_cast<unsigned long long*>(words + scan_off);
  }
  __host__ __device__ unsigned long long* scanB_status() const {          // [ceil(n / SCAN_TILE)]
    return reinterpret_cast<unsigned long long*>(words + scanB_off);
  }
  __host__ __device__ unsigned long long* scanC_status() const {          // [ceil(n / SCAN_TILE)]
    return reinterpret_cast<unsigned long long*>(words + scanC_off);
  }
  __host__ __device__ uint32_t* ghist_near(size_t n) const {              // [4][256]
    return words + scanC_off + 2 * ((n + SCAN_TILE - 1) / SCAN_TILE);
  }
  static void carve(Carver& c, size_t n, DepthSortScratch& s) {
    const size_t ntiles = (n + depth_sort_tile(n) - 1) / depth_sort_tile(n);
    const size_t nscan = (n + SCAN_TILE - 1) / SCAN_TILE;
    s.scan_off = 1088 + 4 * ntiles * 256;  // even: the 64-bit status words are 8-byte aligned
    s.scanB_off = s.scan_off + 2 * nscan;
    s.scanC_off = s.scanB_off + 2 * nscan;
    s.nwords = s.scanC_off + 2 * nscan + 1024;
    s.words = c.take<uint32_t>(s.nwords);
  }
};

struct GeomState {
  int32_t* radii;           // internal radii: written only when the caller passes no radii array of its own
  float4* splats;           // (view-space depth = splats[3 i + 2].y: no separate depth array)
  float* cov3D;             // debug forwards only (views); the backward recomputes it from scale and rotation
  uint32_t* point_offsets;  // inclusive scan of the tile counts in Gaussian-id order (= the reference's array);
                            // not used by this pipeline: filled only by a debug forward (for the views)
  uint8_t* clamped;         // bit0..2
  uint32_t* total;          // [64] device-side counters: [0] num_rendered (all instances of all Gaussians), [2] touched
                            // Gaussians (backward), [4] order violations (debug self-check); near/far frames (api.hip):
                            // [6] instances of the near phase, [7] depth-order index of the first far Gaussian,
                            // [8] instances of the far phase, [9] tiles still live after the near phase, [10] far
                            // Gaussians emitted, [11] scan tile in which the near budget was crossed (+1; 0 = not yet),
                            // [12] 1 = this frame was binned near/far, [13] quads the near blend left unfinished
  uint2* slotinfo;          // {first slot of the Gaussian's instance run, x0 | y0 << 10 | rect_width << 20}
  uint2* gpack;             // {tiles_touched, packed rect} per Gaussian: ONE 8-byte gather in depth order (the only
                            // copy of the tile counts)
  uint32_t* order;          // [P] Gaussian ids sorted by (depth bits, id); culled Gaussians last.  = dvalsA (written by
                            // the depth sort only: the unsorted ids are the positions 0 .. P-1)
  uint32_t* dkeysA;         // [P] depth-sort ping-pong buffers
  uint32_t* dkeysB;
  uint32_t* dvalsB;
  uint32_t* nkeys2;         // [P] second ping-pong pair of the NEAR sort (partial depth sort, api.hip): the near candidates
  uint32_t* nvals2;         //     are compacted into (dkeysB, dvalsB) and sorted between that pair and this one
  uint4* sdesc;             // [P+1] the emitters' descriptor of order[i], ONE 16-byte load each:
                            //   x = first instance slot (exclusive scan in depth order; sdesc[P].x = R),
                            //   y = Gaussian id, z = packed tile rect x0 | y0 << 10 | width << 20,
                            //   w = ceil(2^32 / width) (exact division by multiply-high)
  uint4* sdescB;            // [P+1] far phase: the descriptors of the far Gaussians that are emitted, compacted
  uint8_t* touched;         // [P] 1 = the blend backward wrote at least one gradient record for this Gaussian
  uint32_t* tlist;          // [P] compacted ids of touched Gaussians (backward); count in total[2]
  DepthSortScratch dsort;
  static GeomState carve(char* blob, size_t P, size_t* bytes = nullptr) {
    Carver c(blob);
    GeomState g;
    g.radii = c.take<int32_t>(P);
    g.splats = c.take<float4>(P * SPLAT_F4);
    g.cov3D = c.take<float>(P * 6);
    g.point_offsets = c.take<uint32_t>(P);
    g.clamped = c.take<uint8_t>(P);
    g.total = c.take<uint32_t>(64);
    g.slotinfo = c.take<uint2>(P);
    g.gpack = c.take<uint2>(P);
    g.order = c.take<uint32_t>(P);
    g.dkeysA = c.take<uint32_t>(P);
    g.dkeysB = c.take<uint32_t>(P);
    g.dvalsB = c.take<uint32_t>(P);
    g.nkeys2 = c.take<uint32_t>(P);
    g.nvals2 = c.take<uint32_t>(P);
    g.sdesc = c.take<uint4>(P + 1);
    g.sdescB = c.take<uint4>(P + 1);
    g.touched = c.take<uint8_t>(P);
    g.tlist = c.take<uint32_t>(P);
    DepthSortScratch::carve(c, P, g.dsort);
    if (bytes) *bytes = align_up(c.off) + ALIGN;
    return g;
  }
};

// Per-image state (replaces ImageState, rasterizer_impl.cu:153-159).
struct ImageState {
  uint2* ranges;        // [tiles]
  uint32_t* quad_last;  // [tiles][4] max n_contrib over each 8x8 quad's pixels (bounds the backward walk)
  float* final_T;       // [H*W]
  uint32_t* n_contrib;  // [H*W]
  uint32_t* tile_order; // [tiles] backward only: tile ids by descending length of the list walk (k_tile_order)
  uint2* rangesB;       // [tiles] near/far frames: the tile's second (far) list segment; (0, 0) otherwise.  A tile's
                        // list is ranges[t] followed by rangesB[t]; list positions (n_contrib, quad_last) count through
  uint8_t* quad_done;   // [tiles][4] near/far frames: 1 = every pixel of the quad was finished by the near phase
  uint32_t* live_sat;   // [(gy+1)(gx+1)] summed-area table of the tiles still live after the near phase
  static ImageState carve(char* blob, int W, int H, size_t* bytes = nullptr) {
    Carver c(blob);
    ImageState s;
    size_t tiles = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    size_t N = (size_t)W * H;
    s.ranges = c.take<uint2>(tiles);
    s.quad_last = c.take<uint32_t>(tiles * 4);
    s.final_T = c.take<float>(N);
    s.n_contrib = c.take<uint32_t>(N);
    s.tile_order = c.take<uint32_t>(tiles);
    s.rangesB = c.take<uint2>(tiles);
    s.quad_done = c.take<uint8_t>(tiles * 4);
    s.live_sat = c.take<uint32_t>((size_t)((W + TILE - 1) / TILE + 1) * ((H + TILE - 1) / TILE + 1));
    if (bytes) *bytes = align_up(c.off) + ALIGN;
    return s;
  }
};

// Per-instance state (replaces BinningState, rasterizer_impl.cu:161-177).  Instances are EMITTED in
// depth order (Gaussians walked through GeomState::order), each Gaussian's run contiguous, with a
// 32-bit key = tile id.  A stable sort on the tile id alone (2 radix passes at 1080p instead of the 6
// a (tile|depth) 64-bit key needs) then yields exactly the reference's order: by tile, then depth
// bits, then Gaussian id.  The sort ping-pongs between (tkeysA, point_list) and (tkeysB, ivalsB); the
// final pass always lands in tkeysA/point_list.
// After the forward only point_list is live, so the backward's per-instance gradient records
// (grad_inst, 48 B each, indexed by emission slot) alias the sort buffers.
struct BinningState {
  uint32_t* point_list;  // [R] sorted Gaussian ids
  uint32_t* tkeysA;      // [R] sorted tile ids (valid until the backward runs)
  uint32_t* tkeysB;
  uint32_t* ivalsB;
  SortScratch tsort;
  float4* grad_inst;     // [R][3], aliases tkeysA..tsort
  uint8_t* inst_flag;    // [R] 1 = grad_inst[slot] was written by the blend backward
  uint32_t* chunk_first; // [R/EMIT_CHUNK + 2] depth-order index of the Gaussian covering slot k*EMIT_CHUNK
  uint32_t* chunk_firstB; // [R/EMIT_CHUNK + 2] the same for the far phase of a near/far frame (index into sdescB)
  static BinningState carve(char* blob, size_t R, size_t* bytes = nullptr) {
    Carver c(blob);
    BinningState b;
    b.point_list = c.take<uint32_t>(R);
    size_t mark = align_up(c.off);
    b.tkeysA = c.take<uint32_t>(R);
    b.tkeysB = c.take<uint32_t>(R);
    b.ivalsB = c.take<uint32_t>(R);
    SortScratch::carve(c, R, b.tsort);
    size_t sort_end = c.off;
    Carver g(blob);
    g.off = mark;
    b.grad_inst = g.take<float4>(R * GRAD_F4);
    c.off = sort_end > g.off ? sort_end : g.off;
    b.inst_flag = c.take<uint8_t>(R);
    b.chunk_first = c.take<uint32_t>(R / EMIT_CHUNK + 2);
    b.chunk_firstB = c.take<uint32_t>(R / EMIT_CHUNK + 2);
    if (bytes) *bytes = align_up(c.off) + ALIGN;
    return b;
  }
};

// Number of splat instances as the binning kernels see it.  `dev == nullptr`: the host knows it (`cap` IS the count:
// the synchronous forward).  Otherwise the host has sized grids and buffers for `cap` instances WITHOUT waiting for the
// count (api.hip, speculative forward) and every kernel reads it from device memory, clamped to `cap` -- an overflowing
// frame then produces in-bounds garbage that the host discards and redoes with the exact count.
// Gate (asynchronous near/far frames, api.hip): the far chain of such a frame is enqueued on a second stream before
// anyone knows whether it is needed; the near blend's last workgroup decides and stores 2 * seq + (needed ? 1 : 0) into
// a library-owned device word (and into the signal word the stream waits for).  Every kernel of that chain carries
// the device word and the value that opens it and leaves at once -- touching nothing: the blobs may already be in use
// by the backward, or freed -- when the word says anything else (seq only grows: a stale kernel never finds it open).
struct Count {
  const uint32_t* dev;
  int cap;
  const uint32_t* gate = nullptr;
  uint32_t gate_open = 0;
  bool bounded = false;  // grids of at most GATED_GRID_MAX workgroups although not gated (a capacity that is rarely used)
  // (the word is device memory and is read like the count below -- it was stored, write-through, before the stream
  // wait that precedes this kernel's launch was satisfied.  Tens of thousands of workgroups ask: an agent-scope atomic
  // load each serialises on the one address -- 0.14 ms for the 62 000 waves of an idle k_emit -- and a load from the
  // host-visible signal word itself stalled them long enough to starve the other stream's kernels of wave slots.)
  __device__ __forceinline__ bool closed() const { return gate && __builtin_nontemporal_load(gate) != gate_open; }
  __device__ __forceinline__ int get() const {
    if (!dev) return cap;
    const uint32_t v = __builtin_nontemporal_load(dev);
    return v < (uint32_t)cap ? (int)v : cap;
  }
};

// Work units of a binning kernel (emit chunks, sort tiles) are walked with a grid stride: a chain whose capacity is
// known to be used launches one workgroup per unit (the loop runs once), the gated far chain of an asynchronous frame
// -- capacity = every instance behind the near budget, almost always unused -- launches at most GATED_GRID_MAX
// workgroups per kernel, so that an idle chain costs a few hundred workgroup launches instead of ~40 000 that compete
// with the other stream's kernels for wave slots.  `f(unit)` may return early (workgroup-uniformly).
constexpr int GATED_GRID_MAX = 1024;
template <typename F>
__device__ __forceinline__ void for_each_unit(const int nunits, F&& f) {
  for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
    f(unit);
    __syncthreads();  // (the unit's LDS is reused by the next one)
  }
}
__host__ __device__ inline int units_of(int n, int per) { return (int)(((long long)n + per - 1) / per); }
// workgroups to launch for a chain's kernel whose work unit covers `per` instances
inline int chain_grid(const Count& c, int per) {
  const int u = units_of(c.cap, per);
  return (c.gate || c.bounded) && u > GATED_GRID_MAX ? GATED_GRID_MAX : u;
}
struct FrameParams {
  int P, D, M, W, H, gx, gy;
  float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
  int ref_rects;  // 1 = emit the reference's full 3-sigma tile square (gsr_set_reference_rects), 0 = footprint-culled
};

// ---- stage launchers (each enqueues on `s`; returns hipGetLastError()) ----
bool preprocess_counts_depth_digits(const FrameParams& fp, const float* shs, const float* colors_precomp);
hipError_t launch_preprocess(const FrameParams& fp, const float* means3D, const float* scales, const float* rotations,
                             const float* opacities, const float* shs, const float* cov3D_precomp,
                             const float* colors_precomp, const float* view, const float* proj, const float* campos,
                             GeomState g, int* radii_out, bool write_cov3D, unsigned long long* done_word,
                             unsigned long long* publish, uint32_t ticket, uint32_t* ghist_acc, uint32_t* ghist_clear,
                             hipStream_t s);
hipError_t launch_point_offsets(const FrameParams& fp, GeomState g, hipStream_t s);
// (order: the depth-sorted ids the scan walks -- g.order, or the sorted near candidates of a partial depth sort, of which
// only the first "near limit" entries exist)
hipError_t launch_scan_offsets(const FrameParams& fp, GeomState g, Count R, uint32_t* chunk_first, uint2* ranges,
                               uint2* rangesB, uint32_t* counts0, uint32_t near_budget,
                               unsigned long long* publish_near, uint32_t ticket, const uint32_t* top_hist,
                               const uint32_t* order, bool order_is_near_list, hipStream_t s);
// Partial depth sort: the near candidates (depth keys whose top byte lies in the groups the near budget can reach,
// from top_hist) are compacted, in id order, into (keys_out, vals_out); their number goes to *n_out, their digit
// counts to ghist_near.
hipError_t launch_compact_near(const FrameParams& fp, GeomState g, const uint32_t* top_hist, uint32_t near_budget,
                               uint32_t* keys_out, uint32_t* vals_out, uint32_t* n_out, uint32_t* ghist_near,
                               unsigned long long* publish, uint32_t ticket, hipStream_t s);
hipError_t launch_clear_words(Count gate, uint32_t* words, size_t n, hipStream_t s);
hipError_t launch_scan_offsets_far(const FrameParams& fp, GeomState g, Count capB, uint32_t slot_base, const uint32_t* sat,
                                   uint32_t* chunk_firstB, uint32_t* counts0, unsigned long long* publish,
                                   uint32_t ticket, hipStream_t s);
// (digit_shift0 / digit_mask0: the digit of the tile sort's first pass, whose per-sort-tile counts the emitter leaves in
// counts0 -- the low bits of the tile id for the LSD sort, its top eight bits for the bucket sort, see tile_sort_buckets)
hipError_t launch_emit(const FrameParams& fp, const uint4* sdesc, Count R, uint32_t* chunk_first, uint32_t* tkeys_out,
                       uint32_t* ivals_out, uint8_t* inst_flag, uint32_t* counts0, uint32_t digit_shift0,
                       uint32_t digit_mask0, bool key16, bool store_pairs, hipStream_t s);
// what k_emit_scatter (the tile sort's first pass with the pairs generated in place) needs from the emitter's side
struct EmitFusion {
  FrameParams fp;
  const uint4* sdesc;  // the emitted Gaussians' descriptors in depth order (GeomState::sdesc, or sdescB in a far phase)
  Count R;
  const uint32_t* chunk_first;
};
hipError_t launch_emit_scatter(const EmitFusion& ef, uint16_t* keys_out, uint32_t* vals_out, int shift0, int nbits0,
                               const uint32_t* counts, const uint32_t* chunk_base, const uint32_t* digit_total,
                               bool arank, hipStream_t s);
// Bucket form of the 16-bit tile sort (radix_sort.hip): the first pass partitions the instances by the TOP eight bits of
// the tile id (stable), then ONE launch finishes every bucket on its own -- a workgroup counts its bucket's low digits,
// scatters it stably and writes the ranges of its tiles: the second pass's histogram, scan and range launches are gone
// (7 launches -> 4 per chain).  A bucket is one workgroup's serial job, so sorts beyond TILE_SORT_BUCKETS_MAX pairs
// (one-chain frames of dense 1080p scenes) keep the two LSD passes, as do 32-bit keys and single-pass sorts.
constexpr int TILE_SORT_BUCKETS_MAX = 16 << 20;  // (measured at 1080p: 2.6 M pairs 24 vs 46 us for the LSD second pass; 27.7 M pairs 176 vs ~150)
bool tile_sort_buckets(int tile_bits, bool key16, int capacity);
struct BucketPass {
  uint2* ranges;        // the chain's tile ranges (written in full: (0, 0) for tiles without instances)
  int tiles;
  uint32_t list_base;   // first list position of the chain (a far chain's lists start behind the near capacity)
};
// (split_frame: a near/far frame -- the Gaussians with records are looked for among the emitted ones' descriptors)
hipError_t launch_gather_records(const FrameParams& fp, GeomState g, BinningState b, float* dL_dmean2D,
                                 float* dL_dconic, float* dL_dopacity, float* dL_dcolor, bool split_frame,
                                 hipStream_t s);
// Stable LSD radix sort of n (u32, u32) pairs on key bits [0, end_bit); buffers ping-pong between
// (keysA, valsA) and (keysB, valsB), starting in A when start_in_A.
hipError_t launch_sort_pairs(uint32_t* keysA, uint32_t* valsA, uint32_t* keysB, uint32_t* valsB, SortScratch sc,
                             Count n, int end_bit, bool start_in_A, bool is_depth_sort, bool key16,
                             bool first_hist_done, const EmitFusion* fused_first_pass, uint32_t* key_count,
                             hipStream_t s, const BucketPass* buckets = nullptr);
// (n.dev != nullptr: the pair count is read on the device -- the near sort of a partial depth sort, whose tile size is
// that of a sort of n.cap pairs so that both use the scratch alike; n.gate: the gated full sort of an asynchronous
// frame's far chain)
// vals_are_positions: the values of the unsorted pairs are 0 .. n-1 and are not read (the Gaussians' ids: k_preprocess
// writes no id array)
hipError_t launch_depth_sort(uint32_t* keysA, uint32_t* valsA, uint32_t* keysB, uint32_t* valsB, DepthSortScratch sc,
                             Count n, const uint32_t* ghist, bool vals_are_positions, hipStream_t s);
hipError_t launch_ranges_from_counts(uint2* ranges, int T, uint32_t list_base, Count gate, hipStream_t s);
hipError_t launch_verify_sorted_lists(const uint2* ranges, int T, const uint32_t* point_list, const float4* splats,
                                      uint32_t* violations, hipStream_t s);
hipError_t launch_tile_ranges(const uint32_t* tile_ids, Count R, uint2* ranges, bool key16, uint32_t list_base, hipStream_t s);
// (decide / go / seq: asynchronous near/far frames, phase 1 -- the near blend's last workgroup stores
// 2 seq + (quads left unfinished ? 1 : 0) into *decide and, if none is, seq into *go; gate: phase 2)
struct AsyncWords {
  uint32_t* decide = nullptr;    // signal memory: what the far chain's stream waits for
  uint32_t* go = nullptr;        // signal memory: what the caller's stream waits for
  uint32_t* gate_dev = nullptr;  // device memory: the decision again, for the gates (Count::closed)
  uint32_t seq = 0;
};
hipError_t launch_blend_forward(const FrameParams& fp, GeomState g, BinningState b, ImageState im, const float* bg,
                                float* out_color, float* out_depth, float* out_acc, int phase,
                                unsigned long long* done_word, unsigned long long* publish, uint32_t ticket,
                                AsyncWords aw, Count gate, hipStream_t s);
hipError_t launch_live_sat(const FrameParams& fp, ImageState im, uint32_t* total_live, Count gate, hipStream_t s);
hipError_t launch_release_go(Count gate, uint32_t* go, uint32_t seq, hipStream_t s);
// (behind an asynchronous frame's near blend, same stream: opens the far chain if *live_quads != 0)
hipError_t launch_decide_far(const uint32_t* live_quads, AsyncWords aw, hipStream_t s);
hipError_t launch_tile_order(const FrameParams& fp, ImageState im, hipStream_t s);
hipError_t launch_blend_backward(const FrameParams& fp, GeomState g, BinningState b, ImageState im, const float* bg,
                                 const float* dL_dpix, const float* dL_dacc, bool have_tile_order, hipStream_t s);
hipError_t launch_gaussian_backward(const FrameParams& fp, GeomState g, BinningState b, const int* radii,
                                    const float* means3D, const float* scales, const float* rotations,
                                    const float* shs, const float* cov3D_precomp, const float* view, const float* proj,
                                    const float* campos, bool colors_precomp, float* dL_dmean2D, float* dL_dconic,
                                    float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                                    float* dL_dsh, float* dL_dscale, float* dL_drot, hipStream_t s);
hipError_t launch_mark_visible(int P, const float* means3D, const float* view, unsigned char* present, hipStream_t s);
// optimizer.hip ("next" row: fused activations + Adam)
hipError_t launch_activate(int P, int M, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                           const float* f_dc, const float* f_rest, float* scales, float* rotations, float* opacities,
                           float* shs, hipStream_t s);
hipError_t launch_activate_backward(int P, int M, const float* rotation_raw, const float* scales,
                                    const float* opacities, const float* g_scales, const float* g_rot,
                                    const float* g_opac, const float* g_shs, float* g_scaling_raw,
                                    float* g_rotation_raw, float* g_opacity_raw, float* g_f_dc, float* g_f_rest,
                                    hipStream_t s);
// loss.hip ("next" row: fused L1 + SSIM photometric loss)
size_t loss_workspace_bytes(int C, int H, int W);
hipError_t launch_photometric_loss(int C, int H, int W, const float* img, const float* gt, const float* window11,
                                   float lambda, float* loss_out3, float* dL_dimg, char* workspace, hipStream_t s);
hipError_t launch_init_gaussians(int n, int M, const float* xyz, const float* covs, const float* rgbs, float scale_factor,
                                 float* xyz_out, float* fdc_out, float* frest_out, float* scaling_out,
                                 float* rotation_out, float* opacity_out, hipStream_t s);
hipError_t launch_pack_ply_rows(int P, int M, const float* xyz, const float* fdc, const float* frest,
                                const float* opacity, const float* scaling, const float* rotation, float* rows,
                                hipStream_t s);
hipError_t launch_model_step(int P, int M, float* const* params, float* const* exp_avg, float* const* exp_avg_sq,
                             const float* g_xyz, const float* g_scales, const float* g_rot, const float* g_opac,
                             const float* g_shs, float* a_scales, float* a_rot, float* a_opac, float* a_shs,
                             const float* lr, double beta1, double beta2, double eps, int step, hipStream_t s);
hipError_t launch_adam(int n, float* const* params, float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const size_t* numel, const float* lr, double beta1, double beta2,
                       double eps, int step, int zero_grads, hipStream_t s);

inline int sort_passes(int end_bit) { return (end_bit + 7) / 8; }
// digit width: the key bits are split evenly over the passes (13 tile bits -> 7 + 6, 32 depth bits -> 4 x 8)
inline int sort_digit_bits(int end_bit) { const int p = sort_passes(end_bit); return (end_bit + p - 1) / p; }

// Kernel ids for the optional event profiler (api.hip); order = gsr_kernel_name().
enum KernelId {
  K_PREPROCESS = 0, K_POINT_OFFSETS, K_SCAN_OFFSETS, K_EMIT, K_SORT_HIST,
  K_SORT_SCAN_CHUNKS, K_SORT_SCAN_TOP, K_SORT_SCATTER, K_TILE_RANGES, K_BLEND_FWD, K_BLEND_BWD, K_COMPACT_TOUCHED,
  K_GATHER_RECORDS, K_GAUSSIAN_BWD, K_MARK_VISIBLE, K_DSORT_HIST, K_DSORT_SCAN_CHUNKS, K_DSORT_SCAN_TOP,
  K_DSORT_SCATTER, K_ACTIVATE, K_ACTIVATE_BWD, K_ADAM, K_LOSS_FWD, K_LOSS_FINALIZE, K_LOSS_BWD, K_INIT_GAUSSIANS, K_PACK_PLY, K_MODEL_STEP, K_TILE_ORDER, K_LIVE_SAT, K_COMPACT_NEAR, K_COUNT
};
void prof_begin(int id, hipStream_t s);
void prof_end(hipStream_t s);
extern std::atomic<bool> g_prof_on;
extern std::atomic<unsigned long long> g_prof_mask;  // bit i: kernel id i is recorded
extern thread_local bool t_prof_suppress;             // launches that are not timed (an asynchronous frame's gated chain)
struct ProfScope {  // records a start/stop event pair around the launches in its scope while profiling is on
  hipStream_t s;
  bool on;
  ProfScope(int id, hipStream_t st)
      : s(st), on(!t_prof_suppress && g_prof_on.load(std::memory_order_relaxed) &&
                  ((g_prof_mask.load(std::memory_order_relaxed) >> id) & 1ull)) {
    if (on) prof_begin(id, s);
  }
  ~ProfScope() { if (on) prof_end(s); }
};

// getRect (reference auxiliary.h:39-46): tile rectangle [x0,x1) x [y0,y1) touched by a splat of
// integer pixel radius `radius` centred at (px,py); float arithmetic with truncating casts, exactly
// as the reference so the rectangle (and hence every sort key) matches bit for bit.  Only
// add/sub and a division by 16 are involved, so FMA contraction cannot change the result.
// Upper bound of ln(x) for normal x > 0 from +, *, / only (IEEE, no contraction in this file), so the CPU
// oracle reproduces it bit for bit -- unlike a hardware log -- and the culled tile rectangles stay an
// exact-match quantity: ln x = e ln 2 + 2 atanh((m-1)/(m+1)), series cut after t^5 (remainder < 1.6e-4).
__device__ __forceinline__ float ln_upper(float x) {
  const uint32_t b = __float_as_uint(x);
  const int e = (int)(b >> 23) - 127;
  const float m = __uint_as_float((b & 0x007FFFFFu) | 0x3F800000u);  // [1, 2)
  const float t = (m - 1.0f) / (m + 1.0f);
  const float t2 = t * t;
  const float s = t * (2.0f + t2 * (0.6666667f + t2 * 0.4f));
  return (float)e * 0.6931472f + s + 3e-4f;
}

// Footprint threshold of a splat of opacity op: a pixel can only receive alpha = min(.99, op e^power) >= 1/255 if
// -power <= ln(255 op); tau bounds that from above with slack for the rounding of the blend kernels' own power /
// exp arithmetic (1 % + 0.02).  Used by k_preprocess (footprint box, tile culling) and by the blend kernels
// (exact ellipse-vs-quad test).
__device__ __forceinline__ float footprint_tau(float op) { return ln_upper(255.0f * op) * 1.01f + 0.02f; }

__host__ __device__ inline void tile_rect(float px, float py, int radius, int gx, int gy, int& x0, int& y0, int& x1,
                                          int& y1) {
  const float r = (float)radius;
  int a = (int)((px - r) / 16.0f), b = (int)((py - r) / 16.0f);
  int c = (int)((((px + r) + 16.0f) - 1.0f) / 16.0f), d = (int)((((py + r) + 16.0f) - 1.0f) / 16.0f);
  a = a < 0 ? 0 : a; b = b < 0 ? 0 : b; c = c < 0 ? 0 : c; d = d < 0 ? 0 : d;
  x0 = a > gx ? gx : a; y0 = b > gy ? gy : b; x1 = c > gx ? gx : c; y1 = d > gy ? gy : d;
}

}  // namespace gsr
