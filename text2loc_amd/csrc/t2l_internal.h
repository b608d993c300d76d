// Internal declarations shared by the HIP translation units of libt2l.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "t2l.h"

namespace t2l {

constexpr int kD = T2L_EMBED_DIM;   // 256
constexpr int kS = T2L_OBJECT_SIZE; // 28 object slots per cell
constexpr int kSP = 32;             // slots padded to one 32-row MFMA tile

// ---- search geometry -------------------------------------------------------------------------
constexpr int kTileRows = 32;                      // DB rows per LDS tile (one 32x32 MFMA row block)
constexpr int kRowStrideF = kD + 4;                // LDS row stride in floats (1040 B): ds_read_b128 conflict-free
constexpr int kTileFloats = kTileRows * kRowStrideF;
constexpr int kQPerWave = 32;
constexpr int kScanWaves = 4;
constexpr int kQPerBlock = kQPerWave * kScanWaves; // 128 queries per workgroup
constexpr int kStageCap = 32;                      // staged (score,row) slots per lane between compactions
constexpr int kMaxParts = 64;                      // per-query candidate partitions (= 2 * nsplit) the re-rank merges
constexpr int kMaxPerTiles = 512;                  // tiles per split of one scan launch (13 key code bits)
constexpr int kSegmentRows = (kMaxParts / 2) * kMaxPerTiles * 32;  // rows one scan launch covers (524,288); also the unit inside which the f16 plane deals rows to tiles strided

// Per-kernel timing: a ring of hipEvent pairs recorded on the caller's stream (no sync when recording);
// t2l_kernel_stats() reads them back after the caller's own synchronisation point.
constexpr int kEventRing = 512;
constexpr int kSpanRing = 64;   // in-kernel stamps of the last 64 paired-scan launches (search.hip) ...
constexpr int kSpanWgs = 512;   // ... of up to 512 workgroups each (larger grids are not stamped)
struct EventRing {
  std::vector<hipEvent_t> a, b;
  int head = 0;   // next slot
  int count = 0;  // recorded since the last read (saturates at kEventRing)
  int calls = 0;  // launches seen (option profile_events = n records every n-th)
  bool open = false;
};

struct EncoderWeights;
int search_stream_impl(t2l_ctx* ctx, const float* q, int Q, int K, int32_t* out_idx, double* out_score, hipStream_t s);
int pack_impl(t2l_ctx* ctx, const int32_t* idx, const double* score, int n, double* pairs, hipStream_t s);
int merge_pairs_impl(t2l_ctx* ctx, const double* pairs, int parts, int Q, int K, int32_t* out_idx, double* out_score,
                     hipStream_t s);
int merge_gathered_impl(t2l_ctx* ctx, const void* blocks, int64_t block_bytes, int64_t score_offset, int parts, int Q, int K,
                        int32_t* out_idx, double* out_score, hipStream_t s);
// reduce.hip
int reduce_impl(t2l_ctx* ctx, const float* xyz, const float* rgb, const int64_t* offsets, int n_objects,
                const float* centers, const int32_t* rows, int n_colors, float* out_rgb, float* out_center,
                float* out_npts, int32_t* out_color, hipStream_t s);
// encode.hip

}  // namespace t2l

struct t2l_ctx {
  int device = 0;
  std::string err;
  // database shard
  float* db = nullptr;       // [db_pad,256], rows >= db_rows are zero
  uint4* db_split = nullptr; // bf16 [db_pad][hi 256 | lo 256] planes of the same rows (1 KiB per row)
  int64_t db_rows = 0, db_pad = 0, db_cap = 0, row_offset = 0;
  uint4* db_half = nullptr;  // f16 [db_pad][256] plane of the same rows, scaled by a power of two (512 B per row)
  float* db_norm_max = nullptr;  // dev f32[2]: max row 2-norm (feeds the certificate's error bound), max |element|
  // search workspace
  float* cand_score = nullptr;   // candidate keys [Q][2*nsplit][L]
  int32_t* flags = nullptr;      // dev i32[Q]: 1 = first-stage certificate failed -> fallback kernel
  int32_t* fb_count = nullptr;   // dev i32[128] (2 used): [0] exact-scan fallbacks, [1] stage-2 re-scores of the last search
  int32_t* fb_prev = nullptr;    // the other bank of the same allocation (search.hip: reset_counts); the host swaps the two per call
  int32_t* seg_idx = nullptr;    // per-segment results when the shard exceeds one scan launch
  double* seg_score = nullptr;
  size_t cand_cap = 0, flag_cap = 0, seg_idx_cap = 0, seg_score_cap = 0;  // bytes
  void* loss_ws = nullptr;       // [B][B] matrix + row vectors of the contrastive loss beyond 128 rows (loss.hip)
  size_t loss_ws_cap = 0;
  void* reduce_ws = nullptr;     // work items + float64 accumulators of t2l_reduce_objects
  size_t reduce_ws_cap = 0;
  // encoder
  t2l::EncoderWeights* enc = nullptr;
  void* train = nullptr;         // t2l::TrainState (train.hip)
  void* pn = nullptr;            // t2l::PointNetWeights (pointnet.hip), null when no pointnet.* tensors were loaded
  void* fine = nullptr;          // t2l::FineWeights (fine.hip)
  void* text_head = nullptr;     // t2l::th::Weights (text_head.hip)
  void* text_train = nullptr;    // t2l::TextTrain (train.hip): the text head's training state
  // cross-rank BatchNorm statistics (t2l_train_sync_bn): the accumulator slots live in the caller's buffer and the callback sums a
  // range of it over the ranks, ordered on the stream it is handed
  int (*sync_fn)(void* user, double* buf, int64_t n, void* stream) = nullptr;
  void* sync_user = nullptr;
  double* sync_buf = nullptr;
  bool sync_failed = false;
  void* fast_ws = nullptr;       // operand planes of fast_gemm (text_head.hip)
  size_t fast_ws_cap = 0;
  float* fast_zero = nullptr;
  int text_head_rows = 0;        // token rows per pass of the text head (0 = default 16,384)
  int pn_self_loops = 1;         // PyG PointConv add_self_loops quirk on the bipartite batch (oracle/t2l_oracle_pointnet.py)
  // options
  double eps_scale = 1.0;
  int nsplit_override = 0;
  int search_mode = 0;   // 0 = f16 MFMA scan (default), 1 = exact-f32 MFMA scan, 2 = split-bf16 MFMA scan
  // mode 0 watches how many queries of a batch its certificate sends to the second stage (the fallback kernel writes the
  // count to mapped host memory; no stream operation, no synchronisation) and, when that is more than one in eight —
  // scores packed tighter than the f16 error band — searches with the split-bf16 scan (50x tighter bound) until fewer than
  // one in sixteen would be flagged again
  int encoder_f32 = 0;   // 1: the all-f32-MFMA encoder kernel even when the split-f16 one is safe (encode.hip)
  int encoder_f16 = 0;   // 1: plain-f16 products (one MFMA per operand pair) instead of split-f16: ~1e-4 instead of 2e-7, 28 % faster
  int search_auto = 1;
  int pair_ll = 6;       // per-lane list length of the paired scan (5 or 6)
  int search_tile_sel = 1;  // paired scan with the tile-local top-3 selection (scanp_kernel<..., SEL = 1>; merged records only)
  int search_small = 1;      // batches of <= 16 queries against <= 65,536 rows: the one-launch exact float64 search (search_small.hip)
  int search_small_wgs = 0;  // ... its workgroups per 4-query slice (0 = by query count: 128 for Q <= 2 or Q > 8, else 192)
  bool last_search_small = false;  // the last search ran the one-launch path: t2l_search_fallbacks answers 0 (it leaves the counters alone)
  unsigned* small_ticket = nullptr;       // dev u32[4]: arrival tickets per slice, running totals
  unsigned small_ticket_base[4] = {0, 0, 0, 0};
  void* small_part = nullptr;             // published per-workgroup top-K lists {f64 score | i32 row}
  size_t small_part_cap = 0;
  int wide_repair = 512;  // rows a re-rank wave may re-score in a wide repair before the query goes to an exact scan (0: never)
  int encoder_two_cells = 1;  // encode_cells: two cells per eight-wave workgroup on LDS planes (encode.hip: encode_cells2_kernel); 0: first form
  int search_merge = 2;    // the paired scan merges a workgroup's four lists per query into one 32-byte record (search.hip: MERGE / MG):
                           // 0 never, 1 always, 2 while the f16 report cards show next to no failed first certificates (a repair behind
                           // a merged record re-scores 4x the rows of a plain list's)
  bool merge_live = true;  // (search_merge == 2) what the report cards say right now
  int xcd_qgroups = 4;   // paired scan: query-block groups per XCD rectangle (1 = every XCD sees all queries and 1/8 of the splits;
                         // 4 = a quarter of the queries and half of the splits: -1.3 us of scan span at Q = 4096 x N = 11,259, measured)
  int train_bf16 = 0;       // 1: the training step's GEMMs round their operands to bf16 (one bf16 MFMA per 16-step); 2: split-bf16 (three)
  int text_train_bf16 = 2;  // the same for the TEXT head's training GEMMs (d_model 1024: 466 GFLOP per step at B = 64): default split-bf16 —
                            // f32-class products (<= 2^-16 + 2^-18 relative, f32 accumulation, f32 exponent range) at 1.8x the f32 MFMA path's speed
  int train_gemm_block = 0;   // output block of the training step's tile GEMMs: 64 (2 x 2 tiles per wave: half the operand traffic), 32, or
                              // 0 = by measurement: 64 with bf16 / split-bf16 operands (0.555 -> 0.533 ms per step), 32 in f32 (0.640 vs 0.660)
  int train_keep_adam = 0;  // 1: t2l_train_bind keeps Adam moments + step when the parameter list is unchanged (a re-bind)
  int eff_mode = 0;           // the scan the current t2l_search call runs
  bool heavy = false;         // the database defeats the certificates: flagged queries go to the float64 MFMA stage
  bool all_exact = false;     // ... and nearly all of them: EVERY query goes there, no candidate scan (search_impl)
  unsigned all_exact_calls = 0;
  bool escalated = false;     // the split-bf16 scan is standing in (it counts what the f16 band would still flag)
  int32_t* host_stat = nullptr;      // mapped pinned host int32[8]: {sequence number of the last finished call, flagged, Q, previous exact-stage count, f16 stat, first-certificate failures}
  int32_t* host_stat_dev = nullptr;  // its device address
  int stat_seq = 0, stat_seen = 0;
  // Pipelined searches (option "search_lanes" = n > 1): consecutive t2l_search calls are independent jobs, so call i runs its
  // scan -> re-rank chain on internal stream i % n with that lane's own scratch set; the chains overlap on the GPU (the next
  // scan's workgroups start while the previous call re-ranks; no kernel-boundary bubble between calls). Results are ordered
  // into the caller's stream by t2l_search_join.
  struct SearchLane {
    float* cand_score = nullptr;
    size_t cand_cap = 0, flag_cap = 0;
    int32_t *flags = nullptr, *fb_count = nullptr, *fb_prev = nullptr, *host_stat = nullptr, *host_stat_dev = nullptr;
    int stat_seen = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    bool pending = false, ready = false;
  };
  static constexpr int kMaxLanes = 4;
  SearchLane lanes[kMaxLanes];
  int n_lanes = 1;
  unsigned lane_next = 0;
  hipEvent_t lane_fork = nullptr;
  int stream_min_rows = 65536;  // shards at least this large answer batches of <= 64 queries with the streaming scan
  unsigned long long* scan_span = nullptr;  // dev u64[kSpanRing][kSpanWgs][2]: per workgroup {seq << 40 | start tick, seq << 40 | end tick}
  unsigned* span_grid = nullptr;            // host u32[kSpanRing]: grid of the launch in each ring entry (0: not stamped)
  unsigned span_seq = 0, span_read = 0, busy_read = 0;
  int profile_rerank = 1;  // 0: sampled launches bracket the scan only (an event pair costs the stream ~6 us per kernel)
  int profile_events = 0;  // 0 off, n >= 1: record every n-th launch of each kernel
  std::unordered_map<std::string, t2l::EventRing> events;
};

namespace t2l {

int fail(t2l_ctx* ctx, int code, const std::string& msg);

#define T2L_HIP(ctx, expr)                                                                     \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return ::t2l::fail((ctx), T2L_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

void event_begin(t2l_ctx* ctx, const char* name, hipStream_t s);
void event_end(t2l_ctx* ctx, const char* name, hipStream_t s);
bool event_pair(t2l_ctx* ctx, const char* name, hipEvent_t* a, hipEvent_t* b);

// search.hip
int search_impl(t2l_ctx* ctx, const float* q, int Q, int K, int32_t* out_idx, double* out_score, hipStream_t s);
int db_norm_impl(t2l_ctx* ctx, hipStream_t s);
int merge_impl(t2l_ctx* ctx, const int32_t* idx, const double* score, int parts, int Q, int K, int32_t* out_idx,
               double* out_score, hipStream_t s);
int search_stream_impl(t2l_ctx* ctx, const float* q, int Q, int K, int32_t* out_idx, double* out_score, hipStream_t s);
// search_small.hip
bool search_small_applies(const t2l_ctx* ctx, int Q, int K);
int search_small_impl(t2l_ctx* ctx, const float* q, int Q, int K, int32_t* out_idx, double* out_score, hipStream_t s);
// search_exact.hip
int exact_stage_impl(t2l_ctx* ctx, const float* db, int n_rows, int row_offset, const float* q, int Q, int K, int32_t* out_idx,
                     double* out_score, hipStream_t s);
// encode.hip
int load_weights_impl(t2l_ctx* ctx, const t2l_weight_desc* w, int n, const t2l_model_config* cfg);
int encode_impl(t2l_ctx* ctx, const t2l_packed_cells* in, float* out, hipStream_t s);
void free_weights(t2l_ctx* ctx);
// train.hip
int train_bind_impl(t2l_ctx* ctx, const t2l_train_tensor* tensors, int n, const t2l_model_config* cfg);
int train_forward_impl(t2l_ctx* ctx, const t2l_packed_cells* in, float p, uint32_t seed, float* out_emb, hipStream_t s);
int train_backward_impl(t2l_ctx* ctx, const float* grad_emb, float* grad_pn_feat, hipStream_t s);
int adam_step_impl(t2l_ctx* ctx, float lr, float b1, float b2, float eps, hipStream_t s);
int zero_grad_impl(t2l_ctx* ctx, hipStream_t s);
int pn_train_forward_impl(t2l_ctx* ctx, const float* pos, const float* rgb, const int32_t* cell_offsets, int n_cells, float* out_f2,
                          hipStream_t s);
int pn_train_backward_impl(t2l_ctx* ctx, const float* grad_f2, hipStream_t s);
int search_lanes_impl(t2l_ctx* ctx, const float* q, int Q, int K, int32_t* out_idx, double* out_score, hipStream_t s);
int search_join_impl(t2l_ctx* ctx, hipStream_t s);
void free_lanes(t2l_ctx* ctx);
int adam_state_impl(t2l_ctx* ctx, int set, float* m, float* v, int64_t* step, int64_t* numel, hipStream_t s);
void free_train(t2l_ctx* ctx);
// t2l_text_inter as ONE launch (encode.hip: text_inter_fused_kernel — the cell encoder's per-tile transformer layer at S sentences per
// description): the inter layer's matrices in the encoder's split-f16 fragment packing (mfma_h3.h: pack_split_f16) + its f32 vectors
struct InterFusedW {
  const uint4 *in_hp = nullptr, *out_hp = nullptr, *ff1_hp = nullptr, *ff2_hp = nullptr;
  const float *in_b = nullptr, *out_b = nullptr, *ff1_b = nullptr, *ff2_b = nullptr, *ln1_w = nullptr, *ln1_b = nullptr, *ln2_w = nullptr,
              *ln2_b = nullptr;
};
int text_inter_fused_launch(t2l_ctx* ctx, const InterFusedW& W, bool single, const float* sent, int n_desc, int S, float* out, int* flag,
                            hipStream_t s);
void free_text_train(t2l_ctx* ctx);
int text_adam_step_impl(t2l_ctx* ctx, float lr, float b1, float b2, float eps, hipStream_t s);
int text_zero_grad_impl(t2l_ctx* ctx, hipStream_t s);
int text_adam_state_impl(t2l_ctx* ctx, int set, float* m, float* v, int64_t* step, int64_t* numel, hipStream_t s);
int64_t train_sync_bn_doubles();
void train_sync_changed(t2l_ctx* ctx);  // the accumulator slots moved: the next forward clears them
int text_train_bind_impl(t2l_ctx* ctx, const t2l_train_tensor* tensors, int n, const char* prefix);
int text_train_forward_impl(t2l_ctx* ctx, const float* hidden, int n_sent, int L, int n_desc, float p, uint32_t seed, float* out, hipStream_t s);
int text_train_backward_impl(t2l_ctx* ctx, const float* grad_out, hipStream_t s);
// pointnet.hip
int pointnet_load_impl(t2l_ctx* ctx, const t2l_weight_desc* w, int n);
int pointnet_features_impl(t2l_ctx* ctx, const float* pos, const float* rgb, const int32_t* cell_offsets, int n_cells, float* out,
                           hipStream_t s);
int sample_points_impl(t2l_ctx* ctx, const float* xyz, const float* rgb, const int64_t* point_offsets_dev, int n_objects, uint32_t seed,
                       int flags, float rotate_deg,
                       float* out_pos, float* out_rgb, hipStream_t s);
void free_pointnet(t2l_ctx* ctx);
// fine.hip
int fine_load_impl(t2l_ctx* ctx, const t2l_weight_desc* w, int n, const t2l_model_config* cfg);
int fine_encode_impl(t2l_ctx* ctx, const t2l_packed_cells* in, float* out, hipStream_t s);
int fine_match_impl(t2l_ctx* ctx, const float* cell_desc, const int32_t* cell_index, const float* hint_desc, const int32_t* hint_index,
                    int n_pairs, int n_hints, float* out, hipStream_t s);
void free_fine(t2l_ctx* ctx);
// text_head.hip
int text_head_load_impl(t2l_ctx* ctx, const t2l_weight_desc* w, int n, const char* prefix);
int text_head_impl(t2l_ctx* ctx, const float* hidden, int n_sentences, int n_tokens, float* out, int32_t* overflow, hipStream_t s);
int fast_gemm(t2l_ctx* ctx, const float* A, bool a_trans, const float* B, bool b_trans, const float* bias, float* out, int Mo, int No, int Kc,
              int relu, int accumulate, bool single, hipStream_t s, float* a_colsum = nullptr);  // text_head.hip: the tiled bf16-plane GEMM for training
void fast_colsum(const float* a, int M, int N, float* out, hipStream_t s);
int text_inter_impl(t2l_ctx* ctx, const float* sent, int n_desc, int S, float* out, int32_t* overflow, hipStream_t s);
void free_text_head(t2l_ctx* ctx);
// loss.hip
// hipFuncSetAttribute applies to the CURRENT device's instance of a kernel, and one process may hold contexts on several GPUs: a call
// site's "attribute set" flag is one bit per device (the callers are serialised per context, as everything in this library).
struct PerDeviceOnce {  // (static instances are shared by the contexts of every device and thread: atomic bits; a lost race repeats
  std::atomic<uint64_t> done{0};  //  an idempotent hipFuncSetAttribute)
  bool need(int dev) const { return !((done.load(std::memory_order_acquire) >> (dev & 63)) & 1); }
  void mark(int dev) { done.fetch_or(1ull << (dev & 63), std::memory_order_release); }
};

int loss_impl(t2l_ctx* ctx, const float* a, const float* p, int B, float temp, float* loss, float* ga, float* gp,
              hipStream_t s);

}  // namespace t2l
