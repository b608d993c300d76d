/*
 * t2l.h — C ABI of the MI355X-native coarse-retrieval engine for Text2Loc (libt2l.so).
 *
 * The reference (Yan-Xia/Text2Loc) has NO plugin/operator/FFI layer: its boundary for this path is the
 * duck-typed Python surface consumed by training/coarse.py:63-157 (eval_epoch) and
 * evaluation/pipeline.py:41-87 (run_coarse). This header is the library a binding for that surface
 * loads; text2loc_amd/engine.py is the ctypes binding, text2loc_amd/cell_retrieval.py and
 * text2loc_amd/coarse.py mirror the reference's Python names on top of it.
 *
 * Conventions
 *   - every function returns 0 on success, a negative T2L_E* code on error; the message is available
 *     from t2l_last_error(ctx). Nothing throws across the boundary.
 *   - "dev" pointers are device (HBM) pointers on the context's GPU, "host" pointers are host memory.
 *     All buffers are caller-allocated and caller-owned; the library copies what it keeps.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream). Calls are asynchronous on
 *     that stream unless stated otherwise. A context is not thread-safe: the reference is a single
 *     thread / single process per device, and so is this (one process per GPU; shards meet in RCCL).
 *   - row ids are int32 (N < 2^31) and GLOBAL: local row + the shard's row_offset (t2l_db_set).
 *   - embed dim is fixed at 256 (coarse_embed_dim of the published config, README.md:87-99).
 */
#ifndef T2L_H_
#define T2L_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2L_ABI_VERSION 2
#define T2L_EMBED_DIM 256
#define T2L_OBJECT_SIZE 28 /* args.object_size, models/cell_retrieval.py:32 */
#define T2L_MAX_LOSS_BATCH 1024 /* rows of the contrastive matrix (one fused launch up to 128, a short chain beyond) */
#define T2L_MAX_TOPK 26    /* max(top_k) supported by the fused search (eval default is 10) */

enum {
  T2L_OK = 0,
  T2L_EINVAL = -1,  /* bad argument (shape, K, null pointer, unknown name) */
  T2L_ESTATE = -2,  /* call order: weights / database not loaded */
  T2L_EHIP = -3,    /* a HIP runtime call failed (message carries hipGetErrorString) */
  T2L_ENOMEM = -4
};

typedef struct t2l_ctx t2l_ctx;

/* ---- lifecycle ------------------------------------------------------------------------------- */
int t2l_abi_version(void);
/* Replaces: model.to(device) (evaluation/pipeline.py:252). One context per GPU. */
int t2l_create(t2l_ctx** out, int device_id);
void t2l_destroy(t2l_ctx* ctx);
const char* t2l_last_error(const t2l_ctx* ctx);

/* ---- weights --------------------------------------------------------------------------------- */
/* One fp32 tensor of the reference's checkpoint (torch.save'd state_dict, training/coarse.py:317-345),
 * named by its state_dict key, e.g. "obj_inter_module.0.self_attn.in_proj_weight". */
typedef struct {
  const char* name;
  const float* data; /* host, row-major, as stored by torch */
  int64_t numel;
} t2l_weight_desc;

/* Feature switches of the 3D-submap branch (training/args.py:39-63): */
typedef struct {
  int32_t class_embed;   /* args.class_embed: 1 = class_embedding lookup, 0 = PointNet++ features2 -> mlp_pointnet */
  int32_t color_embed;   /* args.color_embed: 1 = color_embedding lookup, 0 = color_encoder(mean rgb) */
  int32_t use_class, use_color, use_position, use_num; /* args.use_features membership */
  int32_t num_layers;    /* args.object_inter_module_num_layers (2) */
  int32_t num_heads;     /* args.object_inter_module_num_heads (4) */
} t2l_model_config;

/* Replaces: CellRetrievalNetwork.load_state_dict(strict=False) (evaluation/pipeline.py:251) for the
 * object branch. Folds eval-mode BatchNorm into the preceding Linear, re-lays weights out for the
 * kernels and uploads them. Synchronous. Keys of other sub-modules (language_encoder.*) are ignored; a missing
 * REQUIRED key is T2L_EINVAL (stricter than strict=False on purpose: silent zero weights would void parity).
 * "object_encoder.pointnet.*" tensors are optional as a group: when present they are folded and packed for
 * t2l_pointnet_features (the two classifier heads, unused on this path, are ignored). */
int t2l_load_weights(t2l_ctx* ctx, const t2l_weight_desc* w, int32_t n, const t2l_model_config* cfg);

/* ---- per-object reductions over raw points (a1) ------------------------------------------------ */
/* Replaces the host reductions the reference redoes inside every ObjectEncoder.forward call
 * (datapreparation/kitti360pose/imports.py:28-41 via models/object_encoder.py:79-84,121-141):
 *   out_rgb    = mean rgb of the object's points          Object3d.get_color_rgb
 *   out_color  = color_rows[argmin_k ||mean rgb - color_centers[k]||]   Object3d.get_color_text -> known_colors[...]
 *   out_center = mean xyz                                 Object3d.get_center
 *   out_npts   = number of points                         len(obj.xyz)
 * xyz, rgb: dev f32[n_points,3], the objects' points concatenated in object order; point_offsets: HOST i64[n_objects+1]
 * (the caller knows the object sizes; the library cuts objects of more than 4096 points into runs so that the long
 * tail of the size distribution does not serialise on one wave). Synchronises the stream once (upload of the run list).
 * color_centers host f32[n_colors,3] (utils.py:210-224), color_rows host i32[n_colors] (row of color_embedding per centre,
 * i.e. the reference's {name: i for i, name in enumerate(COLOR_NAMES)} applied to COLOR_NAMES), n_colors <= 16.
 * Outputs are dev buffers in exactly the t2l_packed_cells layout. Sums are accumulated in float64. */
int t2l_reduce_objects(t2l_ctx* ctx, const float* xyz, const float* rgb, const int64_t* point_offsets, int32_t n_objects,
                       const float* color_centers, const int32_t* color_rows, int32_t n_colors, float* out_rgb,
                       float* out_center, float* out_npts, int32_t* out_color_idx, void* stream);

/* ---- point batches for PointNet++ (a3, dataloader side) ------------------------------------------ */
/* Replaces: batch_object_points(objects, transform) (dataloading/kitti360pose/utils.py:91-147), which the reference runs per
 * object on the host inside the dataloader, for the three transforms its scripts compose:
 *   transform_flags = 0                                        FixedPoints(256)                      `--no_pc_augment`: EVERY published
 *       command (README.md:89,107,125-126,139-140; evaluation/pipeline.py:215-216, training/coarse.py:182-184) — positions stay in
 *       the cell-normalised frame, the ball-query radii 0.2 / 0.3 / 0.4 are absolute;
 *   T2L_SAMPLE_NORMALIZE                                       FixedPoints + NormalizeScale          evaluation without the flag
 *       (evaluation/pipeline.py:217-218, training/coarse.py:193): centred on the mean of the sample, scaled by 0.999999 / max |coordinate|;
 *   T2L_SAMPLE_ROTATE | T2L_SAMPLE_NORMALIZE, rotate_deg = 120 FixedPoints + RandomRotate(120, axis=2) + NormalizeScale   training
 *       without the flag (training/coarse.py:185-192): one angle per object, uniform in [-rotate_deg, +rotate_deg], about z.
 * xyz, rgb: dev f32[n_points,3] (objects concatenated); point_offsets: DEV i64[n_objects+1]; out_pos / out_rgb: dev
 * f32[n_objects,256,3] — the inputs of t2l_pointnet_features. Sampling is counter-based (the reference uses numpy's global RNG: equal
 * in distribution, not in the draw): index j of object o is floor(u * n_o) with u = (lowbias32(j * 0x9E3779B1 + key_o) >> 8) / 2^24,
 * key_o = seed ^ o * 0x85EBCA77; the angle is rotate_deg * (2u' - 1) with u' = (lowbias32(0xA5A5A5A5 + key_o) >> 8) / 2^24. */
#define T2L_SAMPLE_NORMALIZE 1
#define T2L_SAMPLE_ROTATE 2
int t2l_sample_object_points(t2l_ctx* ctx, const float* xyz, const float* rgb, const int64_t* point_offsets, int32_t n_objects,
                             uint32_t seed, int32_t transform_flags, float rotate_deg, float* out_pos, float* out_rgb, void* stream);

/* ---- PointNet++ object backbone (a3), eval mode ------------------------------------------------ */
/* Replaces: PointNet2.forward(...).features2 (models/pointcloud/pointnet2.py:66-100) as ObjectEncoder.forward calls it
 * once per cell (models/object_encoder.py:92-95): three SetAbstraction layers (FPS 1/2, ball query r = .2/.3/.4 with at most
 * 32 neighbours, PointConv = max over get_mlp(cat[x_j, pos_j - pos_i])), the global get_mlp + max, lin1/lin2 + ReLU.
 * pos, rgb: dev f32[n_objects,256,3] — each object's 256 points as the reference's dataloader delivers them (FixedPoints(256)
 * + NormalizeScale, dataloading/kitti360pose/utils.py:91-147); cell_offsets: HOST i32[n_cells+1] (objects per cell = per PyG
 * batch, offsets[0] = 0); out_features2: dev f32[n_objects,256], ready to be t2l_packed_cells.pn_feat.
 * PARITY UNPINNED: the arithmetic of this stage lives in torch_geometric / torch-cluster / torch-scatter, absent from the
 * reference tree. Semantics here (oracle/t2l_oracle_pointnet.py): FPS starts at point 0 (the reference: random start) and
 * breaks ties by lower index; the ball query keeps the first 32 source points in index order with d^2 < r^2; option
 * "pointnet_pyg_self_loops" (default 1) reproduces PyG 1.7's PointConv(add_self_loops=True) on bipartite inputs, which adds
 * the message of source node k of the cell's batch to centre k of the cell's batch. Synchronises the stream once. */
int t2l_pointnet_features(t2l_ctx* ctx, const float* pos, const float* rgb, const int32_t* cell_offsets, int32_t n_cells,
                          float* out_features2, void* stream);

/* ---- cell encoding (a2+a4) ------------------------------------------------------------------- */
/* Packed SoA replacement of List[List[Object3d]] (+ PointNet++ features2 when class_embed == 0).
 * Per object o of cell b (offsets[b] <= o < offsets[b+1], dataset order):
 *   class_idx  = known_classes.get(label, 0)            models/object_encoder.py:81
 *   color_idx  = known_colors[get_color_text()]          models/object_encoder.py:83
 *   rgb        = float32(mean rgb of the object's points)   imports.py:28-31, object_encoder.py:124-127
 *   center     = float32(mean xyz)                          imports.py:40-41, object_encoder.py:133-134
 *   n_pts      = float32(len(obj.xyz))                      object_encoder.py:141-143
 *   pn_feat    = PointNet2(...).features2 of the object  [n_objects,256] (NULL when class_embed)   */
typedef struct {
  int32_t n_cells;
  int32_t n_objects;
  const int32_t* offsets;   /* dev i32[n_cells+1] */
  const int32_t* class_idx; /* dev i32[n_objects] */
  const int32_t* color_idx; /* dev i32[n_objects] */
  const float* rgb;         /* dev f32[n_objects,3] */
  const float* center;      /* dev f32[n_objects,3] */
  const float* n_pts;       /* dev f32[n_objects] */
  const float* pn_feat;     /* dev f32[n_objects,256] or NULL */
} t2l_packed_cells;

/* Replaces: CellRetrievalNetwork.encode_objects (models/cell_retrieval.py:65-110) in eval mode.
 * out_emb: dev f32[n_cells,256], unit rows. Objects beyond the first 28 of a cell are ignored
 * (cell_retrieval.py:94-98); the zero pad slots take part in attention and in the max-pool, as in
 * the reference (no padding mask). */
int t2l_encode_cells(t2l_ctx* ctx, const t2l_packed_cells* in, float* out_emb, void* stream);

/* ---- database + search (a6) ------------------------------------------------------------------ */
/* Replaces: the host array cell_encodings (training/coarse.py:81,105-113). Copies n_rows x 256 fp32
 * rows (dev) into library-owned HBM. row_offset = global id of local row 0 (0 on a single GPU; the
 * shard's first row when the DB is row-sharded over ranks). Synchronises the stream. */
int t2l_db_set(t2l_ctx* ctx, const float* emb, int64_t n_rows, int64_t row_offset, void* stream);
int64_t t2l_db_rows(const t2l_ctx* ctx);

/* Replaces: the per-query loop `scores = cell_encodings @ text_encodings[q]; argsort(-scores)[:K]`
 * (training/coarse.py:119-125). queries: dev f32[n_queries,256].
 * out_idx:   dev i32[n_queries,K]  global row ids, best first; -1 where K > n_rows
 * out_score: dev f64[n_queries,K]  float64 dot products of the f32 values (what the reference ranks by);
 *                                  -inf where idx == -1. May be NULL.
 * Result contract: identical to a float64 scan + stable descending sort (exact ties: lower row id
 * first). Internally: f32 MFMA candidate scan -> float64 re-rank -> per-query certificate; queries
 * whose certificate fails are re-done by an exact float64 scan on the device (no host round trip).
 * One stream at a time: a context's scratch (candidate lists, counters, the small-batch path's published lists and ticket
 * counters) is per context, not per stream — calls on one context may be issued on different streams only when the caller
 * orders them (events / synchronisation); concurrent searches need one context each, or t2l_search_lanes. */
int t2l_search(t2l_ctx* ctx, const float* queries, int32_t n_queries, int32_t k, int32_t* out_idx,
               double* out_score, void* stream);
/* n_batches independent searches of n_queries queries each, issued back to back from C: queries dev f32[n_batches, n_queries, 256],
 * out_idx dev i32[n_batches, n_queries, k], out_score dev f64[...] or NULL. Exactly t2l_search called n_batches times (the
 * reference's loop answers query after query, training/coarse.py:119-125) without the caller's per-call binding cost — a Python
 * ctypes call costs ~14 us, twice the device time of a single-query search. */
int t2l_search_many(t2l_ctx* ctx, const float* queries, int32_t n_batches, int32_t n_queries, int32_t k, int32_t* out_idx,
                    double* out_score, void* stream);
/* Pipelined searches. With t2l_set_option("search_lanes", n), 2 <= n <= 4, consecutive t2l_search calls — independent jobs
 * in the reference's loop too (training/coarse.py:119-125 runs query after query) — run their scan -> re-rank chains on n
 * internal streams, round-robin, each with its own scratch: the next call's scan overlaps the previous call's re-rank and no
 * kernel-boundary bubble separates calls (measured: 42.9 -> 40.3 us per 4,096-query call). t2l_search then only ENQUEUES;
 * the outputs of every call issued so far are ordered into `stream` by t2l_search_join (queries and output buffers must
 * stay untouched until then). Batches under 256 queries, multi-segment shards and databases in heavy mode join and run in
 * the caller's stream as before. Default n = 1: t2l_search is stream-ordered and t2l_search_join is a no-op.
 * t2l_search_ordered is t2l_search WITHOUT the lanes whatever "search_lanes" says: pending lane work is joined, then scan and
 * re-rank run on `stream` itself — for callers that consume the result right away (sharded exchange, eval_epoch), where a fork
 * onto a lane followed by an immediate join would cost two cross-stream hops per call for nothing. */
int t2l_search_join(t2l_ctx* ctx, void* stream);
int t2l_search_ordered(t2l_ctx* ctx, const float* queries, int32_t n_queries, int32_t k, int32_t* out_idx,
                       double* out_score, void* stream);

/* The ONE exchange step of the row-sharded database (new design; the reference is single-device):
 * every rank searches its shard (global ids via row_offset), the per-rank [n_queries,k] results are
 * all-gathered (RCCL over xGMI) into idx/score dev [parts][n_queries][k], and this merges them to the
 * global top-k by (score desc, row id asc). parts * k <= 256. out_score may be NULL. No order is assumed inside a part and
 * invalid entries (id -1) may sit anywhere; row ids are meant to be unique across parts (disjoint row shards) — a (score, id)
 * pair present in two parts comes out twice. */
int t2l_merge_topk(t2l_ctx* ctx, const int32_t* idx, const double* score, int32_t parts, int32_t n_queries,
                   int32_t k, int32_t* out_idx, double* out_score, void* stream);

/* Same exchange with ONE collective: t2l_pack_pairs turns a rank's (idx, score) [n_queries,k] into f64 [n_queries,k,2]
 * records {score, (double)row id}; after the all-gather, t2l_merge_pairs merges dev f64 [parts][n_queries][k][2]. */
int t2l_pack_pairs(t2l_ctx* ctx, const int32_t* idx, const double* score, int32_t n_queries, int32_t k, double* pairs,
                   void* stream);
int t2l_merge_pairs(t2l_ctx* ctx, const double* pairs, int32_t parts, int32_t n_queries, int32_t k, int32_t* out_idx,
                    double* out_score, void* stream);
/* Same exchange with ONE collective and NO pack launch: a rank hands t2l_search output pointers into ONE block —
 * ids i32[n_queries][k] at offset 0, scores f64[n_queries][k] at score_offset (8-byte aligned, >= 4 * n_queries * k) — the
 * blocks of all ranks are all-gathered back to back (block_bytes each, 12 bytes per candidate on the wire) and merged here.
 * (text2loc_amd.sharded.ShardedSearcher: search -> all_gather -> this; the per-rank work of an 8-GPU step is three launches.) */
int t2l_merge_gathered(t2l_ctx* ctx, const void* blocks, int64_t block_bytes, int64_t score_offset, int32_t parts,
                       int32_t n_queries, int32_t k, int32_t* out_idx, double* out_score, void* stream);

/* Number of queries of the LAST t2l_search that took the exact-scan fallback (synchronises). */
int t2l_search_fallbacks(t2l_ctx* ctx, int32_t* out_count);
/* Number of queries of the LAST t2l_search whose first certificate failed and whose kept candidates were all re-scored in
 * float64 (second stage; the exact-scan count above is a subset of these). Synchronises. */
int t2l_search_rescored(t2l_ctx* ctx, int32_t* out_count);
/* All counters of the last t2l_search call (synchronises): out8[0] queries that ended in a float64 VALU scan of the shard,
 * [1] queries re-scored beyond the first L candidates (in the re-rank wave or by the fallback kernel), [2] queries the
 * re-rank handed to the fallback kernel, [3] auto-mode probe count, [4] queries deferred to the float64 MFMA stage (heavy
 * mode), [5] = [0] + [4] of the previous call, [6] queries the MFMA stage could not certify, [7] queries it served. */
int t2l_search_counters(t2l_ctx* ctx, int32_t* out8);

/* ---- contrastive loss (a8) ------------------------------------------------------------------- */
/* Replaces: ContrastiveLoss.forward (training/losses.py:269-283) and its autograd backward.
 * anchor/positive: dev f32[batch,256] (need not be normalised: the loss re-normalises, :271-272).
 * loss: dev f32[1]; grad_anchor/grad_positive: dev f32[batch,256] = d loss/d input, or NULL. batch <= 128 is ONE launch;
 * 128 < batch <= T2L_MAX_LOSS_BATCH (the all-gathered global batch of data-parallel training, SURVEY.md 8e: W ranks x B rows,
 * text2loc_amd.losses.ContrastiveLoss(group=...)) runs the same arithmetic as six launches over a [batch][batch] matrix in HBM. */
int t2l_contrastive_loss(t2l_ctx* ctx, const float* anchor, const float* positive, int32_t batch,
                         float temperature, float* loss, float* grad_anchor, float* grad_positive,
                         void* stream);

/* ---- text head after the frozen T5 encoder (f-4a), eval mode ------------------------------------ */
/* Replaces: LanguageEncoder.forward from `description_encodings = out.last_hidden_state` up to and including `self.inter_mlp`
 * (models/language_encoder.py:127-135): one nn.TransformerEncoderLayer(d_model 1024, 4 heads, dim_feedforward 4096, post-norm,
 * ReLU, no padding mask) over the token positions of every sentence, max over the tokens, Linear(1024 -> D) + BatchNorm1d (eval).
 * T5 itself stays on PyTorch-ROCm; what follows inter_mlp (the D-wide inter-sentence layer of the coarse model,
 * language_encoder.py:137-147; nothing for the fine model) is t2l_text_inter below. Weights: host fp32 blobs under `prefix` (NULL = "language_encoder."):
 * intra_module.0.{self_attn.in_proj_weight [3072,1024], self_attn.in_proj_bias, self_attn.out_proj.{weight,bias},
 * linear1.{weight [4096,1024],bias}, linear2.{weight [1024,4096],bias}, norm1.*, norm2.*}, inter_mlp.0.{0.weight [D,1024], 0.bias,
 * 1.weight, 1.bias, 1.running_mean, 1.running_var}, D <= 256; intra_module_num_layers must be 1 (the reference's default,
 * training/args.py:71). Synchronous; the library keeps packed copies.
 * t2l_text_head: hidden = dev f32[n_sentences, n_tokens, 1024] (T5's last_hidden_state, sentence-major as the reference's
 * tokenizer call produces it), 1 <= n_tokens <= 32; out = dev f32[n_sentences, D]. Arithmetic: split-f16 MFMA products with f32
 * accumulation (~5e-7 relative; option "encoder_f16" = 1: one f16 product per operand pair, ~1e-4). *overflow (dev int32, may be
 * NULL) is set to 1 when a value entering a product left the f16 range (|v| >= 3e4 or non-finite): `out` is then not to be
 * trusted and the caller runs that batch on its f32 path. Option "text_head_rows" (default 16,384): token rows per pass. */
int t2l_text_head_load_weights(t2l_ctx* ctx, const t2l_weight_desc* w, int32_t n, const char* prefix);
int t2l_text_head(t2l_ctx* ctx, const float* hidden, int32_t n_sentences, int32_t n_tokens, float* out, int32_t* overflow,
                  void* stream);
/* t2l_text_inter — the other half of the head, eval mode. Replaces: LanguageEncoder.forward from
 * `description_encodings.view(batch_size, num_sentence, -1)` to its return value (models/language_encoder.py:137-147):
 * x = sent.view(n_descriptions, n_sentences_per, 256); x += TransformerEncoderLayer(d_model 256, 4 heads, dim_feedforward 1024,
 * post-norm, ReLU)(x) over the sentences of every description (the residual AROUND the layer is the reference's `+=`); max over
 * the sentences. sent = dev f32[n_descriptions * n_sentences_per, 256], description-major (exactly what t2l_text_head returns for
 * the reference's sentence order); out = dev f32[n_descriptions, 256] (NOT normalised: CellRetrievalNetwork.encode_text does that,
 * models/cell_retrieval.py:57-63). 1 <= n_sentences_per <= 32. Needs `inter_module.0.*` (in_proj [768,256], out_proj [256,256],
 * linear1 [1024,256], linear2 [256,1024], norm1, norm2) among the tensors handed to t2l_text_head_load_weights and
 * inter_module_num_layers = 1 (training/args.py:72's default); T2L_ESTATE otherwise (the fine model has no such layer). Same
 * arithmetic, overflow flag and kernels (with d_model 256) as t2l_text_head. */
int t2l_text_inter(t2l_ctx* ctx, const float* sent, int32_t n_descriptions, int32_t n_sentences_per, float* out, int32_t* overflow,
                   void* stream);

/* ---- fine stage (f-1): CrossMatch downstream of the text branch, eval mode ---------------------- */
/* Replaces CrossMatch.load_state_dict for everything except language_encoder.* (evaluation/pipeline.py:258-262): tensors named
 * as in the reference's fine checkpoint — object_encoder.* at fine_embed_dim = 128 (models/cross_matcher.py:57),
 * cross_objects.{i}.* / cross_hints.{i}.* (nn.TransformerDecoderLayer: self_attn, multihead_attn, linear1/2, norm1-3),
 * mlp_offsets.{0,2}.*. cfg: class_embed / color_embed / use_* as for the coarse model; num_layers =
 * args.fine_num_decoder_layers (1..4), num_heads = args.fine_num_decoder_heads (4). Synchronous. */
int t2l_fine_load_weights(t2l_ctx* ctx, const t2l_weight_desc* w, int32_t n, const t2l_model_config* cfg);

/* Replaces the 3D-submap branch of CrossMatch.forward (models/cross_matcher.py:97-104): ObjectEncoder.forward on the
 * padded cells + reshape + F.normalize. `in`: every cell holds EXACTLY pad_size = 16 objects (cut / padded on the host as
 * Kitti360TopKDataset.load_pose_and_cell does, dataloading/kitti360pose/eval.py:147-156), n_objects = 16 * n_cells.
 * out_desc: dev f32[n_cells,16,128]. The descriptors depend on the cell only: compute them once per database cell. */
int t2l_fine_encode_objects(t2l_ctx* ctx, const t2l_packed_cells* in, float* out_desc, void* stream);

/* Replaces the CCAT module + offset head of CrossMatch.forward (models/cross_matcher.py:106-131) for n_pairs (query, cell)
 * pairs in ONE launch (the reference: one Python-level forward per query, evaluation/pipeline.py:113-116):
 *   for i: obj = cross_objects[i](obj, hints); hints = cross_hints[i](hints, obj);  offsets = mlp_offsets(max over hints)
 * cell_desc: dev f32[*,16,128] (t2l_fine_encode_objects); hint_desc: dev f32[*,n_hints,128] = LanguageEncoder(is_fine)
 * output of the PyTorch text branch; cell_index / hint_index: dev i32[n_pairs] rows of those tables for each pair, or NULL
 * for the identity (pair p uses row p). out_offsets: dev f32[n_pairs,2] = the pose estimate inside the cell, in [0,1]^2. */
int t2l_fine_match(t2l_ctx* ctx, const float* cell_desc, const int32_t* cell_index, const float* hint_desc,
                   const int32_t* hint_index, int32_t n_pairs, int32_t n_hints, float* out_offsets, void* stream);

/* ---- training step of the object branch (a9) --------------------------------------------------- */
/* Replaces, for the object branch, the body of train_epoch (training/coarse.py:31-58): model.train() forward of
 * CellRetrievalNetwork.encode_objects, loss.backward() through it, optimizer.zero_grad() / optim.Adam.step().
 * The text branch (T5 + head) stays on PyTorch autograd; the two meet at the [batch,256] embeddings.
 *
 * One LIVE tensor of the model: the library reads parameters from `data` and accumulates (+=) gradients into `grad` on
 * every backward — exactly torch's .data / .grad of the reference's nn.Parameters, bound by pointer, never copied.
 * BatchNorm buffers ("...1.running_mean", "...1.running_var") are bound with grad = NULL and updated in place by the
 * training-mode forward (momentum 0.1, unbiased batch variance); num_batches_tracked is an int64 the caller bumps. */
typedef struct {
  const char* name; /* state_dict key */
  float* data;      /* dev f32[numel] */
  float* grad;      /* dev f32[numel], or NULL for buffers */
  int64_t numel;
} t2l_train_tensor;

/* Binds the tensors the configuration uses (same names t2l_load_weights takes; a missing one is T2L_EINVAL) and
 * resets the optimizer state (Adam moments = 0, step = 0). Pointers must stay valid until the next bind. Synchronous. */
int t2l_train_bind(t2l_ctx* ctx, const t2l_train_tensor* tensors, int32_t n, const t2l_model_config* cfg);

/* encode_objects under model.train(): BatchNorm1d uses the statistics of THIS batch of objects (all objects of all
 * cells, also those beyond slot 28) and updates the running buffers; the four dropout sites of every
 * nn.TransformerEncoderLayer (attention probabilities, dropout1, feed-forward dropout, dropout2) drop with probability
 * dropout_p (torch default 0.1) using counter-based masks: element i of site j is kept iff
 * lowbias32(i*0x9E3779B1 + (seed ^ j*0x85EBCA77)) >> 8 >= dropout_p*2^24 (torch's own RNG stream is not reproducible
 * outside torch; the mask function is part of this ABI so that a checker can replay it). Activations are kept inside
 * the context until the next forward. `in` device arrays must stay valid until t2l_encode_cells_backward returns.
 * out_emb: dev f32[n_cells,256]. */
int t2l_encode_cells_train(t2l_ctx* ctx, const t2l_packed_cells* in, float dropout_p, uint32_t seed, float* out_emb,
                           void* stream);

/* loss.backward() through the forward above: grad_emb = d loss / d out_emb, dev f32[n_cells,256]. Parameter gradients
 * are ADDED to the bound grad buffers (float atomics: summation order, hence the last bits, varies between runs).
 * grad_pn_feat: dev f32[n_objects,256] receiving d loss / d pn_feat (class_embed == 0), or NULL. */
int t2l_encode_cells_backward(t2l_ctx* ctx, const float* grad_emb, float* grad_pn_feat, void* stream);

/* The PointNet++ object backbone under model.train() — trained jointly with the rest in the published configuration
 * (README.md:87-99, no --pointnet_freeze; models/pointcloud/pointnet2.py:18-100 via models/object_encoder.py:86-99). Needs
 * every object_encoder.pointnet.{sa1,sa2,sa3}.point_conv.local_nn.* / ga.mlp.* / lin1.* / lin2.* tensor in the t2l_train_bind
 * call: all WITH gradient buffers (trained jointly), or all WITHOUT (--pointnet_freeze, object_encoder.py:53-55: no backward,
 * but the forward still runs in training mode). The reference calls the backbone once per cell, so every BatchNorm1d uses the
 * statistics of that cell's rows and updates its running statistics once per cell, in cell order; this call does the whole
 * batch at once with exactly that segmentation. pos, rgb: dev f32[n_objects,256,3]; cell_offsets: HOST i32[n_cells+1];
 * out_features2: dev f32[n_objects,256] (feed it to t2l_encode_cells_train as pn_feat). Activations stay in the context
 * (only real edges are rows; ~9 GB at 64 cells) until the next call. PARITY UNPINNED like t2l_pointnet_features. */
int t2l_pointnet_features_train(t2l_ctx* ctx, const float* pos, const float* rgb, const int32_t* cell_offsets, int32_t n_cells,
                                float* out_features2, void* stream);
/* Backward of the call above: grad_features2 = d loss / d features2 (what t2l_encode_cells_backward wrote to grad_pn_feat),
 * dev f32[n_objects,256]. Parameter gradients are ADDED to the bound buffers. */
int t2l_pointnet_backward(t2l_ctx* ctx, const float* grad_features2, void* stream);

/* optimizer.zero_grad() and torch.optim.Adam(lr, betas, eps).step() (no weight decay, no amsgrad — what
 * training/coarse.py:258 constructs) over every bound tensor that has a gradient buffer. torch.optim.Adam skips parameters
 * whose .grad is None and counts steps per parameter; here the bound tensors form two groups with a step counter each — the
 * object branch (steps on every call) and the PointNet++ backbone, which steps only when t2l_pointnet_backward ran since the
 * last t2l_zero_grad (a batch fed precomputed features2 leaves the backbone's weights AND moments untouched). */
int t2l_zero_grad(t2l_ctx* ctx, void* stream);

/* ---- the text head in TRAINING mode (f-4 / the text half of a9) ---------------------------------- */
/* Replaces: LanguageEncoder.forward downstream of T5's last_hidden_state under model.train() + its autograd backward
 * (models/language_encoder.py:127-147 as run by training/coarse.py:44,55-56; the published command trains this head —
 * --fixed_embedding freezes T5 only, README.md:87-99).
 * t2l_text_train_bind: live device pointers (data, grad) of `<prefix>intra_module.0.*` (d_model 1024, 4 heads, ff 4096),
 * `<prefix>inter_mlp.0.{0.weight [256,1024], 0.bias, 1.weight, 1.bias, 1.running_mean, 1.running_var}` (the two running buffers
 * without grad) and `<prefix>inter_module.0.*` (d_model 256, 4 heads, ff 1024); prefix NULL = "language_encoder.". No copies: the
 * kernels read the parameters and accumulate (+=) into the gradient buffers the caller's optimizer owns (torch.optim.Adam steps
 * them). Pointers must stay valid until the next bind.
 * t2l_text_head_train: hidden = dev f32[n_sentences, n_tokens, 1024], sentence-major and description-major (sentence j of
 * description i is row i * S + j, S = n_sentences / n_descriptions — the reference's own order); out = dev f32[n_descriptions, 256],
 * not normalised. Training-mode semantics: the four dropout sites of both TransformerEncoderLayers drop with probability
 * dropout_p by the counter-based masks of t2l_encode_cells_train (sites 0-3: the token layer, 4-7: the inter-sentence layer);
 * BatchNorm1d of inter_mlp normalises with the statistics of THIS batch of sentences and updates the running buffers (momentum
 * 0.1, unbiased running variance). Activations are kept inside the context until the next forward.
 * t2l_text_head_backward: grad_out = dev f32[n_descriptions, 256] (dLoss / d out); parameter gradients are accumulated into the
 * bound buffers; `hidden` receives no gradient (T5 is frozen; a caller that trains T5 keeps the PyTorch path).
 * Arithmetic: option "text_train_bf16" (default 2): GEMM operands as split-bf16 (hi + lo bf16, three bf16 MFMAs per 16-step: products
 * within 2^-16 + 2^-18 relative, f32 accumulation, f32's exponent range — gradients need no loss scaling); 1: plain bf16 operands
 * (BASELINE config 4's arithmetic). (An f32-MFMA operand form, 0, existed until round 5: it measured slower than PyTorch's f32 step
 * and is rejected now.) Everything else (softmax, LayerNorm, BatchNorm, pooling, dropout) is f32. */
int t2l_text_train_bind(t2l_ctx* ctx, const t2l_train_tensor* tensors, int32_t n, const char* prefix);
int t2l_text_head_train(t2l_ctx* ctx, const float* hidden, int32_t n_sentences, int32_t n_tokens, int32_t n_descriptions, float dropout_p,
                        uint32_t seed, float* out, void* stream);
int t2l_text_head_backward(t2l_ctx* ctx, const float* grad_out, void* stream);

/* The head's optimizer (training/coarse.py:42,56 with optim.Adam(model.parameters()), :258: the published command trains the head's
 * 13.6 M parameters). t2l_text_adam_step: torch.optim.Adam's arithmetic (defaults: no weight decay, no amsgrad) over every tensor that
 * t2l_text_train_bind received WITH a gradient buffer, in bind order, in ONE launch; exp_avg / exp_avg_sq live inside the library, the
 * step count starts at 0 with a bind (a re-bind of an unchanged list under option "train_keep_adam_state" keeps moments and step).
 * t2l_text_zero_grad zeroes the bound gradient buffers in place (one launch). t2l_text_adam_state: as t2l_adam_state (m, v:
 * dev f32[numel], concatenated in bind order; m == v == NULL queries numel and, set == 0, step). */
int t2l_text_adam_step(t2l_ctx* ctx, float lr, float beta1, float beta2, float eps, void* stream);
int t2l_text_zero_grad(t2l_ctx* ctx, void* stream);
int t2l_text_adam_state(t2l_ctx* ctx, int32_t set, float* m, float* v, int64_t* step, int64_t* numel, void* stream);

/* Data-parallel training with the reference's batch statistics. The reference trains ONE process on the whole batch
 * (training/coarse.py:31-58): every BatchNorm1d of the object branch (object_encoder.py:41-52, 121-149) and of inter_mlp
 * (language_encoder.py:99) normalises over all objects / sentences of the batch. With the batch split over ranks, the per-channel
 * sums (forward: sum y, sum y^2, rows; backward: sum dy, sum dy * xhat, rows) must be added up over the ranks between the statistics
 * and the apply launch of every BatchNorm — torch.nn.SyncBatchNorm's exchange. The library has no collectives of its own
 * (RCCL stays with the host: sharded.py / torch.distributed), so the caller supplies both the memory and the sum:
 *   buf   dev f64[t2l_train_sync_bn_doubles()] — replaces the library's own accumulator slots from the next forward on;
 *   fn    called from inside t2l_encode_cells_train / _backward / t2l_text_head_train / _backward, on the calling thread, once per
 *         BatchNorm stage (3 or 4 per direction of an object-branch step, 1 per direction of a text-head step): must replace buf[0..n) at `range` by its sum over
 *         all ranks, ordered on `stream` after what is already enqueued there and before what follows (torch.distributed.all_reduce
 *         on the current stream does exactly that). Returns 0, or non-zero to fail the enclosing call with T2L_ESTATE.
 * fn == NULL returns to per-rank statistics (and the library's own slots). Running statistics are updated with the global values on
 * every rank; the parameter gradients of the BatchNorm layers stay per rank (the gradient all_reduce adds them up). PointNet++'s
 * BatchNorms are per cell (pointnet_train.h) and never cross ranks. */
typedef int (*t2l_allreduce_fn)(void* user, double* range, int64_t n, void* stream);
int64_t t2l_train_sync_bn_doubles(void);
int t2l_train_sync_bn(t2l_ctx* ctx, double* buf, int64_t n_doubles, t2l_allreduce_fn fn, void* user);

int t2l_adam_step(t2l_ctx* ctx, float lr, float beta1, float beta2, float eps, void* stream);

/* optimizer.state_dict() / load_state_dict() for the engine-stepped tensors (torch.optim.Adam keeps exp_avg / exp_avg_sq /
 * step per parameter; here they live inside the library). numel (out, may be NULL) = total elements over the stepped
 * tensors, concatenated in the order of t2l_train_bind's parameter walk (feature branches, mlp_merge, obj_inter_module.*).
 * m == v == NULL: query numel and (set == 0) step only. Otherwise m, v: dev f32[numel]; set == 0 copies the moments out and
 * writes *step, set != 0 copies them in and takes *step (*step = object-branch step | backbone step << 32). A re-bind with an unchanged parameter list (same names and sizes,
 * new pointers: model.to(), re-assigned .grad) keeps moments and step IF option "train_keep_adam_state" is 1 at the time of
 * the re-bind (default 0: every bind starts from zero moments — a different model may use the same names). */
int t2l_adam_state(t2l_ctx* ctx, int32_t set, float* m, float* v, int64_t* step, int64_t* numel, void* stream);

/* ---- knobs (tests / bench) ------------------------------------------------------------------- */
/* "certify_eps_scale" (default 1.0): multiplies the f32 error bound of the search certificate; a huge
 *     value forces every query through the exact fallback (used by the parity tests).
 * "search_mode"       (default 0): 0 = f16 MFMA scan (operands scaled by exact powers of two), 2 = split-bf16 (three bf16 MFMAs per
 *     product) scan. Both feed the same float64 re-rank + certificate, so the RESULTS are identical; only the speed differs.
 *     (1 = an exact-f32 MFMA scan until round 5: 212 us per 4,096 queries against 30; removed.)
 * "search_auto"       (default 1): mode 0 only — when more than 1 in 8 queries of a batch fail the f16 certificate (scores
 *     packed tighter than its error band) later searches use the split-bf16 scan until fewer than 1 in 16 would.
 * "train_bf16"        (default 0): 1 = the GEMMs of t2l_encode_cells_train / t2l_encode_cells_backward round their operands to
 *     bf16 (RNE) and run on the bf16 MFMA with f32 accumulation; parameters, activations, BatchNorm / LayerNorm / softmax,
 *     the loss and Adam stay f32 (what torch.autocast(bfloat16) keeps in f32 too). BASELINE config 4's "bf16".
 *     2 = split-bf16: every operand as hi + lo bf16, three MFMAs per 16-step (relative product error <= 2^-16 + 2^-18, f32's
 *     exponent range): the f32 goldens are met to 1e-4 at close to the bf16 variant's speed. Applies to the PointNet++
 *     training calls too.
 * "train_gemm_block"  (default 0 = by measurement; 32, 64): output block of the training step's tile GEMMs. 64 = every wave holds
 *     2 x 2 accumulator tiles (half the L2 -> L1 operand traffic, a quarter of the workgroups): 4 % faster with bf16 / split-bf16
 *     operands, 3 % slower in f32 (the deeper operand ring of the 32 x 32 form does not fit beside 64 accumulators) — so 0 picks 64
 *     with train_bf16 != 0 and 32 otherwise. Results are identical up to float32 summation order.
 * "search_xcd_qgroups" (default 4; 1, 2, 4, 8): paired scan — the workgroups of one XCD form a rectangle of (query blocks) x
 *     (splits): with g groups an XCD's L2 pulls 1/g of the query batch in the prologue and g/8 of the f16 plane over the main
 *     loop (1 = all queries, 1/8 of the plane: round 2's mapping). Applies when the grid divides evenly, else 1.
 * "search_heavy"      (set by the engine, see search_auto): 1 = queries no certificate settles go to the float64 MFMA exact
 *     stage instead of the fallback kernel's float64 VALU scan. With search_auto = 0 the caller may force it.
 * "train_keep_adam_state" (default 0): see t2l_adam_state.
 * "search_small"      (default 1): batches of <= 16 queries against a shard of < stream_min_rows (and <= 65,536) rows, search_mode 0:
 *     the whole search as ONE launch, exact float64 from the start (search_small.hip) — every workgroup scores its rows with the
 *     re-rank's arithmetic, ranks them and publishes its best K write-through; the last workgroup to arrive merges the lists. Ids and
 *     float64 scores are bit-identical to the batched path's. 10.8 us per single-query call issued from C at N = 11,259 against
 *     21.2 us on the batched two-launch path (0: that path, kept for the A/B line of bench.py and for Q > 16).
 * "search_small_wgs"  (default 0 = by query count, 128 or 192): its workgroups (published lists) per slice of 4 queries, <= 256.
 * "stream_min_rows"   (default 65536): batches of <= 64 queries against a shard of at least this many rows use the
 *     HBM-streaming scan (every CU streams a disjoint DB slice once) instead of the batched scan.
 * "search_nsplit"     (default 0 = auto): DB row splits per query block in the scan kernel.
 * "pointnet_pyg_self_loops" (default 1): see t2l_pointnet_features.
 * "encoder_f32"       (default 0): 1 = run t2l_encode_cells, t2l_pointnet_features and t2l_fine_match entirely on the f32
 *     MFMA. By default their big contractions use split-f16 MFMAs (hi*hi + hi*lo + lo*hi, ~5e-7 relative) behind range
 *     safeguards — bounds derived from the weights at load time (encoder, fine stage), a row-norm guard on the raw
 *     descriptors (fine stage), a magnitude watch (PointNet++) — and whatever fails them is computed by the f32 kernels.
 * "encoder_f16"       (default 0): 1 = ONE f16 product per operand pair in those kernels instead of the three of the split
 *     form (the high halves of the same packing): embeddings / offsets / features within ~1e-4 of the reference's instead of
 *     2e-7 (the published target is 1e-3), 20-40 % less time. The range safeguards stay in force.
 * "search_lanes"      (default 1): 2..4 = pipelined searches, see t2l_search_join.
 * "search_wide_repair" (default 512, at most 1024): rows a re-rank wave may re-score in float64 to settle a query whose certificate failed
 *                     (every kept key that reaches the threshold + every row of a list whose floor does) before the query is
 *                     handed to an exact scan of the whole shard; 0 = off (tests of the exact stages).
 * "encoder_two_cells" (default 1): t2l_encode_cells with two cells per eight-wave workgroup, activations as split-f16 planes in LDS and the
 *                     merge / out_proj / feed-forward weight fragments shared by both cells (split-f16 and plain-f16 arithmetic, two or
 *                     more feature slots). 0 = one cell per four-wave workgroup on f32 tiles — the kernel that serves models with ONE
 *                     feature slot (and, in its f32 form, "encoder_f32"); the option exists so that the tests can hold that kernel
 *                     to the reference goldens, which are four-feature models. Same results to rounding.
 * "search_merge_lists" (default 2): the paired scan's candidate hand-off to the re-rank. 0 = four 24-byte lists per (query, workgroup);
 *                     1 = ONE 32-byte record (the best 7 of their 24 keys, the source list in two more code bits, + a bound on every other
 *                     key): a third of the bytes written back at the end of the launch, -0.7 us per step at Q = 4096; 2 = records while the
 *                     report cards show fewer than 1 query in 64 failing its first certificate (a repair behind a record re-scores 4x the
 *                     rows of a plain list's), plain lists otherwise. Results are bit-identical in every setting.
 * "search_tile_sel"   (default 1): with merged records, the paired scan selects through a local top 3 per group of 8 scores (the best two
 *                     enter the lane's list, the third travels as the record's decodable bound B1; the re-rank re-scores that group's 8
 *                     rows when a query's certificate fails on B1 alone). 0 = every score inserted into the list (round 5's form; A/B).
 *                     Results are identical either way.
 * "search_pair_ll"    (default 6): per-lane list length of the paired scan (5: experiment, halves the certificate's margin).
 * "profile_events"    (default 0): n >= 1 records hipEvents around every n-th launch of each kernel (t2l_kernel_stats);
 *                     two records cost ~6 us of queue time per bracketed kernel, which matters beside a 30 us kernel.
 * "profile_rerank"    (default 1): 0 = sampled search launches bracket the scan only.
 * "stats_reset"       (any value): forget every kernel-time sample so far (host-only: no stream operation, no sync). */
int t2l_set_option(t2l_ctx* ctx, const char* name, double value);

/* Per-kernel device time measured with hipEvent pairs recorded on the caller's stream around each launch
 * (no synchronisation while recording; enabled by option "profile_events" >= 1, off by default).
 * Returns the average duration (ms) and the number of launches recorded since the previous call for
 * name = "search_scan" | "search_rerank" | "encode_cells" | "contrastive_loss" | "reduce_objects" | "train_forward" |
 * "train_backward" | "adam_step" | "pointnet" | "fine_objects" | "fine_match" | "search_fallback" | "search_exact" | "search_small" (at most
 * the last 512), "pointnet_train_index" | "pointnet_train_forward" | "pointnet_train_backward";
 * "search_scan_span" and "search_scan_busy" need no option: every workgroup of the paired scan stamps its own start and end
 * (100 MHz clock, the last 64 launches) — span = first workgroup start -> last workgroup end, busy = mean workgroup
 * duration (the GPU time a launch used: what still means something when pipelined launches overlap);
 * then clears the record. Synchronises on the recorded events. */
int t2l_kernel_stats(t2l_ctx* ctx, const char* name, float* out_avg_ms, int32_t* out_count);

#ifdef __cplusplus
}
#endif
#endif /* T2L_H_ */
