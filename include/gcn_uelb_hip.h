/*
 * C ABI of libgcn_uelb_hip.so: hand-written gfx950 (MI355X) kernels for the GCN hot path of
 * Shota-OSD/graph-convnet-uelb.  The reference is pure Python/PyTorch and has no FFI of its own;
 * each entry point below names the reference function(s) it replaces (paths relative to the
 * reference checkout).  The Python classes that keep the reference's nn.Module / factory API call
 * these through ctypes (graph-convnet-uelb_amd/src/gcn/hip/lib.py).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the comment says "host"; tensors are contiguous,
 *    fp32 ("float") or int64 ("long long"), laid out exactly as the reference's tensors:
 *      node tensors [B][V][C][H], edge tensors [B][V_i][V_j][C][H], H fastest.
 *  - the library allocates nothing: the caller owns every buffer, including `saved` (activations
 *    kept for backward) and `ws` (scratch); the *_floats / *_bytes queries give the sizes.
 *  - work is enqueued on `stream` (a hipStream_t passed as void*); no call synchronises.
 *  - return value 0 = ok, negative = error; gcn_last_error() returns the message for the calling
 *    thread.  One host thread per device; calls that share buffers must be stream-ordered.
 *  - H must be a multiple of 4 and <= GCN_MAX_H.
 *  - `rng` points at two 64-bit words {seed, step} in device memory, read by the dropout code at
 *    execution time (so a captured HIP graph sees the step it is replayed with).  May be NULL when
 *    training == 0.
 */
#ifndef GCN_UELB_HIP_H
#define GCN_UELB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCN_ABI_VERSION 3
#define GCN_MAX_H 320
#define GCN_LAYER_NPARAMS 12
#define GCN_MAX_MLP_LAYERS 4

int gcn_abi_version(void);
const char* gcn_last_error(void);

/* Measurement aid, no counterpart in the reference: after gcn_prof_begin(substr) every launch whose
 * kernel name contains substr is timed with HIP events on the stream it is launched on: the main layer
 * kernels by their own begin / end time stamps (hipExtLaunchKernelGGL start / stop events), the others by
 * an event bracket from which an empty bracket recorded beside it is subtracted.  gcn_prof_end() waits for
 * the events and returns the summed time and the launch count.  Host-side only, not for use while a
 * stream is being captured into a graph. */
int gcn_prof_begin(const char* kernel_substr);
int gcn_prof_end(double* total_ms, int* launches);

/* ------------------------------------------------------------------------------------------------
 * One residual gated graph-conv layer.
 * Replaces ResidualGatedGCNLayer.forward (src/gcn/models/gcn_layers.py:108-129), i.e.
 * EdgeFeatures.forward (:81-94), NodeFeatures.forward (:55-70), BatchNormEdge/BatchNormNode
 * (:5-43), ReLU + residual + Dropout (:124-128), and what autograd derives from them.
 *
 * params[12], in the reference's state_dict order:
 *   node_feat.U.weight, node_feat.U.bias, node_feat.V.weight, node_feat.V.bias,
 *   edge_feat.U.weight, edge_feat.U.bias, edge_feat.V.weight, edge_feat.V.bias,
 *   bn_node.batch_norm.weight, bn_node.batch_norm.bias, bn_edge.batch_norm.weight, bn_edge.batch_norm.bias
 * aggregation: 0 = "sum", 1 = "mean".  stream_id: distinguishes dropout streams (use the layer index).
 * ---------------------------------------------------------------------------------------------- */
/* Launch plan the library picks for a problem size (introspection for tests and tools; nothing in the reference
 * corresponds): out[0] = row-chunk length of the edge backward pass, out[1] = chunks per graph, out[2] = 16-row
 * j tiles per (graph, row, commodity), out[3] / out[4] = workgroups of the forward / backward edge kernel,
 * out[5] = 1 if the weight-streaming kernels (hidden_dim > 64) run, out[6] = 1 if the edge kernels with two column
 * blocks per wavefront and the weight-gradient product fused into the backward kernel run (hidden_dim of 2 or 4
 * column blocks, about 300,000 edge rows or more), out[7] = 1 if a ragged last j tile (at most 8 of its 16 columns in
 * use, hidden_dim > 64) runs in a launch of its own with several rows packed into a tile; out[4] then counts the
 * workgroups of the launch for the other j tiles (0 if there is none). */
int gcn_layer_plan(int B, int V, int C, int H, int out[8]);
size_t gcn_layer_saved_floats(int B, int V, int C, int H);
size_t gcn_layer_ws_bytes(int B, int V, int C, int H);

int gcn_layer_fwd(int B, int V, int C, int H, int aggregation, float p_drop, int training,
                  const unsigned long long* rng, int stream_id,
                  const float* x_in, const float* e_in, const float* const* params /* host array of 12 device ptrs */,
                  float* x_out, float* e_out, float* saved, void* ws, size_t ws_bytes, void* stream);

/* dx / de hold d loss / d x_out, d e_out on entry and d loss / d x_in, d e_in on return.
 * grads[12] (host array of device pointers, same order as params) are overwritten, not accumulated. */
int gcn_layer_bwd(int B, int V, int C, int H, int aggregation, float p_drop, int training,
                  const unsigned long long* rng, int stream_id,
                  const float* x_in, const float* e_in, const float* const* params, const float* saved,
                  float* dx, float* de, float* const* grads, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole model: embedding -> L layers -> edge MLP -> (masking) -> loss and edge-error counters.
 * Replaces ResidualGatedGCNModel.forward (src/gcn/models/gcn_model.py:45-123), MLP.forward
 * (src/gcn/models/gcn_layers.py:167-176), loss_edges (src/gcn/models/model_utils.py:51-78) and
 * edge_error (:255-308), plus their autograd backward (src/gcn/training/supervised_strategy.py:94-95).
 *
 * params / grads: host arrays of gcn_model_num_params() device pointers in the reference's
 * state_dict order: nodes_embedding.weight, commodities_embedding.weight,
 * edges_values_embedding.weight, 12 per layer (as above), then mlp_edges.layers.{0,3,..}.weight/.bias
 * pairs, mlp_edges.output_layer.weight, mlp_edges.output_layer.bias.
 * ---------------------------------------------------------------------------------------------- */
typedef struct gcn_model_desc {
    int B, V, C, H, L;
    int mlp_layers;          /* linear layers in the edge MLP, 1..GCN_MAX_MLP_LAYERS */
    int voc_out;             /* 1 (edge scores) or 2 (two-class logits) */
    int n_emb;               /* rows of nodes_embedding (3*C in the reference) */
    int aggregation;         /* 0 sum, 1 mean */
    int training;            /* dropout active */
    int mask_invalid_edges;  /* logits of zero-capacity edges := -1e9 */
    int has_labels;          /* y_edges / edge_cw given: compute loss and edge-error counters (voc_out == 2) */
    float p_layer;           /* dropout of the conv layers (reference ctor default 0.3) */
    float p_mlp;             /* dropout of the MLP (0.2) */
    int advance_rng;         /* gcn_model_fwd adds 1 to the step word of `rng` on the device before anything reads it
                                (a training loop replayed from a HIP graph then needs no launch of its own for that);
                                gcn_model_bwd never changes it */
} gcn_model_desc;

int gcn_model_num_params(const gcn_model_desc* d);
size_t gcn_model_saved_floats(const gcn_model_desc* d);
size_t gcn_model_ws_bytes(const gcn_model_desc* d);

/* Test aids (nothing in the reference corresponds).  gcn_model_saved_layout: float offsets inside `saved` of
 * what gcn_model_fwd keeps of layer `layer`: out[0] pre-BatchNorm edge activations e_tmp [B,V,V,C,H], out[1]
 * pre-BatchNorm node activations x_tmp [B,V,C,H], out[2] BatchNorm statistics [4][H] (edge mean, edge 1/std,
 * node mean, node 1/std), out[3] the layer's edge input, out[4] its node input; out[5..7] = 0.
 * gcn_bn_relu_pattern: mask[row][h] = 1 iff the ReLU behind a BatchNorm is open at that element, decided with
 * the expression the library's kernels use ((tmp - mean) * rstd * gamma + beta > 0 in float32), so that a
 * reference can be given the library's own decision for BatchNorm outputs within rounding of zero. */
int gcn_model_saved_layout(const gcn_model_desc* d, int layer, size_t out[8]);
int gcn_bn_relu_pattern(size_t rows, int H, const float* tmp, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, unsigned char* mask, void* stream);

/* y_pred: [B,V,V,C,voc_out].  loss_out[4] = {loss, sum of class weights, edge error %, 0}.
 * err_counts: int32 [2][B*C] (correct, mask) scratch owned by the caller (H <= 64 with mlp_layers <= 2 counts
 * in the workspace instead and leaves it untouched); err_flags: uint8 [B*C], 1 where the (b, c) slice is not
 * perfectly predicted.  Both may be NULL when has_labels == 0. */
int gcn_model_fwd(const gcn_model_desc* d, const float* cap, const float* demand, const long long* x_nodes,
                  const long long* y_edges, const float* edge_cw, const float* const* params,
                  const unsigned long long* rng, float* y_pred, float* loss_out, int* err_counts,
                  unsigned char* err_flags, float* saved, void* ws, size_t ws_bytes, void* stream);

/* dloss: device scalar, upstream gradient of the loss (NULL = 1).  dy_pred: upstream gradient of
 * y_pred, [B,V,V,C,voc_out] (NULL = 0).  grads are overwritten. */
int gcn_model_bwd(const gcn_model_desc* d, const float* cap, const float* demand, const long long* x_nodes,
                  const long long* y_edges, const float* edge_cw, const float* const* params,
                  const unsigned long long* rng, const float* saved, const float* loss_out,
                  const float* dloss, const float* dy_pred, float* const* grads,
                  void* ws, size_t ws_bytes, void* stream);

/* gcn_model_bwd that also tells the caller when parts of the gradient are final, so that a data-parallel trainer can
 * start averaging them over the ranks while the rest of the backward pass still runs (the reference has no such hook:
 * its nn.DataParallel reduces all gradients after backward(), src/gcn/train/trainer.py:54; torch's DistributedDataParallel
 * buckets are the model).  done_events: NULL, or L + 1 event handles (hipEvent_t; entries may be NULL): entry k < L is
 * recorded on `stream` once every gradient of conv layer k is final, entry L once those of the edge MLP are.  Layers
 * finish from the top down; layer 0 and the embeddings are final when the call's last kernel is (entry 0 is recorded
 * there).  Up to 64 channels the layers' gradients are finished by shared launches at the end: all entries are
 * recorded after the last kernel. */
int gcn_model_bwd_ev(const gcn_model_desc* d, const float* cap, const float* demand, const long long* x_nodes,
                     const long long* y_edges, const float* edge_cw, const float* const* params,
                     const unsigned long long* rng, const float* saved, const float* loss_out,
                     const float* dloss, const float* dy_pred, float* const* grads,
                     void* ws, size_t ws_bytes, void* stream, void* const* done_events);

/* ------------------------------------------------------------------------------------------------
 * Stand-alone loss and edge-error metric on existing two-class logits [B,V,V,C,2].
 * Replaces loss_edges (src/gcn/models/model_utils.py:51-78) and edge_error / _edge_error (:255-308)
 * for callers that already hold predictions; inside gcn_model_fwd both are fused into the head.
 * loss_out[4] = {loss, sum of class weights, edge error %, 0}; err_flags uint8 [B*C].
 * edge_cw == NULL in gcn_loss_edges_fwd computes the edge error only (loss_out[0] is then NaN).
 * ---------------------------------------------------------------------------------------------- */
size_t gcn_loss_ws_bytes(int B, int V, int C);
int gcn_loss_edges_fwd(int B, int V, int C, const float* logits, const long long* y_edges, const float* edge_cw,
                       float* loss_out, unsigned char* err_flags, void* ws, size_t ws_bytes, void* stream);
int gcn_loss_edges_bwd(int B, int V, int C, const float* logits, const long long* y_edges, const float* edge_cw,
                       const float* loss_out, const float* dloss, float* dlogits, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Capacity-aware beam search, batched on the GPU.
 * Replaces BeamSearchAlgorithm.search / _search_single_batch / _beam_search_for_commodity of the
 * four algorithms in src/gcn/algorithms/beamsearch_uelb.py (:37-55, :89-177, :187-274, :289-377,
 * :387-457) and the score preparation of BeamSearchAlgorithm.__init__ (:14-21).
 *
 * algorithm: 0 standard, 1 deterministic, 2 greedy (beam size forced to 1), 3 unconstrained.
 * y [B,V,V,C] fp32 log-probabilities, cap [B,V,V] fp32, comm [B,C,3] int64 (src, dst, demand).
 * perms [n_perms,C] int32: the permutations torch.randperm(C) would return, in draw order;
 * perm_start[b] int32: index of graph b's first permutation (graph b uses perm_start[b] + attempt).
 * run_mask uint8 [B] (NULL = all): graphs with 0 are skipped and their outputs left untouched.
 * Outputs: paths int32 [B,C,PL] padded with -1, PL = max(V, 10); path_len int32 [B,C];
 * feasible int32 [B]; attempts int32 [B] (permutations consumed by standard / greedy).
 * ws: gcn_beam_ws_bytes() bytes: working capacities for strict mode when V*V*4 > 32 KiB, and the results of a
 *     first pass that searches every (graph, commodity) against the initial capacities in parallel; the ordered
 *     pass keeps such a result whenever the capacity updates so far leave that commodity's edge predicate
 *     (cap != 0 and demand <= cap) unchanged, and searches again otherwise: same paths as the sequential
 *     reference order, bit for bit.  With a smaller or null ws the first pass is skipped.  The workspace also holds
 *     the graphs' adjacency lists (edges with non-zero capacity, up to 24 per node on average) and the scores
 *     gathered per list entry, built by two launches of their own: a step then examines the beams' out-edges
 *     instead of K x V entries, in the same order and with the same tie-break; graphs with more edges, more than
 *     2048 nodes, K > 64, or a smaller workspace keep the dense scan.  On graphs whose capacity matrix exceeds
 *     32 KiB, in strict mode with 16 commodities or more and the full workspace, the ordered pass runs in rounds of
 *     two launches (walk until a commodity's predicate differs from the capacities the first pass searched against,
 *     then search all remaining commodities against the capacities reached); the state between the launches
 *     (position, attempt, working capacities per list entry) lives in the workspace too.  Same paths, bit for bit.
 * ---------------------------------------------------------------------------------------------- */
size_t gcn_beam_ws_bytes(int B, int V, int C, int K);
int gcn_beam_search(int B, int V, int C, int K, int algorithm, int mode_strict, int max_iter, const float* y,
                    const float* cap, const long long* comm, const int* perms, int n_perms, const int* perm_start,
                    const unsigned char* run_mask, int* paths, int* path_len, int* feasible, int* attempts, void* ws,
                    size_t ws_bytes, void* stream);

/* five_d != 0: in = logits [B,V,V,C,2], y = log_softmax(in, dim 4)[..., 1];
 * five_d == 0: in = scores [B,V,V,C],   y = log_softmax(in, dim 2) (over the next node j).
 * Exact zeros become -1e-20 (beamsearch_uelb.py:21). */
int gcn_beam_scores(int B, int V, int C, int five_d, const float* in, float* y, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Maximum load factor per graph of decoded paths (the step right after the beam search).
 * Replaces mean_feasible_load_factor (src/common/graph/graph_utils.py:90-192) up to its final mean over
 * the batch, and the per-commodity "path reaches its destination" test of
 * Evaluator._compute_completion_and_approx (src/gcn/train/evaluator.py:45-52).
 * paths / path_len as produced by gcn_beam_search (PL columns).  usage_ws: B*V*V int32 scratch.
 * max_lf float [B] (+inf if a zero-capacity edge is used); complete uint8 [B*C].
 * ---------------------------------------------------------------------------------------------- */
int gcn_load_factor(int B, int V, int C, int PL, const int* paths, const int* path_len, const float* cap,
                    const long long* comm, int* usage_ws, float* max_lf, unsigned char* complete, void* stream);

/* The same computation for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The B*V*V int32 usage
 * scratch that gcn_load_factor takes from the caller lives in a fixed array inside the library's code object
 * (GCN_STATIC_USAGE_INTS int32 = 4 MiB, held by every GPU the library runs on until the process ends); the call
 * makes no hipMalloc / hipFree and no other allocating call.  B*V*V > GCN_STATIC_USAGE_INTS is refused (-2).
 * Every call on one GPU shares that GPU's array (it is cleared at the start of each call, on `stream`): calls
 * that overlap on one GPU (different streams or host threads) race; stream-ordered calls are fine.  The first
 * call on a device looks up the array's device address (hipGetSymbolAddress); make it outside stream capture. */
#define GCN_STATIC_USAGE_INTS (1u << 20)
int gcn_load_factor_static(int B, int V, int C, int PL, const int* paths, const int* path_len, const float* cap,
                           const long long* comm, float* max_lf, unsigned char* complete, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stochastic path sampling of the REINFORCE training path, batched on the GPU.
 * Replaces PathSampler.sample / _sample_single_path / _top_p_sample (src/gcn/algorithms/path_sampler.py
 * :51-214, :272-405, :407-426), the reachability precomputation (:220-260) and, in gcn_path_sample_bwd, what
 * autograd derives from the returned log-probabilities (src/gcn/training/reinforcement_strategy.py:427-428).
 *
 * y [B,V,V,C] fp32 edge scores (voc_out == 1 predictions), cap [B,V,V], comm [B,C,3] int64 (src, dst, demand).
 * temperature > 0, top_p = nucleus mass, eps = weight of the uniform mixture (0 = none).  V <= 256.
 * rng: two 64-bit words {seed, call counter} in device memory; the draws are a counter-based function of
 * (seed, counter, graph, commodity, step), so a call is reproducible and the caller advances the counter.
 * given_paths == 0: paths are drawn and written; given_paths != 0: paths / path_len are inputs and only their
 * log-probabilities and entropies are evaluated (rng may be NULL).
 * Outputs: paths int32 [B,C,PL] padded with -1, PL = max(V, 10); path_len int32 [B,C]; logp float [B,C] (the
 * reference's per-graph value is the sum over C); step_entropy float [B,C,PL] with n_entropy int32 [B,C] valid
 * entries per commodity (fallback steps record none).
 * gcn_path_sample_bwd ADDS d(sum_c logp[b,c]) / d y * dlogp[b] into dy [B,V,V,C]; the caller clears dy or passes
 * a gradient to accumulate into.  Paths never visit a node twice, so no two additions touch the same element.
 * gcn_reachability: reach uint8 [B,V,V], reach[b,i,j] = 1 iff j can be reached from i over edges with cap > 0.
 * ---------------------------------------------------------------------------------------------- */
int gcn_path_sample(int B, int V, int C, float temperature, float top_p, float eps, const float* y, const float* cap,
                    const long long* comm, const unsigned long long* rng, int given_paths, int* paths, int* path_len,
                    float* logp, float* step_entropy, int* n_entropy, void* stream);
int gcn_path_sample_bwd(int B, int V, int C, float temperature, float eps, const float* y, const float* cap,
                        const long long* comm, const int* paths, const int* path_len, const float* dlogp, float* dy,
                        void* stream);
int gcn_reachability(int B, int V, const float* cap, unsigned char* reach, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Entropy bonus of the REINFORCE loss: mean entropy of the next-node distributions softmax(y / T over j) of
 * the (graph, node, commodity) rows that have at least two entries above 1e-8 (all rows if there is none).
 * Replaces the entropy_grid computation of ReinforcementLearningStrategy.compute_loss
 * (src/gcn/training/reinforcement_strategy.py:460-476) and its autograd backward.
 * y [B,V,V,C]; row_ws: 2*B*V*C floats kept from forward to backward; out float[4] = {entropy, rows averaged,
 * 1 if only multi-choice rows were averaged, 0}.  gcn_entropy_grid_bwd ADDS dout[0] * d entropy / d y into dy
 * (dout NULL = 1).
 * ---------------------------------------------------------------------------------------------- */
int gcn_entropy_grid_fwd(int B, int V, int C, float temperature, const float* y, float* row_ws, float* out, void* stream);
int gcn_entropy_grid_bwd(int B, int V, int C, float temperature, const float* y, const float* row_ws, const float* out,
                         const float* dout, float* dy, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Expansion of a packed mini-batch into the model's dense input tensors on the GPU, so that only the compact
 * form is copied from the host.  Replaces the per-sample array construction of DatasetReader.process_batch
 * (src/common/data_management/dataset_reader.py:117-145: node features from the commodities, edge labels from
 * the node visiting order) and the per-array conversions and uploads of BaseTrainingStrategy.prepare_batch_data
 * (src/gcn/training/base_strategy.py:89-102).
 * Inputs (device): edges_u8 [B,V,V] adjacency, comm_i32 [B,C,3] (src, dst, demand), flow_i16 [B,C,V] visiting
 * order of each node on the commodity's path (0 = not visited).  Outputs: x_edges int64 [B,V,V], x_nodes int64
 * [B,V,C], y_edges int64 [B,V,V,C], comm int64 [B,C,3], demand float [B,C].  succ_ws: B*C*V int16 scratch.
 * ---------------------------------------------------------------------------------------------- */
int gcn_unpack_batch(int B, int V, int C, const unsigned char* edges_u8, const int* comm_i32, const short* flow_i16,
                     long long* x_edges, long long* x_nodes, long long* y_edges, long long* comm, float* demand,
                     short* succ_ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Adam update of n parameter tensors in one or a few launches (multi-tensor).  Replaces the
 * torch.optim.Adam(lr) step of the trainer (src/gcn/train/trainer.py:95,
 * src/gcn/training/supervised_strategy.py:98-100) with the same arithmetic (no amsgrad; weight_decay is the
 * L2 form added to the gradient; bias corrections from `step`, the 1-based count of this update):
 *   g += wd * p ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
 *   p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)
 * params / grads / exp_avg / exp_avg_sq: host arrays of n device pointers (fp32), numel: host array of n sizes.
 * ---------------------------------------------------------------------------------------------- */
int gcn_adam_step(int n, float* const* params, const float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const long long* numel, float lr, float beta1, float beta2, float eps,
                  float weight_decay, long long step, void* stream);

#ifdef __cplusplus
}
#endif
#endif
