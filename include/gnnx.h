/* gnnx.h — C ABI of the MI355X-native GNNExplainer mask-optimisation engine (libgnnx_hip.so).
 *
 * The reference (RexYing/gnn-model-explainer) has no FFI: its boundary for this path is the
 * Python class API of explainer/explain.py.  These entry points are what a ctypes binding
 * behind that API calls; each cites the reference code it replaces.
 * Paths are relative to the reference root.
 *
 * Conventions: plain C, no torch types.  Every device buffer is owned by the caller (PyTorch);
 * the library allocates only its small per-plan tables at gnnx_plan_create (and one per-iteration Adam
 * scalar table the first time a given gnnx_hyper is run), never per-target or per-iteration buffers.  gnnx_run is asynchronous on the given hipStream_t.  Return 0 = ok, non-zero =
 * error with text in gnnx_last_error().  One plan per device, not thread-safe.
 *
 * Packed layout ("segmented over targets"):
 *   target t has n_t sub-graph nodes, ld_t = round_up(n_t, 32);
 *   square arrays (A, M, Abar)  : ld_t x ld_t floats at float offset offQ[t], row-major, zero padded;
 *   row arrays (X, feature-like): ld_t rows x 32 floats at row offset offR[t], zero padded;
 *   yhat (predicted class id of every sub-graph node, as float): one float per row.
 */
#ifndef GNNX_H
#define GNNX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNX_FEAT_STRIDE 32 /* floats per row of row arrays; D, H, O <= 32 */
#define GNNX_MAX_CLASSES 32
#define GNNX_LOSS_TERMS 16  /* per (target, iteration): [0..4] pred, size, lap, ent, feat_size (explain.py:808-819); [5] the mask density the reference prints
                             * (ExplainModule.mask_density, explain.py:680-683: sum of the masked adjacency / sum of the adjacency, AFTER the epoch's step,
                             * :142-148); [8..15] the class probabilities of the epoch's forward (explain.py:710-714; the first 8 classes) */

typedef struct gnnx_plan_s* gnnx_handle;

/* Which sub-graphs are optimised in one batched job.  Replaces the per-target Python setup of
 * Explainer.explain (explainer/explain.py:80-117) for a whole list of targets
 * (explain_nodes :234-236, explain_nodes_gnn_stats :296-299, explain_graphs :362-363). */
typedef struct {
    int32_t num_targets;
    const int32_t* n;          /* host [T] sub-graph sizes (extract_neighborhood, explain.py:492-501) */
    const int32_t* target_row; /* host [T] node_idx_new (explain.py:496); ignored in graph mode */
    const int32_t* gt_label;   /* host [T] label used by the prediction loss (explain.py:750-753) */
    int32_t D, H, O, C;        /* input dim, hidden dim, embedding dim, classes (models.py:83-132) */
    int32_t graph_mode;        /* 0: GcnEncoderNode head (models.py:363-376); 1: GcnEncoderGraph max-pool head (models.py:269-316) */
    int32_t mask_relu;         /* 0: mask_act = "sigmoid"; 1: mask_act = "ReLU" (explain.py:669-670, 757-760) - dense streaming
                                * kernels only; like the reference it yields NaN masks whenever an entry of M0 lies outside (0, 1] */
    int32_t bn;                /* 1: --bn (apply_bn, models.py:222-228, 241-253): every node's hidden activation standardised over its
                                * features after the ReLU of the two hidden layers; dense streaming kernels only */
} gnnx_problem;

/* Frozen encoder parameters, HOST pointers in the reference state_dict layouts:
 * conv_first/conv_block.0/conv_last .weight [d_in, d_out] and .bias [d_out] (models.py:31-52),
 * pred_model.weight [C, H+H+O], pred_model.bias [C] (models.py:198). */
typedef struct {
    const float* W[3];
    const float* b[3];
    const float* Wp;
    const float* bp;
} gnnx_model;

/* Optimiser + regulariser constants: Adam of utils/train_utils.py:9-10 (torch defaults) and
 * ExplainModule.coeffs (explain.py:624-631). */
typedef struct {
    double lr, beta1, beta2, eps; /* DOUBLES, like the Python floats torch.optim.Adam holds: its bias corrections 1 - beta^k and the
                                   * lerp / addcmul weights 1 - beta are evaluated in double and only then rounded to fp32
                                   * ((float)(1 - 0.999) != 1.0f - 0.999f by 1.3e-5) */
    float c_size, c_feat_size, c_ent, c_lap;
    int32_t num_iters;   /* args.num_epochs (explain.py:137) */
    int32_t record_loss; /* 1: fill loss[T][num_iters][GNNX_LOSS_TERMS] (explain.py:808-819 scalars) */
    int32_t use_graph;   /* 1: capture the launch sequence once into a hipGraph and replay it */
    int32_t use_resident; /* 1: targets may take the on-chip-resident kernels gnnx_plan_analyze routed them to (gnnx_get_route);
                           * 0: every target runs on the dense streaming kernels.  With record_loss the resident path is taken only when
                           * EVERY target is routed to the sparse on-chip-resident kernel and the encoder has the reference's widths
                           * (its logging form + k_dead_entries for the entries off the edges); other plans log on the streaming kernels. */
    /* The other optimisers / LR schedulers the reference's build_optimizer can return (utils/train_utils.py:7-22):
     *   opt          0 = Adam (lr, beta1, beta2, eps above), 1 = SGD(momentum), 2 = RMSprop(alpha, eps), 3 = Adagrad(eps) - torch
     *                defaults otherwise (no weight decay, dampening 0, not centered, lr_decay 0);
     *   lr_schedule  HOST pointer to num_iters doubles = the learning rate of every iteration of this call, as the scheduler
     *                (StepLR / CosineAnnealingLR, stepped after every optimiser step: explain.py:144-146) leaves it in
     *                optimizer.param_groups[0]["lr"]; NULL = constant lr. */
    int32_t opt;
    int32_t edge_results_only; /* 1: the caller reads the result as edge lists (gnnx_gather_values / gnnx_gather_edges): on the edge-sparse
                                * routes Abar is then DEFINED ONLY ON THE EDGES of the sub-graphs - the kernels skip the ld^2 zero-fill of
                                * every target's block (28 GB per launch on the 16 384-target BA-House x100k set); 0: dense Abar blocks */
    double momentum, alpha;
    const double* lr_schedule;
} gnnx_hyper;

int gnnx_plan_create(const gnnx_problem* prob, const gnnx_model* model, gnnx_handle* out);
int gnnx_destroy(gnnx_handle h);

/* Layout the caller packs into. ld/offQ/offR are host arrays of num_targets entries. */
int64_t gnnx_total_q(gnnx_handle h);    /* floats in each square array */
int64_t gnnx_total_rows(gnnx_handle h); /* rows in each row array */
int gnnx_get_layout(gnnx_handle h, int32_t* ld, int64_t* offQ, int64_t* offR);
size_t gnnx_workspace_bytes(gnnx_handle h);

/* The hot loop: num_iters x {masked adjacency, 3-layer GCN forward, loss, analytic backward through the
 * mask, Adam on edge mask + feature mask} for every target of the plan (explain.py:137-146 with
 * ExplainModule.forward :685-715, .loss :740-820, loss.backward :142, optimizer.step :144).
 *   A, X, yhat : device inputs in the packed layout (read-only)
 *   M          : device in/out; in = initial edge mask M0 (construct_edge_mask, explain.py:645-663),
 *                out = mask after num_iters Adam steps (ExplainModule.mask)
 *   Abar       : device out; masked adjacency of the LAST forward, i.e. after num_iters-1 steps
 *                (ExplainModule.masked_adj consumed at explain.py:209-211)
 *   feat_mask  : device out [T][32]; feature-mask parameter after num_iters steps (ExplainModule.feat_mask)
 *   loss       : device out [T][num_iters][GNNX_LOSS_TERMS] or NULL
 *   workspace  : device scratch of gnnx_workspace_bytes(h) bytes
 *   stream     : hipStream_t */
int gnnx_run(gnnx_handle h, const gnnx_hyper* hyper, const float* A, const float* X, const float* yhat,
             float* M, float* Abar, float* feat_mask, float* loss, void* workspace, size_t workspace_bytes,
             void* stream);

/* Checkpoint / resume of the optimiser (and the hook of the windowed parity tests): the state torch.optim.Adam keeps for the
 * two parameters of ExplainModule (explain.py:622; utils/train_utils.py:9-10) - `exp_avg`, `exp_avg_sq` and the step count.
 * gnnx_run_resume(h, hyper, r, ...) == gnnx_run(...) started from that state: iteration k of the call applies Adam step
 * first_iter + k + 1 (bias corrections 1 - beta^(first_iter + k + 1)), M (in/out) holds the mask after first_iter steps.
 *   first_iter : steps already taken (0 with all-NULL inputs == gnnx_run)
 *   m, v       : DEVICE, packed square layout like M: exp_avg / exp_avg_sq of the mask, or NULL = zeros
 *   feat       : DEVICE [T][3][32]: feat_mask, its exp_avg, its exp_avg_sq (columns >= D ignored), or NULL = zeros
 *   m_out, v_out, feat_out : optional DEVICE outputs in the same layouts = the state after the call.  On the edge-sparse routes
 *                (gnnx_get_route 4..8) only the entries on edges of the sub-graph are written (the only live ones: the
 *                others never reach an output, explain.py:665-678), so zero-fill m_out / v_out first if the rest is read.
 * m_out / v_out must not alias m / v unless they are the same pointer (in-place continuation is allowed). */
typedef struct {
    int32_t first_iter;
    const float* m;
    const float* v;
    const float* feat;
    float* m_out;
    float* v_out;
    float* feat_out;
} gnnx_resume;
int gnnx_run_resume(gnnx_handle h, const gnnx_hyper* hyper, const gnnx_resume* resume, const float* A, const float* X,
                    const float* yhat, float* M, float* Abar, float* feat_mask, float* loss, void* workspace,
                    size_t workspace_bytes, void* stream);

/* method="att" - the attention GraphConv of the reference (/root/reference/models.py:36-37 the att_weight parameters, :62-68
 * `x_att = x W_att; att = x_att x_att^T; adj = adj * att` in every layer).  att_weights: HOST pointer, [3][32][32] floats, zero
 * padded, W_att of layer l at l * 1024 + b * 32 + a (b = input column, as the reference stores it).  Once set, every gnnx_run /
 * gnnx_run_resume of the plan optimises all targets in k_att (gnnx_att.hpp: one workgroup per target, edge-list state, all
 * iterations in one launch, forward and backward through the attention products).  Node and graph mode (GcnEncoderNode /
 * GcnEncoderGraph heads), sigmoid mask, no --bn, no loss logging (other combinations return an error: the Python mirror sends
 * them to its PyTorch-ROCm route); gnnx_forward, gnnx_grad_baseline and gnnx_time_kernel implement the base encoder and return
 * an error on such a plan.  The call synchronises
 * with the host (the edge arrays are sized from a device count). */
int gnnx_set_att_weights(gnnx_handle h, const float* att_weights);

/* Inspect the packed adjacency A (DEVICE pointer, the layout of gnnx_get_layout) and choose the kernel of every target.  All sparse
 * routes optimise only the mask entries on EDGES of the sub-graph - the only ones that reach an output of the reference
 * (explain.py:665-678, 209-211; non-edge entries of M then keep their initial values):
 *   - edge state fits one compute unit (n <= 512, <= 2048 undirected edges, rows of <= 256 entries, LDS budget; node and graph mode):
 *     the sparse on-chip-resident kernel, in the smallest size class that holds the target - 64 threads (n <= 32), 256 (n <= 128),
 *     512 (node mode: <= 256 row slots within two hops of the target), 1024; node-mode batches of 512-thread and single-tile targets
 *     run as ONE mixed launch;
 *   - larger node-mode targets (n <= 16383, <= 8192 rows and < 65535 directed entries within two hops of the target):
 *     k_sparse_large - the entries of the rows within two hops in LDS, row arrays in the workspace, far edges as closed recursions;
 *   - single-tile node-mode targets that fit no sparse class: the dense on-chip-resident kernel;  everything else streams.
 * Optional: without this call the plan uses the dense resident kernels for batches of <= 3 row blocks per target and streams the rest.
 * Synchronises `stream`.  Environment: GNNX_SPARSE_RESIDENT=0 disables the sparse kernels, GNNX_SPARSE_LARGE=0 / GNNX_TINY_SPARSE=0 /
 * GNNX_SPARSE_512=0 / GNNX_SPARSE_MIXED=0 single classes, GNNX_KEEP_256=0|1 the class merge of saturated batches, GNNX_DEBUG_ROUTE=1 prints
 * the analysis of every target that ends up streaming. */
int gnnx_plan_analyze(gnnx_handle h, const float* A, void* stream);

/* gnnx_plan_analyze that also looks at the packed features X (DEVICE pointer, [R][32] rows as gnnx_run takes them; null = do not look).
 * Node mode, the reference's encoder widths (D = 10, H = O = 20): when every target the sparse on-chip-resident kernel takes has
 * CONSTANT feature rows (X[j] == X[0] bit for bit inside the target - ConstFeatureGen, gengraph.py:60-61, i.e. every synthetic dataset
 * of the reference; featureless graphs in general), its launches use the constant-feature form: Abar . X (models.py:70) and the X part of
 * dL/dAbar need no gathers.  Same products in the same order as the general form: results are bit-identical.  The form checks the X
 * handed to gnnx_run / gnnx_run_resume again; a target whose rows are not constant there comes back NaN (it fails loudly).
 * GNNX_XCONST=0 keeps the general form. */
int gnnx_plan_analyze_features(gnnx_handle h, const float* A, const float* X, void* stream);

/* The kernel every target is routed to (host array of num_targets entries): 0 = dense streaming kernels, 1..3 = dense
 * on-chip-resident kernel of that many 32-row blocks, 4 / 5 / 6 = sparse on-chip-resident kernel in its 1024- / 256- / 64-thread size
 * class (n <= 512 / 128 / 32), 7 = k_sparse_large (node mode, n <= 16383), 8 = sparse on-chip-resident kernel, 512-thread class
 * (node mode, n <= 512, at most 256 row slots within two hops of the target). */
int gnnx_get_route(gnnx_handle h, int32_t* route);

/* Decision trace (parity instrumentation of the boundary; tests/test_decision_parity.py): device buffers the runs that follow fill, or
 * null / null to switch it off.  The reference's trajectory is piecewise smooth: between two iterations at which a ReLU gate
 * (models.py:241, 251) or a max-pool (models.py:283-300) changes sides it is a continuous function of its state, and an implementation
 * that takes the SAME side of every gate stays within round-off of it.  The trace records the side taken:
 *   gates     [num_iters][R][2] uint32, R = rows of the row arrays (gnnx_get_layout: target t owns rows offR[t] .. offR[t] + n_t):
 *             bit c of word (iter, row, l) = U_{l+1}[row][c] > 0 in the forward of that iteration; rows outside the row set of the
 *             layer (node mode: beyond two hops of the target for l = 0, beyond its neighbours for l = 1 - their activations reach
 *             no output) read 0;
 *   pool_rows [T][num_iters][96] int32 (graph mode, else null): the row the max-pool of pooled column (layer, c) picked, -1 where the
 *             column does not exist.
 * Recorded by the sparse on-chip-resident kernel in its logging form: every target of the plan routed there, D = 10 (node) / 14
 * (graph), H = O = 20, use_resident = 1 - a run on any other plan returns an error while a trace is set. */
int gnnx_set_trace(gnnx_handle h, uint32_t* gates, int32_t* pool_rows);

/* Measurement hook: device time (ms, HIP events on the side streams the kernels run on) of the on-chip-resident
 * launches of the LAST gnnx_run, in situ (i.e. while the other kernels of that run were executing):
 * ms[0..2] = dense resident kernels of 1..3 row blocks, ms[3..5] = sparse resident kernel, 1024- / 256- / 64-thread
 * size class, ms[6] = sparse kernel for larger targets, ms[7] = sparse resident kernel, 512-thread class (8 floats);
 * 0 where nothing was launched.
 * Waits for those launches to finish. */
int gnnx_resident_times(gnnx_handle h, float* ms);

/* Device-side packing of the plan's sub-graphs from the full graph in CSR form (all pointers are DEVICE
 * pointers): replaces the host's dense slicing `adj[nb][:, nb]`, `feat[nb]`, `argmax(pred[nb])` of
 * Explainer.extract_neighborhood / explain (explain.py:492-501, 94-106) for the whole batch.
 *   indptr [N+1] int64, indices [nnz] int32, weights [nnz] or NULL (all ones): symmetric adjacency
 *   feat [N][feat_stride], pred_label [N] (predicted class id as float) or NULL (graph mode)
 *   nb: concatenated ascending neighbour lists, nb_off [T+1] offsets (nb_off[t+1]-nb_off[t] == n[t] of the plan)
 *   A, X, yhat: outputs in the packed layout; zero-filled by this call before packing */
int gnnx_pack_csr(gnnx_handle h, const int64_t* indptr, const int32_t* indices, const float* weights, const float* feat,
                  int32_t feat_stride, const float* pred_label, const int32_t* nb, const int64_t* nb_off, float* A,
                  float* X, float* yhat, void* stream);
/* gnnx_pack_csr followed by gnnx_plan_analyze_features(h, A, X) as ONE call: the packing kernel counts every row's entries while it
 * places them, so the analysis needs no pass of its own over the dense blocks (two launches fewer per batch). */
int gnnx_pack_csr_analyze(gnnx_handle h, const int64_t* indptr, const int32_t* indices, const float* weights, const float* feat,
                          int32_t feat_stride, const float* pred_label, const int32_t* nb, const int64_t* nb_off, float* A, float* X,
                          float* yhat, void* stream);

/* One forward only (no update): fills Abar from M and returns softmax probabilities of the head,
 * probs device out [T][GNNX_MAX_CLASSES] (ExplainModule.forward, explain.py:685-715). */
int gnnx_forward(gnnx_handle h, const float* A, const float* X, const float* M, const float* feat_mask_in,
                 float* Abar, float* probs, void* workspace, size_t workspace_bytes, void* stream);

/* Gradient baseline (`model="grad"`, explain.py:125-133, adj_feat_grad :717-738), node mode: one forward + backward of the
 * encoder on the UNMASKED sub-graphs (no edge mask, no feature mask) with loss -log softmax(logits[target])[label], where the
 * plan's gt_label holds the PREDICTED label of every target (explain.py:130), then
 *     out = sigmoid(|dL/dA| + |dL/dA|^T) * A      (out: device, packed square layout). */
int gnnx_grad_baseline(gnnx_handle h, const float* A, const float* X, float* out, void* workspace, size_t workspace_bytes,
                       void* stream);

/* Measurement hook for bench.py: relaunch one kernel class `reps` times on the current workspace
 * state between two hipEvents on `stream`; returns the average launch duration in milliseconds and
 * the algorithmic bytes / flops of one launch.  kind: 0 = fused mask/regulariser/Adam kernel, 1/2 = forward
 * contraction of layer 1/2, 3 = head kernel, 4 = backward contraction into layer 1; graph mode only:
 * 5 = forward contraction of layer 3, 6 = row-local backward of layer 3, 7 = backward contraction into layer 2;
 * 8 / 9 = one WHOLE launch (all num_iters iterations; overwrites M / Abar) of the sparse / single-tile dense
 * on-chip-resident kernel, with the algorithmic work of its targets (28 n^2 B and 6 n^2 (D + 2H) flop per iteration).
 * Kinds 0-7 walk the tile tables gnnx_run would walk (the streaming remainder of a hybrid batch). */
int gnnx_time_kernel(gnnx_handle h, const gnnx_hyper* hyper, int32_t kind, int32_t reps, const float* A,
                     const float* X, const float* yhat, float* M, float* Abar, void* workspace,
                     size_t workspace_bytes, void* stream, float* ms_avg, double* alg_bytes, double* alg_flops);

/* ---- index work either side of the loop (csrc/gnnx_graph.hpp) ------------------------------------------------------ */

/* k-hop walk sets of a batch of targets over the resident CSR graph (all pointers DEVICE pointers): the set
 * {u : (A + A^2 + ... + A^k)[v][u] > 0} of graph_utils.neighborhoods (utils/graph_utils.py:147-158; a dense O(N^3)
 * product on the whole graph in the reference) as the ascending id list Explainer.extract_neighborhood builds from it
 * (explain.py:492-501), plus node_idx_new = the target's position in its own list (explain.py:496).  Two passes:
 *   nb == NULL : sizes[t] = |set| for every target (the host needs them for gnnx_plan_create); when target_row is given the
 *                size pass reports it too (same value as the emit pass), so ONE copy tells the host sizes and rows and the
 *                emit pass may run with target_row == NULL and report nothing back;
 *   nb != NULL : nb[nb_off[t] .. nb_off[t] + |set|) = the list (nb_off = the prefix sums of the sizes for compact lists; a caller that
 *                cannot wait for the sizes may pass any offsets that leave room - e.g. t * num_nodes - and ask for sizes in the same
 *                pass: ONE launch, lists padded), target_row[t] = position of the target (-1: not in its own
 *                set, i.e. an isolated node - the reference then fails on an empty neighbourhood).
 * scratch: gnnx_khop_scratch_bytes(num_nodes, num_targets) bytes (0 when the bitmaps fit LDS: num_nodes <= 131072). */
size_t gnnx_khop_scratch_bytes(int32_t num_nodes, int32_t num_targets);
int gnnx_khop(const int64_t* indptr, const int32_t* indices, int32_t num_nodes, int32_t n_hops, const int32_t* targets,
              int32_t num_targets, int32_t* sizes, const int64_t* nb_off, int32_t* nb, int32_t* target_row, void* scratch,
              size_t scratch_bytes, void* stream);

/* Initial edge masks from the host's RNG stream: `raw` (DEVICE, gnnx_total_raw(h) floats) holds, target after target, the
 * n_t x n_t values of the ONE normal_ draw construct_edge_mask makes per target (explain.py:645-652), unpadded; this
 * spreads them over the padded blocks of M (padding = 0). */
int64_t gnnx_total_raw(gnnx_handle h);
int gnnx_scatter_masks(gnnx_handle h, const float* raw, float* M, void* stream);

/* The explanation as edge lists - the returned masks are exactly zero off the sub-graph's edges (explain.py:209-211).
 * gnnx_edge_counts: counts[t] (DEVICE int64 [T]) = upper-triangle (r < c) non-zeros of target t's block of A.
 * gnnx_gather_edges: for those edges in row-major order, at eoff[t] (DEVICE int64 [T+1], the prefix sums of counts):
 *   rc [E][2] = (r, c);  abar [E] = Abar[r][c] (the masked adjacency of the last forward) or NULL;
 *   m_rc [E][2] = M[r][c], M[c][r] (final mask parameters) or NULL.  Uses the workspace of gnnx_run as scratch. */
int gnnx_edge_counts(gnnx_handle h, const float* A, int64_t* counts, void* stream);
int gnnx_gather_edges(gnnx_handle h, const float* A, const float* Abar, const float* M, const int64_t* eoff, int32_t* rc,
                      float* abar, float* m_rc, void* workspace, size_t workspace_bytes, void* stream);
/* The edge structure of a batch never changes: gnnx_edge_positions records, in the order of gnnx_gather_edges, where every
 * edge lives - epos [E][2] (DEVICE int64) = float index of (r, c) and of (c, r) in the packed square arrays (rc filled too) -
 * and gnnx_gather_values then reads the values of any later run with one indexed load per edge. */
int gnnx_edge_positions(gnnx_handle h, const float* A, const int64_t* eoff, int32_t* rc, int64_t* epos, void* workspace,
                        size_t workspace_bytes, void* stream);
/* gnnx_plan_analyze* counts the edges while it looks at the adjacency anyway (one copy back for the routing figures and the counts):
 * gnnx_edge_counts_host hands those counts to the host (HOST int64 [T]; fails when no analysis has run), and gnnx_edge_layout is
 * gnnx_edge_positions from the row starts that analysis left on the device - one launch instead of three, no workspace.  Only valid
 * while A is the adjacency the analysis saw; gnnx_edge_counts / gnnx_edge_positions work on any A. */
int gnnx_edge_counts_host(gnnx_handle h, int64_t* counts);
int gnnx_edge_layout(gnnx_handle h, const float* A, const int64_t* eoff, int32_t* rc, int64_t* epos, void* stream);
int gnnx_gather_values(const int64_t* epos, int64_t num_edges, const float* Abar, const float* M, float* abar, float* m_rc, void* stream);

/* The seeded initial masks (explain.py:645-652: one normal_ draw of n x n values per target from a CPU generator seeded per target) for the
 * edge-sparse kernels, with the engine walked on the DEVICE: seeds [T] (DEVICE int64: the generator seed of every target), eoff / rc = the
 * edge layout of gnnx_edge_layout (DEVICE); scratch = a DEVICE array with room for every target's ld x ld block (the Abar array serves:
 * nothing reads it before the run) - it receives the raw mt19937 state words of every target's draws; words [E][4] (DEVICE uint32) = for
 * every upper-triangle edge (r, c) the two raw words of the Box-Muller pair that holds M[r][c], then those of M[c][r].  The host turns them
 * into the values ATen's normal_ gives them (gnnx_host_transform_edge_words, include/gnnx_host.h): bit-identical to
 * torch.manual_seed(seed); torch.FloatTensor(n, n).normal_(1.0, std) on the edges, with 16 bytes per directed entry crossing PCIe instead of
 * the host stepping through all n^2 draws.  Targets of fewer than 16 values (n <= 3) are left untouched (the host draws them whole). */
int gnnx_mt_edge_words(gnnx_handle h, const int64_t* seeds, const int64_t* eoff, const int32_t* rc, uint32_t* scratch, uint32_t* words,
                       void* stream);

/* Post-processing of a batch of explanations on the device, on the edge lists of gnnx_gather_edges (all pointers DEVICE):
 * gnnx_denoise_edges = io_utils.denoise_graph(masked_adj, node_idx, threshold_num=k, max_component=True)
 * (utils/io_utils.py:193-245, called at explain.py:306-308, 364-370) for every target: keep[e] = 1 for the edges of the largest
 * connected component of the graph made of the k heaviest undirected edges (ties at the threshold kept; first component in
 * node order among equals); threshold [T]; stats [T][3] = nodes, edges, smallest node id of that component. */
int gnnx_denoise_edges(gnnx_handle h, const int64_t* eoff, const int32_t* rc, const float* vals, int32_t threshold_num,
                       uint8_t* keep, float* threshold, int32_t* stats, void* workspace, size_t workspace_bytes, void* stream);
/* ROC-AUC of edge scores against 0/1 ground truth (explain.py:325-328: roc_auc_score over the concatenated pred / real of all
 * targets) as exact pair counts: counts (DEVICE, 4 x uint64) = positives, #{pos > neg}, #{pos == neg}, negatives;
 * AUC = (counts[1] + counts[2] / 2) / (counts[0] counts[3]).  pos_scratch: DEVICE, E floats. */
int gnnx_auc_counts(const float* vals, const uint8_t* real, int64_t num_edges, float* pos_scratch, unsigned long long* counts, void* stream);

/* Pipelined jobs (pipeline.BatchPipeline): the calling THREAD's plan-building uploads (gnnx_plan_create, gnnx_plan_analyze) go to
 * `stream` instead of the null stream, whose hardware queue may be busy with another batch's launch; NULL restores the default.
 * With a service stream the table blocks of a plan are copied from pinned staging and only ENQUEUED there (the call does not wait for
 * the copy); every later call on that plan that takes another stream orders it behind the upload with an event, so the caller needs no
 * synchronisation of its own.  With the default (NULL) uploads stay synchronous.  (The per-iteration optimiser scalar table of gnnx_run
 * is uploaded once per process and hyper-parameter set and shared by all plans.)  gnnx_lane_stream(i), i = 0..2: the library's process-wide launch lanes (hipStream_t) - the
 * streams the resident launches of gnnx_run execute on; gnnx_debug_spin keeps a stream busy for `micros` microseconds.  The
 * last two exist so that a caller can check which of ITS streams share a hardware queue with a lane (HIP binds streams to
 * GPU_MAX_HW_QUEUES queues round-robin; streams on one queue execute in order). */
int gnnx_set_service_stream(void* stream);
/* a hipStream_t restricted to the compute units set in `mask` (bit c of word c / 32 = CU c; hipExtStreamCreateWithCUMask): a pipelined job
 * reserves a few CUs for its prepare / fetch kernels - the workgroups of a resident launch hold their CUs for milliseconds.  NULL on failure. */
void* gnnx_stream_create_cu_mask(const uint32_t* mask, int32_t words);
void* gnnx_lane_stream(int32_t i);
int gnnx_debug_spin(void* stream, int32_t micros);
/* Self-check of the lane sums the kernels are built on (gnnx_kernels.hpp: xor32_sum / xor16_sum - gfx950's v_permlane32_swap / v_permlane16_swap
 * through inline assembly): in = 64 device floats (one wave), out = 256 device floats: [0, 64) xor32_sum, [64, 128) v + shfl_xor(v, 32),
 * [128, 192) xor16_sum, [192, 256) v + shfl_xor(v, 16).  The two forms of each must agree bit for bit (tests/test_gpu_lane_sums.py). */
int gnnx_debug_lane_sums(const float* in, float* out, void* stream);

/* The library keeps the device blocks of destroyed plans for the next plan (hipMalloc / hipFree per batch serialise a pipelined
 * job: hipFree synchronises the device); this returns the idle ones of the current device to the driver. */
int gnnx_pool_trim(void);

/* Packing of the one mixed launch of the edge-sparse resident kernels (gnnx_sparse.hpp: k_sparse_resident_mixed, sp_mix_tiny): how many
 * single-tile targets (n <= 32) share one 512-thread workgroup for a model of these widths - 8 when their LDS slices and the shared model
 * block fit the pool, else as many as fit (>= 5).  bench.py maps targets to workgroups with it for the executed-work roofline. */
int gnnx_sparse_tiny_per_workgroup(int32_t D, int32_t H, int32_t C);

/* Packed single-wave launch (gnnx_sparse.hpp: k_sparse_resident_tiny16 / 12; round 6): targets of the 64-thread class (n <= 32) whose slim LDS
 * form fits a slice run sixteen (GNNX_TINY_PACK=12: twelve) to a compute unit in a launch of their own, when the plan runs the algebraic
 * constant-feature form at the reference's widths (gnnx_plan_analyze_features; D = 10, H = O = 20, C <= 4); bit-identical to the class's
 * other launches (the same body).  *per_cu = 16 / 12, or 0 when the plan has no such launch; *n_packed = its targets; flags[t] = 1 for the
 * targets it takes (num_targets ints).  Each may be NULL.  The loop the packed targets run is explain.py:137-146, as for every resident class. */
int gnnx_tiny_pack_info(gnnx_handle h, int32_t* per_cu, int32_t* n_packed, int32_t* flags);

/* ---- the XL route: node-mode targets of ANY size, CSR-native (csrc/gnnx_xl.hpp, k_sparse_large<.., XL> in csrc/gnnx_sparse_large.hpp) ----------------
 *
 * What the reference does per target in Explainer.explain (explain.py:80-117: `sub_adj = adj[nb][:, nb]`, `sub_feat`, ExplainModule with its dense
 * n x n mask, :137-146 the loop, :208-211 the masked adjacency handed back) WITHOUT a dense n x n block anywhere: the sub-graphs are built as CSRs of
 * local ids from the resident full-graph CSR and the k-hop lists of gnnx_khop; masks, Adam moments and results are EDGE LISTS - for target t the
 * upper-triangle edges (r < c) of its sub-graph in row-major order, eoff[t] .. eoff[t + 1], one (M[r][c], M[c][r]) pair per edge: the layout of
 * gnnx_gather_edges.  Same arithmetic as route 7 (one source; bit-identical on targets both take), no limit on n, on the rows or entries within
 * two hops; a row within two hops of the target may hold at most 1024 entries.  The explain loop of a 100k-node graph (explainer_main.py:309-313 ->
 * explain.py:296-299) routes here every target the LDS-resident classes cannot take (BA-House x100k: 2.4 % of the nodes have sub-graphs of
 * 16 384 ... 49 028 nodes - 9.6 GB per dense array - and another 1.5 % exceed route 7's LDS budget).
 *
 *   gnnx_xl_create   plan for T targets: prob->n / target_row / gt_label as in gnnx_plan_create (node mode, sigmoid mask, no --bn, C <= 8)
 *   gnnx_xl_count    k_xl_rowdeg + k_xl_rowptr: every sub-graph row's entries among the list members (binary search in the ascending list), the
 *                    packed feature rows / predicted class ids, the row pointers; edges_host [T] (HOST, may be NULL) = upper-triangle edges per
 *                    target.  Synchronises `stream` (the host sizes the second workspace and the caller's edge lists from the counts).
 *   gnnx_xl_build    k_xl_emit: columns / rows / weights of every directed entry and rc [E][2] (DEVICE out) = the (r, c) local ids of every edge
 *   gnnx_xl_run      all num_iters iterations of every target in ONE launch, one workgroup per target (explain.py:137-146); asynchronous.
 *                    state->M_e [E][2] in: initial masks (construct_edge_mask, explain.py:645-652, on the edges), out: after the steps;
 *                    m_e / v_e / feat: optimiser state to resume from (gnnx_resume) or NULL; *_out: the state afterwards or NULL;
 *                    abar_e [E]: masked adjacency of the LAST forward (explain.py:209-211); feat_mask [T][32].
 *   gnnx_xl_set_trace  gates [num_iters][R][2] (R = gnnx_xl_total_rows) as gnnx_set_trace; reference widths only; NULL = off
 * Workspaces (DEVICE, caller-owned): ws_rows of gnnx_xl_rows_bytes (known at create), ws_entries of gnnx_xl_entries_bytes (known after count);
 * both must live from count to the last run.  indptr / indices / weights / nb / nb_off / feat / pred_label: as gnnx_pack_csr. */
typedef struct gnnx_xl_s* gnnx_xl_handle;
typedef struct {
    int32_t first_iter;
    float* M_e;
    const float* m_e;
    const float* v_e;
    const float* feat;   /* [T][3][32] */
    float* m_out_e;
    float* v_out_e;
    float* feat_out;     /* [T][3][32] */
} gnnx_xl_state;
int gnnx_xl_create(const gnnx_problem* prob, const gnnx_model* model, gnnx_xl_handle* out);
int gnnx_xl_destroy(gnnx_xl_handle h);
int64_t gnnx_xl_total_rows(gnnx_xl_handle h);
size_t gnnx_xl_rows_bytes(gnnx_xl_handle h);
int gnnx_xl_count(gnnx_xl_handle h, const int64_t* indptr, const int32_t* indices, const float* weights, const int32_t* nb, const int64_t* nb_off,
                  const float* feat, int32_t feat_stride, const float* pred_label, void* ws_rows, size_t ws_rows_bytes, int64_t* edges_host,
                  void* stream);
int64_t gnnx_xl_total_edges(gnnx_xl_handle h);
size_t gnnx_xl_entries_bytes(gnnx_xl_handle h);
int gnnx_xl_get_layout(gnnx_xl_handle h, int32_t* ld, int64_t* offR, int64_t* eoff); /* HOST out: ld [T], offR [T], eoff [T + 1]; any may be NULL */
int gnnx_xl_build(gnnx_xl_handle h, const int64_t* indptr, const int32_t* indices, const float* weights, const int32_t* nb, const int64_t* nb_off,
                  void* ws_rows, void* ws_entries, size_t ws_entries_bytes, int32_t* rc, void* stream);
/* gnnx_mt_edge_words for XL targets WITHOUT the n^2 scratch: seeds [T] (DEVICE int64) -> words [E][4] (DEVICE uint32) = the two raw mt19937 words of the
 * Box-Muller pair of M[r][c], then of M[c][r], for every edge - one pass of each target's engine over its n^2 (+ 16) draws, the entries picked as
 * their stream positions pass (k_mt_edge_words_xl).  The host finishes with the transform of include/gnnx_host.h - gnnx_host_transform_edge_words -: bit-identical to
 * torch.manual_seed(seed); torch.FloatTensor(n, n).normal_(1.0, std) on the edges (construct_edge_mask, explain.py:645-652). */
int gnnx_xl_mt_edge_words(gnnx_xl_handle h, const int64_t* seeds, void* ws_rows, void* ws_entries, uint32_t* words, void* stream);
/* The walk of ONE target's engine spread over many workgroups: mt19937 is linear over GF(2), so the state `jump_draws` draws ahead is the XOR of
 * sliding windows of the plain sequence selected by g(x) = x^jump mod phi(x) (utils/mt_jump.py computes phi and g; poly = HOST, 624 uint32, bit i of the
 * polynomial = bit i & 31 of word i >> 5; jump_draws a multiple of 624).  Once set (process-wide), gnnx_xl_mt_edge_words jumps from segment start to
 * segment start (k_mt_segment_starts) and walks the segments in parallel (k_mt_edge_words_seg): the 2.3e9 draws of a 47 913-node sub-graph in ~35 ms
 * instead of 1.7 s.  NULL / 0: serial walks. */
int gnnx_set_mt_jump_poly(const uint32_t* poly, int64_t jump_draws);
/* ... with the polynomials of the strides jump, 4 jump, 16 jump (polys [levels][624], levels <= 3: x^(4^l jump) mod phi - utils/mt_jump.jump_polys)
 * the segment starts of a target form a radix-4 tree instead of one chain of K - 1 jumps: (K - 1) / 16 + 6 jumps deep (the 274 segments of the
 * 47 913-node sub-graph: 23 instead of 273 dependent jumps). */
int gnnx_set_mt_jump_polys(const uint32_t* polys, int64_t jump_draws, int32_t levels);
int gnnx_xl_set_trace(gnnx_xl_handle h, uint32_t* gates);
/* Measurement hook: ticks [T][4] (DEVICE int64) receives, from every later gnnx_xl_run, the wall_clock64 value (100 MHz) at the start of target t's
 * workgroup, after its setup, after its iteration loop and at its end; NULL = off.  parallel.py calibrates the sharded job's cost model on them. */
int gnnx_xl_set_clocks(gnnx_xl_handle h, int64_t* ticks);
int gnnx_xl_run(gnnx_xl_handle h, const gnnx_hyper* hyper, const gnnx_xl_state* state, float* abar_e, float* feat_mask, void* ws_rows,
                void* ws_entries, void* stream);

const char* gnnx_last_error(void);
const char* gnnx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GNNX_H */
