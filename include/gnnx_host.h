/* gnnx_host.h — host-side helper of the MI355X-native GNNExplainer engine (libgnnx_host.so): the initial edge masks.
 *
 * The reference draws every target's initial mask with ONE torch CPU-generator call,
 *     mask = torch.FloatTensor(n, n); mask.normal_(1.0, calculate_gain("relu") * sqrt(2 / (n + n)))
 * (construct_edge_mask, explainer/explain.py:645-652), and parity needs exactly those bits (torch's vectorised Box-Muller over
 * mt19937 cannot be reproduced on the device).  In Python that call costs ~8 us of interpreter + dispatch per target under the
 * GIL (3.2 ms of an 8.5 ms syn1 batch on two threads).  This library makes the SAME ATen call - at::Tensor::normal_
 * on a view of the caller's buffer with a private at::Generator seeded per target - from plain C++ threads, no GIL, no Python.
 * Plain C ABI (pointers and sizes); links libtorch_cpu / libc10 of the PyTorch the process already has loaded.
 */
#ifndef GNNX_HOST_H
#define GNNX_HOST_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* out[off[k] .. off[k] + n[k]^2) = the n[k] x n[k] values of ONE normal_(1.0, sqrt(2) * sqrt(2 / (n + n))) draw from a CPU generator
 * seeded with seeds[k] (the seed protocol of the golden runs: torch.manual_seed(1000 + target) immediately before the explanation,
 * which makes the targets independent).  `out` is HOST memory (pinned or not), `threads` worker threads split the targets by
 * equal shares of the values.  Returns 0, or non-zero with the text in gnnx_host_last_error(). */
int gnnx_host_draw_masks(int32_t num_targets, const int32_t* n, const int64_t* seeds, const int64_t* off, float* out, int32_t threads);

/* The same draw with the slice length given: a target of more than 2 x slice_values values (a multiple of 16; the call above uses 2^21)
 * is drawn as slices of that many values by several threads - a walker steps the raw mt19937 stream of the target and leaves the engine
 * state at every slice boundary, ATen's normal_ then draws every slice from its state.  ATen's CPU normal_ fills a contiguous float
 * tensor with one engine draw per value before it transforms 16 values at a time, so the slices are bit-identical to the one-call draw;
 * without this the largest target of a batch (BA-House x100k: n = 5600, 31 M values) bounds the whole draw at 0.3 s. */
int gnnx_host_draw_masks_sliced(int32_t num_targets, const int32_t* n, const int64_t* seeds, const int64_t* off, float* out, int32_t threads,
                                int64_t slice_values);

/* The same draws, keeping only the values on the EDGES: out[e] = (M[r][c], M[c][r]) for edge e = (rc[2e], rc[2e + 1]), r < c, local node ids,
 * the edges of all targets target after target (eoff [T + 1]).  The edge-sparse kernels read the initial mask nowhere else.  The values on
 * the edges are positions of one mt19937 stream per target, so the engine passes over the whole stream - as STATE only: ATen's normal_ is one
 * engine draw per value, then a Box-Muller transform of 16 values at a time, and 624 = 39 x 16, so a walker regenerates the raw state block by
 * block (no tempering, no transform), copies the 16 raw words of every block that holds an edge entry into a staged engine state, and ATen
 * itself draws them from it, 38 synthetic blocks per normal_ call (only the Box-Muller pair that holds a wanted value is staged, eight pairs
 * to a synthetic block - checked once per process against ATen's own draw, whole blocks otherwise): bit-identical to the full draw by construction, at 0.4 instead of 4.6 ns per normal of
 * the stream, and only 2E values leave the host instead of sum(n^2) (12 MB instead of 4 GB for the 16 384-target BA-House x100k set).
 * slice_values: work-item length (large targets are cut into chunks of 32 * slice_values values from states a walker leaves behind). */
int gnnx_host_draw_edge_masks(int32_t num_targets, const int32_t* n, const int64_t* seeds, const int64_t* eoff, const int32_t* rc, float* out,
                              int32_t threads, int64_t slice_values);

/* The same values from the raw engine words the DEVICE picked (gnnx_mt_edge_words, include/gnnx.h): words [E][4] (HOST uint32) = for edge e
 * the two raw mt19937 state words of the Box-Muller pair that holds M[r][c], then those of M[c][r]; the host stages them - eight pairs to a
 * synthetic 16-value block - and ATen's own normal_ tempers and transforms them: out[e] = (M[r][c], M[c][r]), bit-identical to
 * gnnx_host_draw_edge_masks, without stepping through the n^2 draws of any target (targets of fewer than 16 values, which take ATen's
 * scalar path, are drawn here from their seed).  Needs the pair-staging property of the host's normal_ (gnnx_host_pair_staging_ok(): 1 / 0,
 * checked once per process against ATen's one-call draw); without it the caller falls back to gnnx_host_draw_edge_masks. */
int gnnx_host_pair_staging_ok(void);
int gnnx_host_transform_edge_words(int32_t num_targets, const int32_t* n, const int64_t* seeds, const int64_t* eoff, const int32_t* rc,
                                   const uint32_t* words, float* out, int32_t threads);

const char* gnnx_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* GNNX_HOST_H */
