"""Batches of targets through the whole hot path, stages overlapped: the steady state of `Explainer.explain_nodes` over many batches.

One batch = what the reference does per target in Explainer.explain (explainer/explain.py:74-221): neighbourhood extraction
(:492-501), ExplainModule construction with its seeded normal_ mask (:583-663), the optimisation loop (:137-146) and the masked
adjacency handed back (:208-221).  Here a batch runs as three stages on three HIP streams:

  prepare  (worker thread, stream P): k-hop walk sets on the device (gnnx_khop) -> plan -> device-side packing (gnnx_pack_csr) ->
           routing (gnnx_plan_analyze) -> edge layout (gnnx_edge_counts / _positions; the edge ids (r, c) go back to the host
           here, they do not depend on the optimisation), while C++ host threads draw the seeded initial masks
           (gnnx_host_draw_masks) into a pinned buffer -> one H2D copy + gnnx_scatter_masks;
  optimise (caller's thread, stream L): gnnx_run - all iterations;
  fetch    (caller's thread, stream F): gnnx_gather_values + D2H of the edge values and the feature masks into pinned buffers.

Batch k + 1 is prepared and batch k - 1 fetched while batch k optimises, so a long job costs max(stage) per batch instead of
their sum; nothing is shared between batches but the resident graph.  Results come back in order as engine.EdgeMasks.
"""
import os
import queue
import threading
import time
from collections import deque

import numpy as np
import torch

from . import engine


class _Part:
    """One route group of a prepared batch: the targets idx (positions in the batch) on a MaskOptimJob (dense-packed classes) or an XLJob (CSR-native)."""
    __slots__ = ("idx", "job", "xl", "rc", "eoff", "E")


class _Prepared:
    __slots__ = ("targets", "job", "dn", "ready", "rc", "eoff", "E", "times", "error", "launch_cus", "parts")


class BatchPipeline:
    def __init__(self, graph, state_dict, labels, hyper, n_hops=3, seed_base=1000, rng_threads=None, depth=None, prepare_workers=2, reserve_cus=0, lib=None,
                 device_hook=None, rng_threads_big=None, edge_draw=None, edge_draw_min_values=0.0, device_walk=None, xl_min_n=None, xl_device_walk=None):
        """graph: engine.DeviceGraph (resident); labels [N]: the label the prediction loss uses (explain.py:750-753); the initial
        mask of target v is drawn from a generator seeded with seed_base + v (the seed protocol of the golden runs)."""
        self.graph, self.sd, self.labels, self.hyper = graph, state_dict, np.asarray(labels), hyper
        import dataclasses
        self._edge_hyper = dataclasses.replace(hyper, edge_results_only=True)      # results leave as edge lists: no dense Abar blocks
        self.n_hops, self.seed_base = int(n_hops), int(seed_base)
        self.rng_threads = int(rng_threads) if rng_threads else engine.default_rng_threads()
        # batches of more than 2e7 normals (BA-House x100k: 1e9 per 16 384 targets = 4 GB written): measured on the 256-CPU host of the GPU box
        # (tools/probe_rng_big.py, profiles/r04_host_rng_16384targets.txt) 32 threads 94-140 ms, 64 threads 100-190, 96-128 threads 160-230 -
        # the draw is bound by the memory system, not the cores; the largest targets are cut into slices (gnnx_host_draw_masks_sliced:
        # the 31 M values of the n = 5600 target 63 -> 10-14 ms), which matters when one target dominates a batch
        self.rng_threads_big = int(rng_threads_big) if rng_threads_big else (int(rng_threads) if rng_threads else engine.default_rng_threads(big=True))
        # device_hook(values [E] on the device, job): called on the fetch stream once a batch's edge values are gathered, before their D2H
        # copy - the sharded job all-gathers the masks of every rank there (RCCL over xGMI; bench.py --gpus N)
        self.device_hook = device_hook
        # Large batches on the edge-sparse kernels keep only the EDGE entries of the host draw (see _prepare: 16 MB instead of 4 GB leave the host
        # for the 16 384-target BA-House x100k set).  History: the first form of gnnx_host_draw_edge_masks still ran ATen's normal_ over every
        # slice of the stream that held an edge entry - all of them, for targets of thousands of nodes - so it cost what the full draw costs
        # (4.6 ns of one core per normal: 154 ms full vs 197 ms edges only on the GPU box's 16-core quota, profiles/r04_host_rng_edges_16384targets.txt)
        # and stayed off.  The block-granular form passes over the stream as engine STATE only (0.3 ns per draw) and lets ATen transform just the
        # 16-value blocks that hold an entry (6.6 % of them on that set): 0.4 ns per normal of the stream, 560 -> 57-66 ms in the 8-core build
        # container (profiles/r04_host_rng_edges_blocks.txt), bit-identical - ON by default; edge_draw=False / GNNX_PIPE_EDGE_DRAW=0 restores the
        # full stream.
        self.edge_draw = bool(int(os.environ.get("GNNX_PIPE_EDGE_DRAW", "1"))) if edge_draw is None else bool(edge_draw)
        self.edge_draw_min_values = float(os.environ.get("GNNX_PIPE_EDGE_MIN", edge_draw_min_values))      # batches of fewer normals keep the full draw.  Round 6: 0 (was 2e7) - the edge draw
        # for small batches too: a syn1 batch's full draw is 2 M normals = 8 core-ms and 8 MB across PCIe for 45 KB of edge entries; A/B of the driver's command, three
        # alternating pairs: 267.1 / 237.9 / 267.4 k with the full draw, 275.2 / 266.2 / 271.1 k with the edge draw, host core-seconds per batch 0.0083 -> 0.0069
        # (tools/r6_edge_min_ab.sh).  Bit-identical masks (tests/test_pipeline.py).
        self.rng_threads_edges = int(os.environ.get("GNNX_PIPE_EDGE_THREADS", self.rng_threads_big))   # (the ranks of a node share its cores: the sharded bench passes its share)
        # Round 5: the engine of the edge draw walks on the DEVICE (gnnx_mt_edge_words: every target's mt19937 stream as raw state words, the two
        # words of each entry's Box-Muller pair gathered, 16 bytes per directed entry to the host) and the host only lets ATen transform those
        # pairs (gnnx_host_transform_edge_words): bit-identical, O(E) instead of O(sum n^2) host work - the 16 384-target BA-House x100k batch
        # cost 0.64 core-seconds per step with the host walk, which bound a node's ranks to its CPU quota beyond two GPUs.  Needs the
        # pair-staging property of the host's normal_ (checked once per process); device_walk=False / GNNX_PIPE_DEVICE_WALK=0: the host walk.
        # Default: the device walk when the ranks of a node share its cores (LOCAL_WORLD_SIZE > 1), the host walk for a single process - its 0.66
        # core-seconds per 16 384-target step fit one process's quota, and the device walk's kernels share the chip with the optimisation of the
        # batch before: 204.0 k (host) against 183.8 k nodes/s (device) on one GPU in the closing session (profiles/r05_bench_ba100k_16384targets*.json)
        auto_walk = "1" if int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > 1 else "0"
        self.device_walk = bool(int(os.environ.get("GNNX_PIPE_DEVICE_WALK", auto_walk))) if device_walk is None else bool(device_walk)
        # Round 6: targets of more than xl_min_n sub-graph nodes take the XL route (engine.XLJob: sub-graph CSRs built from the resident graph, no dense
        # n x n block, edge-list state) instead of being packed into dense blocks for k_sparse_large / the streaming kernels.  Beyond 16 383 nodes
        # there is no alternative (a dense block of the largest BA-House x100k sub-graph is 9.6 GB per array); below, the two kernels are the same
        # source and bit-identical (tests/test_xl_route.py) and the XL prepare stage is O(nnz) instead of O(n^2) - measured in profiles/r06_*.
        # GNNX_XL_MIN_N overrides; 0 / a huge value sends everything / nothing there.
        if xl_min_n is None:
            xl_min_n = int(os.environ.get("GNNX_XL_MIN_N", "512"))
        self.xl_min_n = int(xl_min_n)
        n_classes = int(np.asarray(state_dict["pred_model.weight"].detach().cpu() if torch.is_tensor(state_dict["pred_model.weight"]) else state_dict["pred_model.weight"]).shape[0])
        self.xl_ok = (not hyper.record_loss) and bool(hyper.use_resident) and n_classes <= 8      # (what gnnx_xl_create takes)
        # the XL targets' seeded masks: engine walk on the device (k_mt_edge_words_xl, one serial pass per target: the 2.3e9 draws of a 48 k-node
        # sub-graph take 1.7 s of one workgroup, hidden behind the other batches of a long job) or on the host (gnnx_host_draw_edge_masks: 0.3 ns per
        # draw and core).  Default: the device when the ranks of a node share its cores or the batch is large, else the host.
        self.xl_device_walk = (None if os.environ.get("GNNX_XL_DEVICE_WALK") is None else bool(int(os.environ["GNNX_XL_DEVICE_WALK"]))) if xl_device_walk is None else bool(xl_device_walk)
        # Optimisations in flight.  depth=None (default): as many as keep the chip full and no more - ceil(1.3 x 256 CUs / the compute units ONE
        # launch keeps busy), between 2 and 5, re-evaluated per batch (_launch_cus; four when a batch has streaming targets): every further launch in flight only queues behind the
        # others and lengthens the fill and drain of a short job.  Measured (profiles/r05_pipeline_workers_depth_room.txt): syn1 (116 workgroups
        # per launch) 20-batch regions 234-241 k nodes/s at four in flight, 249-250 k at three; Tree-Cycles (360 single-wave workgroups, six per
        # CU) 365-373 k at three, 449-455 k at four or more; steady state (300 batches) indifferent.  A number fixes it.
        self.record_launch_ms = os.environ.get("GNNX_PIPE_LAUNCH_MS", "1") != "0"   # per-batch device time of the optimisation launch (eight event queries per batch)
        env_depth = os.environ.get("GNNX_PIPE_DEPTH")                               # (measurement knobs)
        if env_depth:
            depth = int(env_depth)
        reserve_cus = int(os.environ.get("GNNX_PIPE_RESERVE", reserve_cus))
        prepare_workers = int(os.environ.get("GNNX_PIPE_WORKERS", prepare_workers))
        self.auto_depth = depth is None
        self.depth = 5 if depth is None else max(1, int(depth))     # streams (2 prepare + 1 fetch + 5 optimise = the 8 hardware queues)
        self.depth_now = 3 if depth is None else self.depth
        self.lib = lib if lib is not None else engine.get_library()
        dev = graph.feat.device
        self.device = dev
        # `depth` optimisations in flight, each on its own optimise stream (the library rotates its launch lanes from run to run): a
        # batch that fills half the GPU for the 3.75 ms of its slowest workgroup leaves room for the next one to start beside it
        # (reserve_cus > 0 keeps that many compute units out of the optimise streams' CU masks for the prepare / fetch kernels -
        # measured on syn1: 16 reserved CUs made the prepare stage 11 instead of 2.3 ms, its kernels crawl on 16 CUs: off by default)
        self._masks = self._cu_masks(reserve_cus)
        self._rejected = []        # streams that share a hardware queue with one already chosen (kept alive: their queue slot stays taken)
        self._chosen = []
        self.prepare_workers = max(1, int(prepare_workers))
        self.s_loops = [self._distinct_stream("loop") for _ in range(self.depth)]
        self.s_loop = self.s_loops[0]
        self.s_preps = [self._distinct_stream("aux") for _ in range(self.prepare_workers)]
        self.s_prep = self.s_preps[0]
        self.s_fetch = self._distinct_stream("aux")
        self._pinned = {}          # (name, slot) -> grow-only pinned host buffers
        self._raw_done = {}        # raw slot -> event behind the H2D copy that read it
        self._ring = 0
        # pinned staging slots of the small per-batch host buffers (edge ids): a slot is reused by batch k + slots, so it must outnumber
        # the batches whose buffer finish() has not copied yet: those being prepared + those optimising + those queued for / in fetch
        self._rc_slots = self.prepare_workers + 2 * self.depth + 4
        self.stats = []            # per batch: host milliseconds of the stages (measurement)

    def _cu_masks(self, reserve, num_cus=256):
        """(mask of the optimise streams, mask of the prepare / fetch streams): `reserve` CUs spread evenly over the chip for the latter."""
        if not reserve:
            return None
        aux = np.zeros(num_cus, bool)
        aux[::max(1, num_cus // reserve)][:reserve] = True
        words = lambda bits: np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).ravel()
        return words(~aux), words(aux)

    def _new_stream(self, kind):
        import ctypes
        if self._masks is not None and kind == "loop":      # only the optimise streams are restricted; prepare / fetch kernels may use every CU
            m = np.ascontiguousarray(self._masks[0])
            ptr = self.lib.gnnx_stream_create_cu_mask(m.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), len(m))
            if ptr:
                return torch.cuda.ExternalStream(ptr, device=self.device)
        # prepare / fetch streams at high priority: with the batches in flight filling the chip, their short kernels take the next
        # compute unit a finishing optimisation workgroup frees instead of queueing behind the pending workgroups of the next batch
        return torch.cuda.Stream(self.device, priority=-1 if kind == "aux" else 0)

    # -- streams -------------------------------------------------------------------------------------------------------------
    def _distinct_stream(self, kind, tries=16, spin_us=2000):
        """A new stream on a hardware queue none of the pipeline's other streams uses.  HIP binds a stream to one of GPU_MAX_HW_QUEUES
        (<= 8) queues when it is created and executes the packets of one queue in order, so two optimise streams on one queue would run
        their batches one after the other, and a prepare / fetch stream that lands on the queue where a 4 ms optimisation runs makes the
        next batch's k-hop kernels wait for that launch - the stages would not overlap (measured: 4.1 instead of 0.3 ms for the k-hop pass,
        and end-to-end rates between 119 k and 152 k nodes/s from run to run before the streams were checked against each other).  Found
        by trial: keep the streams already chosen busy for 2 ms each (gnnx_debug_spin), time a trivial operation on the candidate."""
        dev = self.device
        probe = torch.zeros(64, dtype=torch.int32, device=dev)
        for _ in range(tries):
            cand = self._new_stream(kind)
            if not self._chosen:
                self._chosen.append(cand)
                return cand
            torch.cuda.synchronize(dev)
            for st in self._chosen:
                self.lib.gnnx_debug_spin(st.cuda_stream, spin_us)
            t0 = time.perf_counter()
            with torch.cuda.stream(cand):
                probe.add_(1)
            cand.synchronize()
            waited = time.perf_counter() - t0
            torch.cuda.synchronize(dev)
            if waited < 0.3 * spin_us * 1e-6:
                self._chosen.append(cand)
                return cand
            self._rejected.append(cand)
        cand = self._new_stream(kind)      # no free queue found: some stages then serialise, correctly but slower
        self._chosen.append(cand)
        self.queue_fallbacks = getattr(self, "queue_fallbacks", 0) + 1
        return cand

    @staticmethod
    def _wait(stream, blocking):
        """Wait until `stream` has drained.  stream.synchronize() SPINS (HIP's default wait: one core at 100 % for as long as the GPU works) -
        fine for the sub-millisecond waits of a small batch, but a large batch waits tens of milliseconds per stage, and the ranks of a node
        share its cores: those waits go through an event created with hipEventBlockingSync, which sleeps."""
        if not blocking:
            stream.synchronize()
            return
        ev = torch.cuda.Event(blocking=True)
        ev.record(stream)
        ev.synchronize()

    # -- pinned staging ----------------------------------------------------------------------------------------------------
    def _pin(self, name, numel, dtype, slot):
        key = (name, slot)
        buf = self._pinned.get(key)
        if buf is None or buf.numel() < numel:
            buf = torch.empty(int(numel * 1.25) + 256, dtype=dtype, pin_memory=True)
            self._pinned[key] = buf
        return buf[:numel]

    @staticmethod
    def _launch_cus(route, num_cus=256, tiny_pack=(0, 0)):
        """Compute units one optimisation launch of this batch keeps busy (an estimate from the routing: gnnx_sparse.hpp's classes - a 512-thread
        target, a PAIR of 256-thread targets or eight single-wave targets per workgroup of the mixed launch, one workgroup per CU; alone, the
        single-wave class packs six workgroups per CU and the 256-thread class two; the packed single-wave launch, `tiny_pack` = (targets per
        CU, targets), sixteen or twelve targets per CU)."""
        route = np.asarray(route)
        n8, n5, n6 = int((route == 8).sum()), int((route == 5).sum()), int((route == 6).sum())
        packed_cus = 0
        if tiny_pack[0] and tiny_pack[1]:
            n6 -= int(tiny_pack[1])
            packed_cus = -(-int(tiny_pack[1]) // int(tiny_pack[0]))
        n47 = int(np.isin(route, (4, 7)).sum())
        other = int((~np.isin(route, (4, 5, 6, 7, 8))).sum())
        if n8 or (n5 and n6):
            cus = n8 + (n5 + 1) // 2 + (n6 + 7) // 8
        else:
            cus = (n5 + 1) // 2 + (n6 + 5) // 6
        return min(cus + n47 + packed_cus, num_cus) if not other else -1      # (-1: streaming targets in the batch)

    # -- stage 1 -----------------------------------------------------------------------------------------------------------
    def _prepare(self, targets, k, s_prep):
        p = _Prepared()
        p.targets, p.error, p.times = targets, None, {}
        t0 = time.perf_counter()
        self.lib.gnnx_set_service_stream(s_prep.cuda_stream)    # this thread's plan-table uploads: not the null stream
        slot = k % self._rc_slots                              # small host buffers (edge ids): more slots than batches in flight (see __init__)
        raw_slot = k % (self.prepare_workers + 1)              # the RNG stream (n^2 floats per target): as few as the preparations in flight;
        ev = self._raw_done.get(raw_slot)                      # it is free again once its H2D copy has finished
        if ev is not None:
            ev.synchronize()
        with torch.cuda.stream(s_prep):
            dn = engine.khop_device(self.graph, targets, self.n_hops, lib=self.lib, one_pass=True)
            p.times["khop_ms"] = (time.perf_counter() - t0) * 1e3
            if (dn.rows < 0).any():
                raise ValueError("a target is not in its own %d-hop walk set (isolated node): the reference fails on it too" % self.n_hops)
            big = (dn.sizes > self.xl_min_n) if self.xl_ok else np.zeros(len(targets), bool)
            p.parts = []
            p.launch_cus = 0
            if not big.all():
                idx = np.nonzero(~big)[0]
                whole = len(idx) == len(targets)
                p.parts.append(self._prepare_dense(p, idx, targets if whole else targets[idx], dn if whole else dn.subset(idx), slot, raw_slot, s_prep))
            if big.any():
                idx = np.nonzero(big)[0]
                whole = len(idx) == len(targets)
                p.parts.append(self._prepare_xl(p, idx, targets if whole else targets[idx], dn if whole else dn.subset(idx), slot, raw_slot, s_prep))
                p.times["xl_targets"] = float(len(idx))
            p.ready = torch.cuda.Event()
            p.ready.record(s_prep)
            self._raw_done[raw_slot] = p.ready
        first = p.parts[0]
        p.job, p.dn, p.rc, p.eoff, p.E = first.job, dn, first.rc, first.eoff, first.E        # (single-part batches: the attributes of rounds 3-5)
        p.times["prepare_ms"] = (time.perf_counter() - t0) * 1e3
        return p

    def _prepare_xl(self, p, idx, targets, dn, slot, raw_slot, s_prep):
        """The XL part of a batch (stream s_prep is current): sub-graph CSRs from the resident graph, seeded masks on the edges."""
        t1 = time.perf_counter()
        xj = engine.XLJob(self.graph, dn, None, self.labels[targets], self.sd, lib=self.lib)       # count (one host wait) + build
        p.times["xl_count_build_ms"] = (time.perf_counter() - t1) * 1e3
        E = xj.E
        rc_host = self._pin("rc_xl", 2 * max(E, 1), torch.int32, slot).view(-1, 2)
        rc_host[:E].copy_(xj._rc[:E], non_blocking=True)
        vals = self._pin("edge_vals_xl", 2 * max(E, 1), torch.float32, raw_slot)[:2 * E].view(-1, 2)
        t2 = time.perf_counter()
        seeds = self.seed_base + targets
        dw = self.xl_device_walk
        if dw is None:      # auto: the device when the host is the shared resource or the batch is large; a handful of huge targets: the host
            dw = self.device_walk or len(targets) >= 64
        if dw and engine.pair_staging_ok():
            words_d = xj.draw_edge_words_device(seeds)
            words_h = self._pin("edge_words_xl", 4 * max(E, 1), torch.int32, raw_slot)[:4 * E].view(-1, 4)
            words_h.copy_(words_d, non_blocking=True)
            self._wait(s_prep, True)
            p.times["xl_device_walk_ms"] = (time.perf_counter() - t2) * 1e3
            t3 = time.perf_counter()
            engine.transform_edge_words(dn.sizes, seeds, xj._eoff, rc_host[:E], words_h, threads=self.rng_threads_edges, out=vals)
            p.times["xl_host_transform_ms"] = (time.perf_counter() - t3) * 1e3
        else:
            self._wait(s_prep, True)
            engine.init_edge_masks_on_edges(dn.sizes, seeds, xj._eoff, rc_host[:E], threads=self.rng_threads_edges, out=vals)
        p.times["xl_masks_ms"] = (time.perf_counter() - t2) * 1e3
        xj.set_masks_on_edges(vals)
        p.times["host_rng_edges_only"] = 1.0
        part = _Part()
        part.idx, part.job, part.xl, part.rc, part.eoff, part.E = idx, xj, True, rc_host, xj._eoff, E
        p.launch_cus += min(len(targets), 256)
        return part

    def _prepare_dense(self, p, idx, targets, dn, slot, raw_slot, s_prep):
        """The dense-packed part of a batch (stream s_prep is current): plan, device-side packing, routing, edge layout, seeded masks."""
        # the seeded masks only need the sizes: C++ threads draw them while the device builds the plan
        box = {}

        def draw():
            t_r = time.perf_counter()
            total = int((dn.sizes.astype(np.int64) ** 2).sum())
            try:
                box["raw"] = engine.init_edge_masks_raw(dn.sizes, seeds=self.seed_base + targets,
                                                        threads=self.rng_threads_big if total > 2e7 else self.rng_threads,
                                                        out=self._pin("raw", total, torch.float32, raw_slot))
            except Exception as e:      # noqa: BLE001 - re-raised on the preparing thread
                box["err"] = e
            box["ms"] = (time.perf_counter() - t_r) * 1e3
        # Batches of more than 2e7 normals whose every target runs on an edge-sparse kernel: the host keeps only the values on the
        # EDGES of its draw (gnnx_host_draw_edge_masks: 12 MB instead of 4 GB for the 16 384-target BA-House x100k set - no 4 GB of
        # pinned writes, H2D copy and scatter).  It needs the edge list first, so the draw follows the plan instead of overlapping it.
        total_values = int((dn.sizes.astype(np.int64) ** 2).sum())
        # (only when the resident kernels will really take the batch: with use_resident off the dense streaming kernels run, and those read
        #  and update EVERY entry of M - ADVICE r4)
        edges_only = (self.edge_draw and total_values > self.edge_draw_min_values and not self.hyper.record_loss and
                      bool(self.hyper.use_resident))
        th = threading.Thread(target=draw)
        if not edges_only:
            th.start()
        t1 = time.perf_counter()
        job = engine.MaskOptimJob.from_csr(self.graph, dn, None, self.labels[targets], self.sd, lib=self.lib)
        p.times["plan_pack_route_ms"] = (time.perf_counter() - t1) * 1e3
        t1b = time.perf_counter()
        job._edge_layout()
        E = int(job._eoff[-1])
        rc_host = self._pin("rc", 2 * max(E, 1), torch.int32, slot).view(-1, 2)
        rc_host[:E].copy_(job._rc[:E], non_blocking=True)
        p.times["edge_layout_ms"] = (time.perf_counter() - t1b) * 1e3
        p.times["plan_pack_route_layout_ms"] = (time.perf_counter() - t1) * 1e3
        lc = self._launch_cus(job.route(), tiny_pack=job.tiny_pack())
        p.launch_cus = -1 if (lc < 0 or p.launch_cus < 0) else p.launch_cus + lc
        if edges_only and not np.isin(job.route(), (4, 5, 6, 7, 8)).all():
            edges_only = False        # a target streams dense blocks: it needs every entry of its mask
            th.start()
        t1c = time.perf_counter()
        if edges_only:
            vals = self._pin("edge_vals", 2 * max(E, 1), torch.float32, raw_slot)[:2 * E].view(-1, 2)
            if self.device_walk and engine.pair_staging_ok():
                words_d = job.draw_edge_words_device(self.seed_base + targets)        # the engine walk + pair gather, enqueued on this stream
                words_h = self._pin("edge_words", 4 * max(E, 1), torch.int32, raw_slot)[:4 * E].view(-1, 4)
                words_h.copy_(words_d, non_blocking=True)
                self._wait(s_prep, True)      # the edge ids and the word pairs are on the host now (tens of milliseconds: sleep, do not spin)
                p.times["device_walk_ms"] = (time.perf_counter() - t1c) * 1e3
                t1d = time.perf_counter()
                engine.transform_edge_words(dn.sizes, self.seed_base + targets, job._eoff, rc_host[:E], words_h, threads=self.rng_threads_edges, out=vals)
                p.times["host_transform_ms"] = (time.perf_counter() - t1d) * 1e3
                p.times["host_rng_device_walk"] = 1.0
            else:
                self._wait(s_prep, True)      # the edge ids are on the host now
                engine.init_edge_masks_on_edges(dn.sizes, self.seed_base + targets, job._eoff, rc_host[:E], threads=self.rng_threads_edges, out=vals)
            p.times["host_rng_ms"] = (time.perf_counter() - t1c) * 1e3
            p.times["host_rng_edges_only"] = 1.0
            t2 = time.perf_counter()
            job.set_masks_on_edges(vals)
        else:
            th.join()
            p.times["wait_for_rng_ms"] = (time.perf_counter() - t1c) * 1e3
            if "err" in box:
                raise box["err"]
            p.times["host_rng_ms"] = box["ms"]
            t2 = time.perf_counter()
            job.set_masks_raw(box["raw"])
        p.times["h2d_scatter_enqueue_ms"] = (time.perf_counter() - t2) * 1e3
        part = _Part()
        part.idx, part.job, part.xl, part.rc, part.eoff, part.E = idx, job, False, rc_host, job._eoff, E
        return part

    def _worker(self, batches, out_q):
        """Feeds out_q with prepared batches IN ORDER; the preparation itself runs on `prepare_workers` threads, each with its own
        stream (a batch's plan + pack + layout is ~2 ms of host and device work; the optimise stage takes a new batch every ~1.9 ms)."""
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(self.prepare_workers)
        pending = deque()
        try:
            for k, targets in enumerate(batches):
                targets = np.ascontiguousarray(targets, np.int64)
                pending.append(pool.submit(self._prepare, targets, k, self.s_preps[k % self.prepare_workers]))
                while len(pending) >= self.prepare_workers + 1:
                    if not self._emit(pending.popleft(), out_q):
                        return
            while pending:
                if not self._emit(pending.popleft(), out_q):
                    return
        finally:
            pool.shutdown(wait=False)
            out_q.put(None)

    @staticmethod
    def _emit(fut, out_q):
        try:
            out_q.put(fut.result())
            return True
        except Exception as e:      # noqa: BLE001 - handed to the consumer
            p = _Prepared()
            p.error = e
            out_q.put(p)
            return False

    # -- stages 2 + 3 ------------------------------------------------------------------------------------------------------
    def _launch(self, p, slot):
        out, hooked = [], []
        for pi, part in enumerate(p.parts):
            job = part.job
            s_loop = self.s_loops[(slot + pi) % len(self.s_loops)]          # (the two parts of a mixed batch optimise side by side)
            with torch.cuda.stream(s_loop):
                s_loop.wait_event(p.ready)
                job.use_stream(s_loop)
                job.launch(self.hyper if part.xl else self._edge_hyper)
                done = torch.cuda.Event()
                done.record(s_loop)
            with torch.cuda.stream(self.s_fetch):
                self.s_fetch.wait_event(done)
                job.use_stream(self.s_fetch)
                vals_d = job.gather_edges_device()
                hooked.append(vals_d[:part.E])
                nm = "_xl" if part.xl else ""
                vals = self._pin("vals" + nm, max(part.E, 1), torch.float32, slot)
                vals[:part.E].copy_(vals_d[:part.E], non_blocking=True)
                fm = self._pin("fmask" + nm, job.T * engine.FEAT_STRIDE, torch.float32, slot).view(job.T, engine.FEAT_STRIDE)
                fm.copy_(job.fmask, non_blocking=True)
                out.append((vals, fm))
        with torch.cuda.stream(self.s_fetch):
            if self.device_hook is not None:      # ONCE per batch (a collective: every rank must call it equally often, whatever its batch's parts)
                self.device_hook(hooked[0] if len(hooked) == 1 else torch.cat(hooked), p.parts[0].job)
            fetched = torch.cuda.Event(blocking=p.times.get("host_rng_edges_only", 0.0) > 0)      # (a large batch: the caller sleeps through its tens of milliseconds)
            fetched.record(self.s_fetch)
        return out, None, fetched

    @staticmethod
    def _merge(n_targets, parts, fetched_parts, D):
        """EdgeMasks of a mixed batch in TARGET order from its parts' edge lists (each part: its targets ascending in batch position)."""
        counts = np.zeros(n_targets, np.int64)
        for part in parts:
            counts[part.idx] = np.diff(part.eoff)
        eoff = np.zeros(n_targets + 1, np.int64)
        np.cumsum(counts, out=eoff[1:])
        E = int(eoff[-1])
        rc = np.zeros((E, 2), np.int32)
        vals = np.zeros(E, np.float32)
        fm = np.zeros((n_targets, D), np.float32)
        n = np.zeros(n_targets, np.int32)
        for part, (pv, pf) in zip(parts, fetched_parts):
            pe = int(part.eoff[-1])
            if pe:
                # destination of the part's e-th edge: its target's start in the merged list + its offset inside the target
                cnt = np.diff(part.eoff)
                dst = np.repeat(eoff[part.idx] - part.eoff[:-1], cnt) + np.arange(pe)
                rc[dst] = part.rc[:pe].numpy()
                vals[dst] = pv[:pe].numpy()
            fm[part.idx] = pf.numpy()[:, :D]
            n[part.idx] = part.job.n
        return n, eoff, rc, vals, fm

    def run(self, batches):
        """batches: iterable of int arrays of target node ids.  Yields one engine.EdgeMasks per batch, in order."""
        q = queue.Queue(maxsize=min(self.depth, 3) + 1)      # prepared batches waiting for their launch
        th = threading.Thread(target=self._worker, args=(iter(batches), q), daemon=True)
        th.start()
        pending = deque()
        slot = 0

        def finish(item):
            p, got, _, fetched = item
            fetched.synchronize()
            if len(p.parts) == 1:
                part, (vals, fm) = p.parts[0], got[0]
                em = engine.EdgeMasks(part.job.n.copy(), part.eoff, part.rc[:part.E].numpy().copy(), vals[:part.E].numpy().copy(),
                                      fm.numpy()[:, :part.job.D].copy())
            else:
                em = engine.EdgeMasks(*self._merge(len(p.targets), p.parts, got, p.parts[0].job.D))
            em.neighbors = p.dn
            em.routes = [("xl" if part.xl else "dense", len(part.idx)) for part in p.parts]
            # device time of this batch's optimisation launch IN the pipeline (HIP events on its launch stream; the launch shares the chip with
            # the batches ahead and behind - the isolated launch time is what bench.py measures afterwards): the longest resident launch
            if self.record_launch_ms:
                try:
                    p.times["launch_in_pipeline_ms"] = max(max(part.job.resident_times()) for part in p.parts if not part.xl)
                except (ValueError, RuntimeError):
                    pass      # (XL parts only / a plan without resident launches)
            self.stats.append(p.times)
            for part in p.parts:
                if part.xl:
                    part.job.lib.gnnx_xl_destroy(part.job.handle)      # (the fetch has completed: nothing reads its tables any more)
                    part.job.handle = None
                else:
                    part.job.close()      # the plan's device tables go back to the library's pool (no hipFree: gnnx_capi.hip)
            return em

        while True:
            p = q.get()
            if p is None:
                break
            if p.error is not None:
                raise p.error
            if self.auto_depth:
                # (batches with streaming targets: four, as measured - 4.78 k nodes/s on the 1024-target dense sample against 3.87 k at two: the
                #  tail of one batch's 300 launches overlaps the head of the next)
                self.depth_now = 4 if p.launch_cus < 0 else int(min(self.depth, max(2, -(-(13 * 256) // (10 * max(1, p.launch_cus))))))
            pending.append((p,) + self._launch(p, slot))
            slot = (slot + 1) % (self.depth * 8)      # (a multiple of the optimise streams: launch k goes to stream k mod depth)
            while len(pending) > self.depth_now:
                yield finish(pending.popleft())
        while pending:
            yield finish(pending.popleft())
        th.join()
